// Host-side runtime of the HiFiGAN generator: owns the (re-laid-out) weights, the activation
// workspace and the launch schedule.  One object per process/GPU; no global state.
//
// Schedule of the bf16 path (all activations time-major / channel-last bf16 in HBM):
//   mel[B,n_mels,T] fp32 --transpose--> [B,T,n_mels]
//   conv_pre (epilogue applies the leaky-relu the first upsampler wants)
//   per stage: polyphase ConvTranspose (2-tap conv -> u*C_out channels, rows land contiguously),
//              per MRF branch: (c1: lrelu on load, lrelu in epilogue) -> (c2: + residual) x3,
//              the branch's last conv scales by 1/num_kernels and accumulates into the stage
//              output; the last branch also applies the next consumer's leaky-relu
//   conv_post + tanh (VALU, HBM-bound) -> wav[B,1,T*hop] fp32
// Every convolution's input activation is therefore applied exactly once per element, either
// in the producer's epilogue or on load of the residual stream.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "conv_tc_mfma.h"
#include "resblock_branch_kernel.h"
#include "resblock_pair_kernel.h"

namespace evmi {

static thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}
const char* last_error_cstr() { return g_last_error.c_str(); }

// kernels from the other translation units
int launch_conv1d_f32(const float*, const float*, const float*, const float*, float*, int, int, int, int,
                      int, int, int, int, int, float, float, int, hipStream_t);
int launch_conv_transpose1d_f32(const float*, const float*, const float*, float*, int, int, int, int, int,
                                int, int, float, hipStream_t);
int launch_tanh_f32(float*, long long, hipStream_t);
int launch_nct_f32_to_tc_bf16(const float*, bf16_t*, int, int, int, hipStream_t);
int launch_conv_post_tanh(const bf16_t*, const float*, float, float*, int, int, int, int, float, hipStream_t);
int launch_istft_head(const bf16_t*, const bf16_t*, const float*, float*, int, int, int, hipStream_t);
int launch_reflect_pad_left1_f32(const float*, float*, long long, int, float, hipStream_t);
int launch_istft_f32(const float*, float*, int, int, hipStream_t);

struct WeightSpec {
  std::string name;
  int64_t numel;
};

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  ~DevBuf() {
    if (p) (void)hipFree(p);
  }
  int ensure(size_t n) {
    if (bytes >= n) return EVMI_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    hipError_t e = hipMalloc(&p, n);
    if (e != hipSuccess) return fail(EVMI_ERR_OOM, std::string("hipMalloc workspace: ") + hipGetErrorString(e));
    bytes = n;
    return EVMI_OK;
  }
};

// One convolution of the bf16 schedule, weights already on the device in kernel layout.
struct TcConv {
  std::string layer;
  const ConvTcLaunch* launch = nullptr;
  int c_in = 0, c_out = 0, ks = 0, dil = 1, pad = 0;
  size_t w_off = 0;     // element offset into the bf16 weight arena
  size_t bias_off = 0;  // element offset into the fp32 bias arena
};

}  // namespace evmi

using namespace evmi;

struct evmi_generator {
  evmi_generator_config cfg;
  int device = 0;
  bool finalized = false;
  std::vector<WeightSpec> specs;
  std::map<std::string, std::vector<float>> host_w;

  // fp32 path: weights in torch layout, one arena
  DevBuf f32_arena;
  std::map<std::string, size_t> f32_off;  // element offsets

  // bf16 path
  bool tc_ok = false;
  std::string tc_why;
  DevBuf tc_w_arena, tc_bias_arena, post_w;
  float post_bias = 0.f;
  TcConv tc_pre;
  std::vector<TcConv> tc_ups;
  std::vector<std::vector<TcConv>> tc_rb;  // [resblock index][conv order: c1_0, c2_0, c1_1, ...]
  // fused (c1, c2) pair launches where the working set fits LDS; nullptr -> two conv_tc launches
  std::vector<std::vector<const PairLaunch*>> tc_pair;  // [resblock index][pair]
  std::vector<std::vector<size_t>> tc_pair_w;           // element offsets of the pair's [KS][C][C] weights (w1 then w2)
  int n_cu = 256;
  size_t istft_w_off = 0, istft_b_off = 0;
  bool use_pairs = true;

  DevBuf ws;

  int ch(int stage) const { return cfg.upsample_initial_channel >> stage; }  // channels entering stage
  int hop() const {
    int h = 1;
    for (int i = 0; i < cfg.num_upsamples; ++i) h *= cfg.upsample_rates[i];
    return cfg.istft_layer ? h * cfg.istft_hop : h;
  }
};

static std::string rb_name(const evmi_generator_config& c, int n, int which, int m, const char* leaf) {
  char buf[96];
  if (c.resblock_type == 1)
    snprintf(buf, sizeof buf, "resblocks.%d.convs%d.%d.%s", n, which, m, leaf);
  else
    snprintf(buf, sizeof buf, "resblocks.%d.convs.%d.%s", n, m, leaf);
  return buf;
}

static void build_specs(evmi_generator* g) {
  const auto& c = g->cfg;
  auto add = [&](const std::string& n, int64_t numel) { g->specs.push_back({n, numel}); };
  add("conv_pre.weight", (int64_t)c.upsample_initial_channel * c.n_mels * 7);
  add("conv_pre.bias", c.upsample_initial_channel);
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int cin = g->ch(i), cout = g->ch(i + 1);
    add("ups." + std::to_string(i) + ".weight", (int64_t)cin * cout * c.upsample_kernel_sizes[i]);
    add("ups." + std::to_string(i) + ".bias", cout);
  }
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int cc = g->ch(i + 1);
    for (int j = 0; j < c.num_kernels; ++j) {
      const int n = i * c.num_kernels + j, k = c.resblock_kernel_sizes[j];
      for (int m = 0; m < c.num_dilations[j]; ++m) {
        add(rb_name(c, n, 1, m, "weight"), (int64_t)cc * cc * k);
        add(rb_name(c, n, 1, m, "bias"), cc);
        if (c.resblock_type == 1) {
          add(rb_name(c, n, 2, m, "weight"), (int64_t)cc * cc * k);
          add(rb_name(c, n, 2, m, "bias"), cc);
        }
      }
    }
  }
  const int c_last = g->ch(c.num_upsamples);
  const int post_out = c.istft_layer ? c.istft_n_fft + 2 : 1;
  add("conv_post.weight", (int64_t)post_out * c_last * 7);
  add("conv_post.bias", post_out);
}

static int validate_cfg(const evmi_generator_config& c) {
  if (c.n_mels <= 0 || c.upsample_initial_channel <= 0) return fail(EVMI_ERR_INVALID_ARG, "config: channels");
  if (c.num_upsamples <= 0 || c.num_upsamples > EVMI_MAX_UPSAMPLES) return fail(EVMI_ERR_INVALID_ARG, "config: num_upsamples");
  if (c.num_kernels <= 0 || c.num_kernels > EVMI_MAX_RESBLOCK_KERNELS) return fail(EVMI_ERR_INVALID_ARG, "config: num_kernels");
  if (c.resblock_type != 1 && c.resblock_type != 2) return fail(EVMI_ERR_INVALID_ARG, "config: resblock_type");
  if (c.upsample_initial_channel % (1 << c.num_upsamples)) return fail(EVMI_ERR_INVALID_ARG, "config: initial channel not divisible by 2^num_upsamples");
  for (int i = 0; i < c.num_upsamples; ++i)
    if (c.upsample_rates[i] <= 0 || c.upsample_kernel_sizes[i] < c.upsample_rates[i] ||
        (c.upsample_kernel_sizes[i] - c.upsample_rates[i]) % 2)
      return fail(EVMI_ERR_INVALID_ARG, "config: upsample kernel/rate");
  for (int j = 0; j < c.num_kernels; ++j) {
    if (c.resblock_kernel_sizes[j] <= 0 || c.resblock_kernel_sizes[j] % 2 == 0) return fail(EVMI_ERR_INVALID_ARG, "config: resblock kernel must be odd");
    if (c.num_dilations[j] <= 0 || c.num_dilations[j] > EVMI_MAX_DILATIONS) return fail(EVMI_ERR_INVALID_ARG, "config: num_dilations");
    for (int m = 0; m < c.num_dilations[j]; ++m)
      if (c.resblock_dilations[j][m] <= 0) return fail(EVMI_ERR_INVALID_ARG, "config: dilation");
  }
  return EVMI_OK;
}

// ---- bf16 weight preparation -----------------------------------------------------------------------
// conv weight w[c_out][c_in][ks] (torch) -> kernel layout [mtile][chunk][tap][BM][KC] bf16 (ConvTcLaunch::wlayout)
static void relayout_conv(const float* w, int c_out, int c_in, int ks, const ConvTcLaunch* L,
                          std::vector<uint16_t>& arena, size_t off) {
  const int BM = L->bm, KC = L->kc, nch = c_in / KC;
  for (int m = 0; m < c_out; ++m)
    for (int j = 0; j < ks; ++j)
      for (int c = 0; c < c_in; ++c) {
        const int mt = m / BM, mi = m % BM, chn = c / KC, ci = c % KC;
        const int cs = L->wlayout == 1   ? ((((ci >> 3) ^ ((mi >> 1) & 7)) << 3) | (ci & 7))  // swizzled 16-byte slot, 128-byte rows
                       : L->wlayout == 3 ? ((((ci >> 3) ^ ((mi >> 2) & 3)) << 3) | (ci & 7))  // 64-byte rows (tools/microbench/conv_tc_pp_kernel.h)
                                         : ci;
        const size_t dst = ((((size_t)mt * nch + chn) * ks + j) * BM + mi) * KC + cs;
        arena[off + dst] = f32_to_bf16_bits(w[((size_t)m * c_in + c) * ks + j]);
      }
}

static int prepare_tc(evmi_generator* g) {
  const auto& c = g->cfg;
  g->tc_ok = false;
  if (c.istft_layer && (c.istft_n_fft != 16 || c.istft_hop != 4)) {
    g->tc_why = "iSTFT head: only n_fft 16 / hop 4 (the reference's gen_istft_* values)";
    return EVMI_OK;
  }
  std::vector<uint16_t> warena;
  std::vector<float> barena;
  auto reserve = [&](TcConv& t) {
    t.w_off = warena.size();
    warena.resize(warena.size() + (size_t)t.c_out * t.c_in * t.ks);
    t.bias_off = barena.size();
    barena.resize(barena.size() + t.c_out);
  };
  auto missing = [&](const std::string& what) {
    g->tc_why = "no MFMA instantiation for " + what;
    return EVMI_OK;
  };
  // conv_pre
  {
    TcConv& t = g->tc_pre;
    t.layer = "conv_pre";
    t.c_in = c.n_mels; t.c_out = c.upsample_initial_channel; t.ks = 7; t.dil = 1; t.pad = 3;
    t.launch = find_conv_tc(t.c_in, t.c_out, t.ks, 1);
    if (!t.launch) return missing("conv_pre c_in=" + std::to_string(t.c_in));
    reserve(t);
    relayout_conv(g->host_w["conv_pre.weight"].data(), t.c_out, t.c_in, t.ks, t.launch, warena, t.w_off);
    memcpy(&barena[t.bias_off], g->host_w["conv_pre.bias"].data(), sizeof(float) * t.c_out);
  }
  // upsamplers in polyphase form
  g->tc_ups.assign(c.num_upsamples, TcConv());
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    if (k != 2 * u) return missing("upsampler with kernel != 2*rate");
    const int cin = g->ch(i), cout = g->ch(i + 1);
    TcConv& t = g->tc_ups[i];
    t.layer = "ups." + std::to_string(i);
    t.c_in = cin; t.c_out = u * cout; t.ks = 2; t.dil = 1; t.pad = 1;
    t.launch = find_conv_tc(t.c_in, t.c_out, 2, 1);
    if (!t.launch) return missing(t.layer + " c_in=" + std::to_string(cin));
    reserve(t);
    // wc[phi*cout + co][tap][ci]: tap 0 reads x[q-1] with w[ci][co][phi+u], tap 1 reads x[q] with w[ci][co][phi]
    const std::vector<float>& w = g->host_w["ups." + std::to_string(i) + ".weight"];
    const std::vector<float>& bs = g->host_w["ups." + std::to_string(i) + ".bias"];
    std::vector<float> wc((size_t)t.c_out * cin * 2);
    for (int phi = 0; phi < u; ++phi)
      for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) {
          const size_t m = (size_t)phi * cout + co;
          wc[(m * cin + ci) * 2 + 0] = w[((size_t)ci * cout + co) * k + phi + u];
          wc[(m * cin + ci) * 2 + 1] = w[((size_t)ci * cout + co) * k + phi];
        }
    relayout_conv(wc.data(), t.c_out, cin, 2, t.launch, warena, t.w_off);
    for (int phi = 0; phi < u; ++phi)
      for (int co = 0; co < cout; ++co) barena[t.bias_off + (size_t)phi * cout + co] = bs[co];
  }
  // residual blocks
  g->tc_rb.clear();
  g->tc_pair.clear();
  g->tc_pair_w.clear();
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int cc = g->ch(i + 1);
    for (int j = 0; j < c.num_kernels; ++j) {
      const int n = i * c.num_kernels + j, k = c.resblock_kernel_sizes[j];
      std::vector<TcConv> convs;
      for (int m = 0; m < c.num_dilations[j]; ++m) {
        for (int which = 1; which <= (c.resblock_type == 1 ? 2 : 1); ++which) {
          TcConv t;
          t.layer = rb_name(c, n, which, m, "");
          t.layer.pop_back();
          t.c_in = cc; t.c_out = cc; t.ks = k;
          t.dil = which == 1 ? c.resblock_dilations[j][m] : 1;
          t.pad = t.dil * (k - 1) / 2;
          t.launch = find_conv_tc(cc, cc, k, t.dil);
          if (!t.launch) return missing(t.layer + " c=" + std::to_string(cc) + " k=" + std::to_string(k));
          reserve(t);
          relayout_conv(g->host_w[rb_name(c, n, which, m, "weight")].data(), cc, cc, k, t.launch, warena, t.w_off);
          memcpy(&barena[t.bias_off], g->host_w[rb_name(c, n, which, m, "bias")].data(), sizeof(float) * cc);
          convs.push_back(t);
        }
      }
      // fused pairs need the plain [tap][C][C] layout: stored as extra copies in the arena
      std::vector<const PairLaunch*> pairs;
      std::vector<size_t> pair_w;
      for (int m = 0; m < c.num_dilations[j]; ++m) {
        const PairLaunch* pl = nullptr;
        if (c.resblock_type == 1 && g->use_pairs) pl = find_resblock_pair(cc, k, c.resblock_dilations[j][m]);
        pairs.push_back(pl);
        size_t off = 0;
        if (pl) {
          off = warena.size();
          warena.resize(warena.size() + 2 * (size_t)cc * cc * k);
          for (int which = 1; which <= 2; ++which) {
            const float* w = g->host_w[rb_name(c, n, which, m, "weight")].data();
            uint16_t* dst = &warena[off + (size_t)(which - 1) * cc * cc * k];
            for (int mo = 0; mo < cc; ++mo)
              for (int ci = 0; ci < cc; ++ci)
                for (int jt = 0; jt < k; ++jt)
                  dst[((((size_t)(ci / pl->kc)) * k + jt) * cc + mo) * pl->kc + ci % pl->kc] =
                      f32_to_bf16_bits(w[((size_t)mo * cc + ci) * k + jt]);  // [chunk][tap][C][kc]
          }
        }
        pair_w.push_back(off);
      }
      g->tc_pair.push_back(pairs);
      g->tc_pair_w.push_back(pair_w);
      g->tc_rb.push_back(convs);
    }
  }
  if (c.istft_layer) {
    // conv_post of the iSTFT head: w[18][c][7] -> bf16 [7][32][c] (rows >= 18 zero), bias [32]
    const int cl = g->ch(c.num_upsamples);
    if (!(cl == 32 || cl == 64 || cl == 128)) return missing("istft head c_in=" + std::to_string(cl));
    const int co = c.istft_n_fft + 2;
    const std::vector<float>& w = g->host_w["conv_post.weight"];
    const std::vector<float>& bs = g->host_w["conv_post.bias"];
    g->istft_w_off = warena.size();
    warena.resize(warena.size() + (size_t)7 * 32 * cl, 0);
    for (int m = 0; m < co; ++m)
      for (int ci = 0; ci < cl; ++ci)
        for (int j = 0; j < 7; ++j)
          warena[g->istft_w_off + ((size_t)j * 32 + m) * cl + ci] = f32_to_bf16_bits(w[((size_t)m * cl + ci) * 7 + j]);
    g->istft_b_off = barena.size();
    barena.resize(barena.size() + 32, 0.f);
    for (int m = 0; m < co; ++m) barena[g->istft_b_off + m] = bs[m];
  } else
  // conv_post: w[1][c][7] -> [7][c] fp32
  {
    const int cl = g->ch(c.num_upsamples);
    if (!(cl == 16 || cl == 32 || cl == 64 || cl == 128)) return missing("conv_post c_in=" + std::to_string(cl));
    const std::vector<float>& w = g->host_w["conv_post.weight"];
    std::vector<float> wk((size_t)7 * cl);
    for (int cidx = 0; cidx < cl; ++cidx)
      for (int j = 0; j < 7; ++j) wk[(size_t)j * cl + cidx] = w[(size_t)cidx * 7 + j];
    int rc = g->post_w.ensure(wk.size() * sizeof(float));
    if (rc) return rc;
    EVMI_HIP_CHECK(hipMemcpy(g->post_w.p, wk.data(), wk.size() * sizeof(float), hipMemcpyHostToDevice));
    g->post_bias = g->host_w["conv_post.bias"][0];
  }
  int rc = g->tc_w_arena.ensure(warena.size() * 2 + 64);
  if (rc) return rc;
  rc = g->tc_bias_arena.ensure(barena.size() * 4 + 64);
  if (rc) return rc;
  EVMI_HIP_CHECK(hipMemcpy(g->tc_w_arena.p, warena.data(), warena.size() * 2, hipMemcpyHostToDevice));
  EVMI_HIP_CHECK(hipMemcpy(g->tc_bias_arena.p, barena.data(), barena.size() * 4, hipMemcpyHostToDevice));
  g->tc_ok = true;
  return EVMI_OK;
}

// ---- profiling helper ------------------------------------------------------------------------------
struct Recorder {
  evmi_launch_record* recs = nullptr;
  int cap = 0, n = 0;
  hipStream_t stream = nullptr;
  std::vector<hipEvent_t> ev;
  bool on() const { return recs != nullptr; }
  int begin() {
    if (!on()) return EVMI_OK;
    hipEvent_t e;
    EVMI_HIP_CHECK(hipEventCreate(&e));
    EVMI_HIP_CHECK(hipEventRecord(e, stream));
    ev.push_back(e);
    return EVMI_OK;
  }
  int end(const char* kernel, const std::string& layer, double flops, double bytes) {
    if (!on()) return EVMI_OK;
    hipEvent_t e;
    EVMI_HIP_CHECK(hipEventCreate(&e));
    EVMI_HIP_CHECK(hipEventRecord(e, stream));
    ev.push_back(e);
    if (n < cap) {
      evmi_launch_record& r = recs[n];
      memset(&r, 0, sizeof r);
      strncpy(r.kernel, kernel, sizeof(r.kernel) - 1);
      strncpy(r.layer, layer.c_str(), sizeof(r.layer) - 1);
      r.flops = flops;
      r.bytes = bytes;
    }
    ++n;
    return EVMI_OK;
  }
  int finish() {
    if (!on()) return EVMI_OK;
    EVMI_HIP_CHECK(hipStreamSynchronize(stream));
    for (int i = 0; i < n && i < cap; ++i) {
      float ms = 0.f;
      EVMI_HIP_CHECK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
      recs[i].ms = ms;
    }
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
    ev.clear();
    return EVMI_OK;
  }
};

#define EVMI_TRY(expr)          \
  do {                          \
    int _rc = (expr);           \
    if (_rc != EVMI_OK) return _rc; \
  } while (0)

static size_t stage_elems_max(const evmi_generator* g, int B, int T) {
  size_t mx = (size_t)B * T * g->cfg.upsample_initial_channel;
  size_t len = T;
  for (int i = 0; i < g->cfg.num_upsamples; ++i) {
    len *= g->cfg.upsample_rates[i];
    const size_t e = (size_t)B * (len + 1) * g->ch(i + 1);  // + 1 row: the iSTFT head's reflection pad
    if (e > mx) mx = e;
  }
  return mx;
}

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static int forward_tc(evmi_generator* g, const float* mel, float* wav, int B, int T, hipStream_t s, Recorder& rec) {
  const auto& c = g->cfg;
  if (!g->tc_ok) return fail(EVMI_ERR_UNSUPPORTED, "bf16 MFMA path unavailable: " + g->tc_why);
  const size_t se = align_up(stage_elems_max(g, B, T), 64);
  const size_t in_e = align_up((size_t)B * T * c.n_mels, 64);
  EVMI_TRY(g->ws.ensure((in_e + 5 * se) * 2));
  bf16_t* base = (bf16_t*)g->ws.p;
  bf16_t* X0 = base;
  bf16_t* buf[5];
  for (int i = 0; i < 5; ++i) buf[i] = base + in_e + (size_t)i * se;
  const bf16_t* warena = (const bf16_t*)g->tc_w_arena.p;
  const float* barena = (const float*)g->tc_bias_arena.p;

  auto run = [&](const TcConv& t, const bf16_t* x, int t_in, int n_rows, bf16_t* out, const bf16_t* res,
                 long long row_stride, long long shift, long long limit, float pre, float post, float scale,
                 int accumulate) -> int {
    ConvTcArgs a;
    a.x = x; a.w = warena + t.w_off; a.bias = barena + t.bias_off; a.res = res; a.out = out;
    a.t_in = t_in; a.n_rows = n_rows; a.c_out = t.c_out; a.dil = t.dil; a.pad = t.pad;
    a.x_batch_stride = (long long)t_in * t.c_in;
    a.out_batch_stride = limit;
    a.out_row_stride = row_stride; a.out_shift = shift; a.out_limit = limit;
    a.pre_slope = pre; a.post_slope = post; a.out_scale = scale; a.accumulate = accumulate;
    EVMI_TRY(rec.begin());
    EVMI_TRY(launch_conv_tc(t.launch, a, B, s));
    const double flops = 2.0 * B * (double)n_rows * t.c_out * t.ks * t.c_in;
    const double bytes = 2.0 * B * ((double)t_in * t.c_in + (double)limit * (1 + (res ? 1 : 0) + (accumulate ? 1 : 0))) +
                         2.0 * t.c_out * t.ks * t.c_in;
    return rec.end(t.launch->name, t.layer, flops, bytes);
  };

  EVMI_TRY(rec.begin());
  EVMI_TRY(launch_nct_f32_to_tc_bf16(mel, X0, B, c.n_mels, T, s));
  EVMI_TRY(rec.end("nct_f32_to_tc_bf16", "mel", 0.0, 6.0 * B * T * c.n_mels));

  // conv_pre -> A (already leaky-relu'd for the first upsampler)
  bf16_t* A = buf[4];
  EVMI_TRY(run(g->tc_pre, X0, T, T, A, nullptr, g->tc_pre.c_out, 0, (long long)T * g->tc_pre.c_out, 1.f,
               c.lrelu_slope, 1.f, 0));
  int len = T;
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i], p = (k - u) / 2;
    const int cout = g->ch(i + 1);
    const int len_out = len * u;
    bf16_t* U = buf[0];
    bf16_t* P[2] = {buf[1], buf[2]};
    bf16_t* T1 = buf[3];
    // A is read by the upsampler only; the stage output ACC reuses buf[4] after that
    EVMI_TRY(run(g->tc_ups[i], A, len, len + 1, U, nullptr, (long long)u * cout, -(long long)p * cout,
                 (long long)len_out * cout, 1.f, 1.f, 1.f, 0));
    bf16_t* ACC = buf[4];
    const bool last_stage = i == c.num_upsamples - 1;
    const long long lim = (long long)len_out * cout;
    for (int j = 0; j < c.num_kernels; ++j) {
      const std::vector<TcConv>& convs = g->tc_rb[i * c.num_kernels + j];
      const int nd = c.num_dilations[j];
      const bf16_t* cur = U;
      // the whole branch in one launch where its pairs are bound by the residual stream's round trips (resblock_branch_kernel.h)
      if (c.resblock_type == 1 && nd >= 1 && nd <= 3) {
        int dils[3] = {1, 1, 1};
        bool pairs_ok = true;
        for (int m = 0; m < nd; ++m) {
          dils[m] = convs[2 * m].dil;
          pairs_ok = pairs_ok && g->tc_pair[i * c.num_kernels + j][m] != nullptr && convs[2 * m + 1].dil == 1;
        }
        const BranchLaunch* bl = pairs_ok ? find_resblock_branch(cout, convs[0].ks, nd, dils) : nullptr;
        if (bl) {
          BranchArgs ba;
          ba.x = U;
          ba.out = ACC;
          for (int m = 0; m < 3; ++m) {
            const int mm = m < nd ? m : nd - 1;
            ba.w[2 * m] = warena + g->tc_pair_w[i * c.num_kernels + j][mm];
            ba.w[2 * m + 1] = ba.w[2 * m] + (size_t)cout * cout * convs[2 * mm].ks;
            ba.b[2 * m] = barena + convs[2 * mm].bias_off;
            ba.b[2 * m + 1] = barena + convs[2 * mm + 1].bias_off;
            ba.dil[m] = dils[mm];
          }
          ba.T = len_out;
          ba.np = nd;
          ba.slope = c.lrelu_slope;
          ba.post_slope = (j == c.num_kernels - 1) ? (last_stage ? c.post_lrelu_slope : c.lrelu_slope) : 1.f;
          ba.out_scale = 1.f / c.num_kernels;
          ba.accumulate = j > 0;
          EVMI_TRY(rec.begin());
          EVMI_TRY(launch_resblock_branch(bl, ba, B, g->n_cu, s));
          EVMI_TRY(rec.end(bl->name, convs[0].layer + "+" + std::to_string(2 * nd), 4.0 * nd * B * (double)len_out * cout * cout * convs[0].ks,
                           2.0 * B * (double)lim * (2 + ba.accumulate) + 4.0 * nd * cout * cout * convs[0].ks));
          continue;
        }
      }
      for (int m = 0; m < nd; ++m) {
        const bool last = m == nd - 1;
        bf16_t* nxt = last ? ACC : P[m & 1];
        const float scale = last ? 1.f / c.num_kernels : 1.f;
        const int accum = last && j > 0;
        const float post = (last && j == c.num_kernels - 1) ? (last_stage ? c.post_lrelu_slope : c.lrelu_slope) : 1.f;
        const PairLaunch* pl = g->tc_pair[i * c.num_kernels + j][m];
        if (c.resblock_type == 1 && pl) {
          PairArgs pa;
          pa.x = cur;
          pa.w1 = warena + g->tc_pair_w[i * c.num_kernels + j][m];
          pa.w2 = pa.w1 + (size_t)cout * cout * convs[2 * m].ks;
          pa.b1 = barena + convs[2 * m].bias_off;
          pa.b2 = barena + convs[2 * m + 1].bias_off;
          pa.out = nxt;
          pa.T = len_out;
          pa.dil1 = convs[2 * m].dil;
          pa.slope = c.lrelu_slope;
          pa.post_slope = post;
          pa.out_scale = scale;
          pa.accumulate = accum;
          pa.timeline = nullptr;
          EVMI_TRY(rec.begin());
          EVMI_TRY(launch_resblock_pair(pl, pa, B, g->n_cu, s));
          EVMI_TRY(rec.end(pl->name, convs[2 * m].layer + "+2", 4.0 * B * (double)len_out * cout * cout * convs[2 * m].ks,
                           2.0 * B * (double)lim * (2 + accum) + 4.0 * cout * cout * convs[2 * m].ks));
        } else if (c.resblock_type == 1) {
          EVMI_TRY(run(convs[2 * m], cur, len_out, len_out, T1, nullptr, cout, 0, lim, c.lrelu_slope,
                       c.lrelu_slope, 1.f, 0));
          EVMI_TRY(run(convs[2 * m + 1], T1, len_out, len_out, nxt, cur, cout, 0, lim, 1.f, post, scale, accum));
        } else {
          EVMI_TRY(run(convs[m], cur, len_out, len_out, nxt, cur, cout, 0, lim, c.lrelu_slope, post, scale, accum));
        }
        cur = nxt;
      }
    }
    A = ACC;
    len = len_out;
  }
  const int cl = g->ch(c.num_upsamples);
  EVMI_TRY(rec.begin());
  if (c.istft_layer) {
    EVMI_TRY(launch_istft_head(A, warena + g->istft_w_off, barena + g->istft_b_off, wav, B, len, cl, s));
    EVMI_TRY(rec.end("istft_head", "conv_post+istft", 2.0 * B * (double)(len + 1) * cl * 7 * (c.istft_n_fft + 2),
                     (double)B * len * (2.0 * cl + 4.0 * c.istft_hop)));
  } else {
    EVMI_TRY(launch_conv_post_tanh(A, (const float*)g->post_w.p, g->post_bias, wav, B, len, cl, 7, 1.f, s));
    EVMI_TRY(rec.end("conv_post_tanh", "conv_post", 2.0 * B * (double)len * cl * 7, (double)B * len * (2.0 * cl + 4)));
  }
  return EVMI_OK;
}

static int forward_f32(evmi_generator* g, const float* mel, float* wav, int B, int T, hipStream_t s, Recorder& rec) {
  const auto& c = g->cfg;
  if (c.istft_layer && (c.istft_n_fft != 16 || c.istft_hop != 4))
    return fail(EVMI_ERR_UNSUPPORTED, "iSTFT head: only n_fft 16 / hop 4 (the reference's gen_istft_* values)");
  const size_t se = align_up(stage_elems_max(g, B, T), 64);
  EVMI_TRY(g->ws.ensure(5 * se * 4));
  float* buf[5];
  for (int i = 0; i < 5; ++i) buf[i] = (float*)g->ws.p + (size_t)i * se;
  const float* arena = (const float*)g->f32_arena.p;
  auto W = [&](const std::string& n) { return arena + g->f32_off.at(n); };
  auto conv = [&](const std::string& layer, const float* x, const std::string& wn, const float* res, float* y,
                  int cin, int tin, int cout, int k, int pad, int dil, float pre, float scale, int accum) -> int {
    EVMI_TRY(rec.begin());
    EVMI_TRY(launch_conv1d_f32(x, W(wn + ".weight"), W(wn + ".bias"), res, y, B, cin, tin, cout, k, 1, pad, dil, 1,
                               pre, scale, accum, s));
    return rec.end("conv1d_f32_direct", layer, 2.0 * B * (double)tin * cout * cin * k,
                   4.0 * B * ((double)tin * cin + (double)tin * cout));
  };
  float* A = buf[4];
  EVMI_TRY(conv("conv_pre", mel, "conv_pre", nullptr, A, c.n_mels, T, c.upsample_initial_channel, 7, 3, 1, 1.f, 1.f, 0));
  int len = T;
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i], p = (k - u) / 2;
    const int cin = g->ch(i), cout = g->ch(i + 1);
    const int len_out = len * u;
    float* U = buf[0];
    float* P[2] = {buf[1], buf[2]};
    float* T1 = buf[3];
    const std::string un = "ups." + std::to_string(i);
    EVMI_TRY(rec.begin());
    EVMI_TRY(launch_conv_transpose1d_f32(A, W(un + ".weight"), W(un + ".bias"), U, B, cin, len, cout, k, u, p,
                                         c.lrelu_slope, s));
    EVMI_TRY(rec.end("conv_transpose1d_f32_direct", un, 2.0 * B * (double)len_out * cout * cin * k / u,
                     4.0 * B * ((double)len * cin + (double)len_out * cout)));
    float* ACC = buf[4];
    for (int j = 0; j < c.num_kernels; ++j) {
      const int n = i * c.num_kernels + j, kk = c.resblock_kernel_sizes[j], nd = c.num_dilations[j];
      const float* cur = U;
      for (int m = 0; m < nd; ++m) {
        const bool last = m == nd - 1;
        float* nxt = last ? ACC : P[m & 1];
        const float scale = last ? 1.f / c.num_kernels : 1.f;
        const int accum = last && j > 0;
        const int d = c.resblock_dilations[j][m];
        if (c.resblock_type == 1) {
          std::string n1 = rb_name(c, n, 1, m, ""), n2 = rb_name(c, n, 2, m, "");
          n1.pop_back();
          n2.pop_back();
          EVMI_TRY(conv(n1, cur, n1, nullptr, T1, cout, len_out, cout, kk, d * (kk - 1) / 2, d, c.lrelu_slope, 1.f, 0));
          EVMI_TRY(conv(n2, T1, n2, cur, nxt, cout, len_out, cout, kk, (kk - 1) / 2, 1, c.lrelu_slope, scale, accum));
        } else {
          std::string n1 = rb_name(c, n, 1, m, "");
          n1.pop_back();
          EVMI_TRY(conv(n1, cur, n1, cur, nxt, cout, len_out, cout, kk, d * (kk - 1) / 2, d, c.lrelu_slope, scale, accum));
        }
        cur = nxt;
      }
    }
    A = ACC;
    len = len_out;
  }
  const int cl = g->ch(c.num_upsamples);
  if (c.istft_layer) {
    float* XP = buf[0];  // [B][cl][len + 1]: leaky-relu then reflection pad (1, 0)
    float* Z = buf[1];   // [B][n_fft + 2][len + 1]
    EVMI_TRY(rec.begin());
    EVMI_TRY(launch_reflect_pad_left1_f32(A, XP, (long long)B * cl, len, c.post_lrelu_slope, s));
    EVMI_TRY(rec.end("reflect_pad_left1_f32", "reflection_pad", 0.0, 8.0 * B * cl * len));
    EVMI_TRY(conv("conv_post", XP, "conv_post", nullptr, Z, cl, len + 1, c.istft_n_fft + 2, 7, 3, 1, 1.f, 1.f, 0));
    EVMI_TRY(rec.begin());
    EVMI_TRY(launch_istft_f32(Z, wav, B, len + 1, s));
    return rec.end("istft_f32", "istft", 0.0, 4.0 * B * len * (c.istft_n_fft + 2 + c.istft_hop));
  }
  EVMI_TRY(conv("conv_post", A, "conv_post", nullptr, wav, cl, len, 1, 7, 3, 1, c.post_lrelu_slope, 1.f, 0));
  EVMI_TRY(rec.begin());
  EVMI_TRY(launch_tanh_f32(wav, (long long)B * len, s));
  EVMI_TRY(rec.end("tanh_f32", "tanh", 0.0, 8.0 * B * len));
  return EVMI_OK;
}

// ---- C ABI -------------------------------------------------------------------------------------------
extern "C" {

int evmi_abi_version(void) { return EVMI_ABI_VERSION; }
const char* evmi_last_error(void) { return last_error_cstr(); }

int evmi_device_info(int device, char* name, int name_len, int* compute_units, int64_t* hbm_bytes) {
  hipDeviceProp_t prop;
  EVMI_HIP_CHECK(hipGetDeviceProperties(&prop, device));
  if (name && name_len > 0) {
    snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
  return EVMI_OK;
}

int evmi_generator_create(const evmi_generator_config* cfg, int device, evmi_generator** out) {
  if (!cfg || !out) return fail(EVMI_ERR_INVALID_ARG, "generator_create: null argument");
  EVMI_TRY(validate_cfg(*cfg));
  evmi_generator* g = new evmi_generator();
  g->cfg = *cfg;
  g->device = device;
  build_specs(g);
  *out = g;
  return EVMI_OK;
}

void evmi_generator_destroy(evmi_generator* g) { delete g; }

int evmi_generator_num_weights(const evmi_generator* g) { return g ? (int)g->specs.size() : 0; }

int evmi_generator_weight_info(const evmi_generator* g, int i, char* name, int name_len, int64_t* numel) {
  if (!g || i < 0 || i >= (int)g->specs.size()) return fail(EVMI_ERR_INVALID_ARG, "weight_info: index");
  if (name && name_len > 0) snprintf(name, name_len, "%s", g->specs[i].name.c_str());
  if (numel) *numel = g->specs[i].numel;
  return EVMI_OK;
}

int evmi_generator_set_weight(evmi_generator* g, const char* name, const float* data, int64_t numel) {
  if (!g || !name || !data) return fail(EVMI_ERR_INVALID_ARG, "set_weight: null argument");
  for (const WeightSpec& s : g->specs) {
    if (s.name == name) {
      if (s.numel != numel)
        return fail(EVMI_ERR_INVALID_ARG, std::string("set_weight: ") + name + " expects " +
                                              std::to_string(s.numel) + " elements, got " + std::to_string(numel));
      g->host_w[name].assign(data, data + numel);
      g->finalized = false;
      return EVMI_OK;
    }
  }
  return fail(EVMI_ERR_INVALID_ARG, std::string("set_weight: unknown tensor '") + name + "'");
}

int evmi_generator_finalize(evmi_generator* g) {
  if (!g) return fail(EVMI_ERR_INVALID_ARG, "finalize: null");
  EVMI_HIP_CHECK(hipSetDevice(g->device));
  size_t total = 0;
  for (const WeightSpec& s : g->specs) {
    if (!g->host_w.count(s.name)) return fail(EVMI_ERR_NOT_READY, "finalize: missing weight '" + s.name + "'");
    g->f32_off[s.name] = total;
    total += align_up((size_t)s.numel, 4);
  }
  EVMI_TRY(g->f32_arena.ensure(total * 4));
  for (const WeightSpec& s : g->specs)
    EVMI_HIP_CHECK(hipMemcpy((float*)g->f32_arena.p + g->f32_off[s.name], g->host_w[s.name].data(),
                             (size_t)s.numel * 4, hipMemcpyHostToDevice));
  {
    hipDeviceProp_t prop;
    EVMI_HIP_CHECK(hipGetDeviceProperties(&prop, g->device));
    g->n_cu = prop.multiProcessorCount;
    const char* e = getenv("EVMI_NO_FUSED_PAIRS");  // tuning switch: fall back to two launches per pair
    g->use_pairs = !(e && e[0] == '1');
  }
  EVMI_TRY(prepare_tc(g));
  g->finalized = true;
  return EVMI_OK;
}

int evmi_generator_hop(const evmi_generator* g) { return g ? g->hop() : 0; }

int64_t evmi_generator_workspace_bytes(const evmi_generator* g, int B, int T, int precision) {
  if (!g || B <= 0 || T <= 0) return 0;
  const size_t se = align_up(stage_elems_max(g, B, T), 64);
  if (precision == EVMI_PREC_BF16) return (int64_t)((align_up((size_t)B * T * g->cfg.n_mels, 64) + 5 * se) * 2);
  return (int64_t)(5 * se * 4);
}

double evmi_generator_macs_per_sample(const evmi_generator* g) {
  if (!g) return 0.0;
  const auto& c = g->cfg;
  double per_frame = (double)c.n_mels * c.upsample_initial_channel * 7;
  double len = 1;
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int u = c.upsample_rates[i], k = c.upsample_kernel_sizes[i];
    const int cin = g->ch(i), cout = g->ch(i + 1);
    per_frame += len * cin * cout * k;  // each input position feeds k outputs per (ci, co)
    len *= u;
    for (int j = 0; j < c.num_kernels; ++j)
      per_frame += len * (double)cout * cout * c.resblock_kernel_sizes[j] * c.num_dilations[j] *
                   (c.resblock_type == 1 ? 2 : 1);
  }
  const int post_out = c.istft_layer ? c.istft_n_fft + 2 : 1;
  per_frame += len * g->ch(c.num_upsamples) * post_out * 7;
  return per_frame / (double)g->hop();
}

static int forward_common(evmi_generator* g, const float* mel, float* wav, int B, int T, int precision, void* stream,
                          evmi_launch_record* recs, int cap, int* n_out) {
  if (!g || !mel || !wav) return fail(EVMI_ERR_INVALID_ARG, "forward: null argument");
  if (!g->finalized) return fail(EVMI_ERR_NOT_READY, "forward: call evmi_generator_finalize first");
  if (B <= 0 || T <= 0) return fail(EVMI_ERR_INVALID_ARG, "forward: B and T must be positive");
  Recorder rec;
  rec.recs = recs;
  rec.cap = cap;
  rec.stream = (hipStream_t)stream;
  int rc;
  if (precision == EVMI_PREC_BF16)
    rc = forward_tc(g, mel, wav, B, T, (hipStream_t)stream, rec);
  else if (precision == EVMI_PREC_F32)
    rc = forward_f32(g, mel, wav, B, T, (hipStream_t)stream, rec);
  else
    return fail(EVMI_ERR_INVALID_ARG, "forward: unknown precision");
  if (rc) return rc;
  EVMI_TRY(rec.finish());
  if (n_out) *n_out = rec.n;
  return EVMI_OK;
}

int evmi_generator_forward(evmi_generator* g, const float* mel, float* wav, int B, int T, int precision, void* stream) {
  return forward_common(g, mel, wav, B, T, precision, stream, nullptr, 0, nullptr);
}

int evmi_generator_forward_profiled(evmi_generator* g, const float* mel, float* wav, int B, int T, int precision,
                                    void* stream, evmi_launch_record* records, int cap, int* n_out) {
  if (!records || cap <= 0) return fail(EVMI_ERR_INVALID_ARG, "forward_profiled: records/cap");
  return forward_common(g, mel, wav, B, T, precision, stream, records, cap, n_out);
}

}  // extern "C"
