// fp32 implicit-GEMM 1-D convolution on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// fmaf chains, 157 TFLOP/s class) for the channel-major training layout x[c][b][t].
//
//   y[co][b][to*os + oo] (+)= bias[co] + sum_{ci in group} sum_{j<k} w[co][ci][j] * x[ci][b][to*s + j*d - p]
//
// GEMM per workgroup: D[BM out-channels x BN output positions of one batch item] over K = (ci, j) in the
// order the weights are stored (w[co] is a contiguous K-slice: no weight re-layout).  Channels are staged CB
// at a time: their input span ((BN-1)*s + (k-1)*d + 1 samples) sits in LDS once and every tap reads it at
// an offset; the weights of the chunk stream through LDS in 32-deep K steps (double-buffered).  With K = 2
// per MFMA the operand traffic is one float per lane per operand per 64-cycle instruction: the kernel is
// bound by the fp32 MFMA rate, not by LDS or HBM.  Any stride / dilation / groups; `os`, `oo` place the
// outputs on a strided grid so that the gradient of a strided convolution runs as `stride` polyphase
// stride-1 convolutions through this same kernel.
#include "common.h"

namespace evmi {

struct ConvF32Args {
  const float* x;     // [c_in][B][t_in]
  const float* w;     // [c_out][cin_g][k]  (or any array with the same [co][K] indexing)
  const float* bias;  // [c_out] or nullptr
  float* y;           // [c_out][B][t_out_total]
  int B, t_in, t_out_total;
  int n_out;          // output positions computed per (co, b): to in [0, n_out)
  int cin_g, cout_g, k, stride, dil, pad;
  int out_stride, out_offset;  // y index = to * out_stride + out_offset
  int accumulate;              // y += instead of y =
  int mtiles_per_group;
};

constexpr int F32_CB = 16;     // input channels staged per chunk
constexpr int F32_KSTEP = 32;  // K depth of one weight tile
constexpr int F32_AS = F32_KSTEP + 1;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(WM * WN * 64) void conv_cbt_f32_mfma_kernel(ConvF32Args a) {
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int MT = BM / (WM * 32), NT = BN / (WN * 32);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* As = reinterpret_cast<float*>(smem);  // [2][BM][F32_AS]
  float* Xs = As + 2 * BM * F32_AS;            // [F32_CB + 3][span_pad]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int g = blockIdx.z / a.mtiles_per_group, mt_idx = blockIdx.z % a.mtiles_per_group;
  const int co0 = g * a.cout_g + mt_idx * BM;
  const int b = blockIdx.y;
  const int to0 = blockIdx.x * BN;
  const int k = a.k, s = a.stride, d = a.dil;
  const int span = (BN - 1) * s + (k - 1) * d + 1;
  const int span_pad = span | 1;  // odd row stride
  const int ti0 = to0 * s - a.pad;
  const int Kg = a.cin_g * k;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // rows of this M tile that exist (the last tile of a group may be partial)
  const int m_valid = min(BM, a.cout_g - mt_idx * BM);
  const float* wbase = a.w + (long long)co0 * Kg;

  const int kh = lane >> 5, ln = lane & 31;
  for (int c0 = 0; c0 < a.cin_g; c0 += F32_CB) {
    const int cb = min(F32_CB, a.cin_g - c0);
    __syncthreads();  // previous chunk fully consumed
    // ---- stage the input span of cb channels (+ zero rows that padded K indices may touch) ----
    for (int v = tid; v < (F32_CB + 3) * span; v += NTHREADS) {
      const int r = v / span, i = v - r * span;
      const int ti = ti0 + i;
      float val = 0.f;
      if (r < cb && ti >= 0 && ti < a.t_in) val = a.x[((long long)(g * a.cin_g + c0 + r) * a.B + b) * a.t_in + ti];
      Xs[r * span_pad + i] = val;
    }
    const int kc = cb * k;                                   // K indices of this chunk
    const int nsteps = (kc + F32_KSTEP - 1) / F32_KSTEP;
    const long long kbase = (long long)c0 * k;               // offset of the chunk inside a weight row
    auto load_a = [&](int step, int buf) {
      float* dst = As + buf * BM * F32_AS;
      for (int v = tid; v < BM * F32_KSTEP; v += NTHREADS) {
        const int m = v / F32_KSTEP, kk = v - m * F32_KSTEP;
        const int kl = step * F32_KSTEP + kk;
        float val = 0.f;
        if (m < m_valid && kl < kc) val = wbase[(long long)m * Kg + kbase + kl];
        dst[m * F32_AS + kk] = val;
      }
    };
    load_a(0, 0);
    // (ci_l, j) of this lane's K index, advanced by 2 per MFMA
    int ci_l = kh / k, j = kh - ci_l * k;
    for (int step = 0; step < nsteps; ++step) {
      __syncthreads();  // tile `step` (and, for step 0, the input span) visible; previous tile consumed
      if (step + 1 < nsteps) load_a(step + 1, (step + 1) & 1);
      const float* Ab = As + (step & 1) * BM * F32_AS;
#pragma unroll 4
      for (int q = 0; q < F32_KSTEP / 2; ++q) {
        float af[MT], bf[NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) af[mt] = Ab[((wm * MT + mt) * 32 + ln) * F32_AS + 2 * q + kh];
        // K indices past the chunk (zero weights) may point past the staged channels: park them on a zero row
        const float* xr = Xs + min(ci_l, F32_CB + 2) * span_pad + j * d;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bf[nt] = xr[((wn * NT + nt) * 32 + ln) * s];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mt], bf[nt], acc[mt][nt], 0, 0, 0);
        // advance the K index by 2: (ci_l, j) <- divmod(ci_l * k + j + 2, k)
        j += 2;
        if (k == 1) { ci_l += 2; j = 0; }
        else if (j >= k) { j -= k; ci_l += 1; }
      }
    }
  }

  // ---- epilogue: D layout: lane column = output position, registers = output channels ----
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int to = to0 + (wn * NT + nt) * 32 + ln;
      if (to >= a.n_out) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m >= m_valid) continue;
        const int co = co0 + m;
        float v = acc[mt][nt][r];
        if (a.bias) v += a.bias[co];
        float* dst = a.y + ((long long)co * a.B + b) * a.t_out_total + (long long)to * a.out_stride + a.out_offset;
        *dst = a.accumulate ? *dst + v : v;
      }
    }
}

static int launch_cfg(const ConvF32Args& a, int& bm) {
  bm = a.cout_g >= 96 ? 128 : (a.cout_g >= 48 ? 64 : 32);
  return (a.cout_g + bm - 1) / bm;
}

int launch_conv_cbt_f32_mfma(ConvF32Args a, int groups, hipStream_t stream) {
  if (a.cin_g <= 0 || a.cout_g <= 0 || a.k <= 0 || a.stride <= 0 || a.dil <= 0 || a.n_out <= 0 || a.B <= 0)
    return fail(EVMI_ERR_INVALID_ARG, "conv_cbt_f32_mfma: bad shape");
  int bm;
  a.mtiles_per_group = launch_cfg(a, bm);
  const int bn = bm == 128 ? 128 : (bm == 64 ? 128 : 256);
  const int span = (bn - 1) * a.stride + (a.k - 1) * a.dil + 1;
  const size_t lds = ((size_t)2 * bm * F32_AS + (size_t)(F32_CB + 3) * (span | 1)) * sizeof(float);
  if (lds > 160 * 1024) return fail(EVMI_ERR_UNSUPPORTED, "conv_cbt_f32_mfma: input span too large for LDS");
  dim3 grid((a.n_out + bn - 1) / bn, a.B, groups * a.mtiles_per_group);
  if (grid.y > 65535 || grid.z > 65535) return fail(EVMI_ERR_UNSUPPORTED, "conv_cbt_f32_mfma: grid limits");
  static thread_local size_t configured[3] = {0, 0, 0};
#define EVMI_F32_LAUNCH(BM, BN, WM, WN, IDX)                                                                       \
  {                                                                                                                \
    if (lds > configured[IDX]) {                                                                                   \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)conv_cbt_f32_mfma_kernel<BM, BN, WM, WN>,                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
      configured[IDX] = lds;                                                                                       \
    }                                                                                                              \
    hipLaunchKernelGGL((conv_cbt_f32_mfma_kernel<BM, BN, WM, WN>), grid, dim3(WM * WN * 64), lds, stream, a);      \
  }
  if (bm == 128) EVMI_F32_LAUNCH(128, 128, 2, 2, 0)
  else if (bm == 64) EVMI_F32_LAUNCH(64, 128, 1, 4, 1)
  else EVMI_F32_LAUNCH(32, 256, 1, 4, 2)
#undef EVMI_F32_LAUNCH
  EVMI_LAUNCH_CHECK("conv_cbt_f32_mfma");
  return EVMI_OK;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

/* y[co][b][to*out_stride + out_offset] (+)= bias[co] + conv(x, w)[co][b][to] for to < n_out; x [c_in][B][t_in],
 * w [c_out][c_in/groups][k], y [c_out][B][t_out_total].  out_stride = 1, out_offset = 0, n_out = t_out_total is the
 * plain convolution. */
int evmi_conv1d_cbt_f32(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int B, int c_in,
                        int t_in, int c_out, int t_out_total, int n_out, int k, int stride, int pad, int dil, int groups,
                        int out_stride, int out_offset, int accumulate, void* stream) {
  if (!x_dev || !w_dev || !y_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_f32: null pointer");
  if (groups <= 0 || c_in % groups || c_out % groups) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_f32: groups");
  ConvF32Args a;
  a.x = x_dev; a.w = w_dev; a.bias = bias_dev; a.y = y_dev;
  a.B = B; a.t_in = t_in; a.t_out_total = t_out_total; a.n_out = n_out;
  a.cin_g = c_in / groups; a.cout_g = c_out / groups; a.k = k; a.stride = stride; a.dil = dil; a.pad = pad;
  a.out_stride = out_stride; a.out_offset = out_offset; a.accumulate = accumulate; a.mtiles_per_group = 1;
  return launch_conv_cbt_f32_mfma(a, groups, (hipStream_t)stream);
}

}  // extern "C"
