// FastSpeech2 feature-prediction forward path in the channel-major layout x[c][b][t], fp32.
// Dense layers (Linear, pointwise and postnet convolutions) run on the fp32 matrix-core implicit GEMM
// (conv_cbt_f32_mfma.hip, k = 1 is a plain GEMM); this file holds everything else:
//   * attention_cbt_kernel  fused multi-head self-attention: S^T = K.Q^T and O^T = V^T.P^T on v_mfma_f32_32x32x2_f32
//                           with the online softmax between them.  Working on the transposed problem makes the
//                           accumulator layout of S^T (lane = query, registers = keys) exactly the B-operand layout
//                           of the second product: P never leaves the registers.  Key padding mask from lengths.
//   * layernorm_cbt_kernel  LayerNorm over the channel axis of every column
//   * dwconv_cbt_kernel     depthwise convolution (+ folded BatchNorm) with SiLU / ReLU epilogue
//   * embedding + FastPitch positional sinusoid, variance bucketise + embedding add, durations from log-durations,
//     the length regulator as a gather in this layout, column masking.
#include <cmath>

#include "common.h"
#include "evmi.h"

namespace evmi {

// ---- embedding + positional sinusoid --------------------------------------------------------------------------------
// out[c][b][l] = l < len[b] ? table[ids[b][l]][c] + pe(l, c) : 0 ;  pe = cat(sin(l * inv_freq), cos(l * inv_freq))
__global__ void fs2_embed_kernel(const int* __restrict__ ids, const int* __restrict__ lens, const float* __restrict__ table,
                                 const float* __restrict__ inv_freq, float* __restrict__ out, int B, int L, int D) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)D * B * L) return;
  const int l = (int)(i % L), b = (int)((i / L) % B), c = (int)(i / ((long long)L * B));
  float v = 0.f;
  if (l < lens[b]) {
    v = table[(long long)ids[b * L + l] * D + c];
    if (inv_freq) {  // NULL: the bare symbol embedding (the aligner's keys)
      const int h = D / 2;
      const float ang = (float)l * inv_freq[c < h ? c : c - h];
      v += c < h ? sinf(ang) : cosf(ang);
    }
  }
  out[i] = v;
}

// x[c][b][t] = t < len[b] ? x + pe(t, c) : 0
__global__ void fs2_add_posemb_kernel(float* __restrict__ x, const int* __restrict__ lens, const float* __restrict__ inv_freq,
                                      int B, int T, int D) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)D * B * T) return;
  const int t = (int)(i % T), b = (int)((i / T) % B), c = (int)(i / ((long long)T * B));
  float v = 0.f;
  if (t < lens[b]) {
    const int h = D / 2;
    const float ang = (float)t * inv_freq[c < h ? c : c - h];
    v = x[i] + (c < h ? sinf(ang) : cosf(ang));
  }
  x[i] = v;
}

// x[c][b][t] = t < len[b] ? x : 0
__global__ void mask_cols_kernel(float* __restrict__ x, const int* __restrict__ lens, int B, int T, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)C * B * T) return;
  const int t = (int)(i % T), b = (int)((i / T) % B);
  if (t >= lens[b]) x[i] = 0.f;
}

// ---- LayerNorm over channels ---------------------------------------------------------------------------------------
// Workgroup = 64 columns x 4 channel slices: every thread keeps its slice of one column in registers (one global
// read, one write), the four partial sums / squared deviations meet in LDS.  Columns are contiguous across lanes.
template <int CPT>  // channels per thread, C <= 4 * CPT
__global__ __launch_bounds__(256) void layernorm_cbt_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ y, int C,
                                                           long long N, float eps) {
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long long n = (long long)blockIdx.x * 64 + lane;
  const bool live = n < N;
  const int per = (C + 3) >> 2, c0 = slice * per, c1 = min(C, c0 + per);
  float v[CPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = c0 + i;
    v[i] = (live && c < c1) ? x[(long long)c * N + n] : 0.f;
    s += v[i];
  }
  red[0][slice][lane] = s;
  __syncthreads();
  const float mean = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const float d = (c0 + i < c1) ? v[i] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  red[1][slice][lane] = q;
  __syncthreads();
  const float var = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / (float)C;
  const float rstd = 1.f / sqrtf(var + eps);
  if (!live) return;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = c0 + i;
    if (c < c1) y[(long long)c * N + n] = (v[i] - mean) * rstd * gamma[c] + beta[c];
  }
}

// ---- depthwise conv: y[c][b][t] = act(bias[c] + sum_j w[c][j] * x[c][b][t + j - pad]) ---------------------------------
__global__ void dwconv_cbt_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ y, int C, int B, int T, int k, int pad, int act) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)C * B * T) return;
  const int t = (int)(i % T);
  const int c = (int)(i / ((long long)T * B));
  const float* xr = x + (i - t);
  float v = bias ? bias[c] : 0.f;
  for (int j = 0; j < k; ++j) {
    const int ti = t + j - pad;
    if (ti >= 0 && ti < T) v = fmaf(w[c * k + j], xr[ti], v);
  }
  if (act == 1) v = v / (1.f + expf(-v));
  else if (act == 2) v = fmaxf(v, 0.f);
  y[i] = v;
}

// ---- variance adaptor pieces ---------------------------------------------------------------------------------------
// x[c][b][l] += table[bucket(values[b][l] * control)][c]; bucket = #bins strictly below the value (torch.bucketize)
__global__ void fs2_bucket_embed_add_kernel(float* __restrict__ x, const float* __restrict__ values, const float* __restrict__ bins,
                                            const float* __restrict__ table, int n_bins, int B, int L, int D, float control) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)D * B * L) return;
  const int c = (int)(i / ((long long)L * B));
  const float v = values[i % ((long long)L * B)] * control;
  int lo = 0, hi = n_bins - 1;  // bins has n_bins - 1 boundaries
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (bins[mid] < v) lo = mid + 1; else hi = mid;
  }
  x[i] += table[(long long)lo * D + c];
}

// dur[b][l] = l < len[b] ? max(0, rint(exp(log_d) - 1) * control) : 0  (int32), the reference's inference rule
__global__ void fs2_durations_kernel(const float* __restrict__ log_d, const int* __restrict__ lens, int* __restrict__ dur, int B,
                                     int L, float control) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  const int l = i % L, b = i / L;
  float d = rintf(expf(log_d[i]) - 1.f) * control;
  d = fmaxf(d, 0.f);
  dur[i] = l < lens[b] ? (int)d : 0;
}

// length regulator in this layout: out[c][b][t] = t < total[b] ? x[c][b][token(b, t)] : 0, token by binary search
// in the inclusive prefix sums cum[b][l] of the durations
__global__ void length_regulate_cbt_kernel(const float* __restrict__ x, const int* __restrict__ cum, float* __restrict__ out, int C,
                                           int B, int L, int T) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)C * B * T) return;
  const int t = (int)(i % T), b = (int)((i / T) % B), c = (int)(i / ((long long)T * B));
  const int* cb = cum + b * L;
  float v = 0.f;
  if (t < cb[L - 1]) {
    int lo = 0, hi = L - 1;  // first l with cum[l] > t
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cb[mid] > t) hi = mid; else lo = mid + 1;
    }
    v = x[((long long)c * B + b) * L + lo];
  }
  out[i] = v;
}

// ---- fused self-attention ------------------------------------------------------------------------------------------
// qkv [3*D][B][T] (q rows, then k rows, then v rows; head h owns channels h*DH .. h*DH+DH-1), lens [B] -> out [D][B][T].
// grid (ceil(T / 128), H, B), 256 threads: every wave owns 32 queries and walks the key tiles of 32.
template <int DH>
__global__ __launch_bounds__(256) void attention_cbt_kernel(const float* __restrict__ qkv, const int* __restrict__ lens,
                                                           float* __restrict__ out, int B, int T, int D, float scale) {
  constexpr int VS = 33;  // V rows are read across channels: odd stride
  __shared__ float Ks[DH * 32];
  __shared__ float Vs[DH * VS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ln = lane & 31, kh = lane >> 5;
  const int h = blockIdx.y, b = blockIdx.z;
  const int len = min(lens[b], T);
  const long long N = (long long)B * T;
  const float* q = qkv + ((long long)(h * DH) * B + b) * T;
  const float* kg = qkv + ((long long)(D + h * DH) * B + b) * T;
  const float* vg = qkv + ((long long)(2 * D + h * DH) * B + b) * T;
  const int tq = blockIdx.x * 128 + wave * 32 + ln;
  const bool qlive = tq < T;
  // Q^T as the B operand of S^T = K.Q^T: lane (query, half) holds Q[query][2s + half], pre-scaled
  float qreg[DH / 2];
#pragma unroll
  for (int s = 0; s < DH / 2; ++s) {
    const float qv = q[(long long)(2 * s + kh) * N + min(tq, T - 1)];  // unconditional (clamped) load, then the select
    qreg[s] = qlive ? qv * scale : 0.f;
  }
  f32x16 acc[DH / 32];
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  for (int k0 = 0; k0 < len; k0 += 32) {
    __syncthreads();  // previous tile consumed
    {
      // the key / value tile: every element of the thread is requested before the first is stored, through unconditional
      // loads from clamped positions (a load under a per-lane condition drains the memory counter: 32 dependent round trips
      // per tile at DH = 128)
      constexpr int NV = DH * 32 / 256;
      float kr[NV], vr[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256, d = v >> 5, kk = v & 31;
        const long long o = (long long)d * N + min(k0 + kk, T - 1);
        kr[i] = kg[o];
        vr[i] = vg[o];
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = tid + i * 256, d = v >> 5, kk = v & 31;
        const bool in = k0 + kk < T;
        Ks[d * 32 + kk] = in ? kr[i] : 0.f;
        Vs[d * VS + kk] = in ? vr[i] : 0.f;
      }
    }
    __syncthreads();
    // S^T tile: rows = keys (A operand lane (key, half): K[key][2s + half]), columns = queries
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
    for (int s = 0; s < DH / 2; ++s) st = __builtin_amdgcn_mfma_f32_32x32x2f32(Ks[(2 * s + kh) * 32 + ln], qreg[s], st, 0, 0, 0);
    // online softmax over the keys of this lane's query: registers x the two halves
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
      if (key >= len) st[r] = -INFINITY;
      mx = fmaxf(mx, st[r]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);  // finite: every processed tile has a valid key
    const float corr = expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      st[r] = expf(st[r] - m_new);
      ps += st[r];
    }
    ps += __shfl_xor(ps, 32, 64);
    l_run = l_run * corr + ps;
    m_run = m_new;
    // O^T += V^T . P^T: step r contracts key (r&3)+8(r>>2) [half 0] and that key + 4 [half 1]: P^T is st[r] as it lies
#pragma unroll
    for (int i = 0; i < DH / 32; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] *= corr;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(i * 32 + ln) * VS + (r & 3) + 8 * (r >> 2) + 4 * kh], st[r], acc[i], 0, 0, 0);
    }
  }
  if (!qlive) return;
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;  // an item of length 0 has no keys: its rows are zeros, not NaN
  float* o = out + ((long long)(h * DH) * B + b) * T + tq;
#pragma unroll
  for (int i = 0; i < DH / 32; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(long long)(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh) * N] = acc[i][r] * inv;
}

// (the bf16-operand attention of precision="bf16" runs on attention_train.hip's forward kernel: launch_mha_fwd_bf16_plain)

static dim3 grid1d(long long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace evmi

using namespace evmi;

#define EVMI_NONNULL(cond, what) \
  if (!(cond)) return fail(EVMI_ERR_INVALID_ARG, std::string(what) + ": null pointer")

extern "C" {

int evmi_fs2_embed_f32(const int* ids_dev, const int* lens_dev, const float* table_dev, const float* inv_freq_dev, float* out_dev,
                       int B, int L, int D, void* stream) {
  EVMI_NONNULL(ids_dev && lens_dev && table_dev && out_dev, "fs2_embed");  // inv_freq NULL: no positional term
  if (B <= 0 || L <= 0 || D <= 0 || (D & 1)) return fail(EVMI_ERR_INVALID_ARG, "fs2_embed: shape");
  hipLaunchKernelGGL(fs2_embed_kernel, grid1d((long long)D * B * L), dim3(256), 0, (hipStream_t)stream, ids_dev, lens_dev, table_dev,
                     inv_freq_dev, out_dev, B, L, D);
  EVMI_LAUNCH_CHECK("fs2_embed");
  return EVMI_OK;
}

int evmi_fs2_add_posemb_f32(float* x_dev, const int* lens_dev, const float* inv_freq_dev, int B, int T, int D, void* stream) {
  EVMI_NONNULL(x_dev && lens_dev && inv_freq_dev, "fs2_add_posemb");
  hipLaunchKernelGGL(fs2_add_posemb_kernel, grid1d((long long)D * B * T), dim3(256), 0, (hipStream_t)stream, x_dev, lens_dev,
                     inv_freq_dev, B, T, D);
  EVMI_LAUNCH_CHECK("fs2_add_posemb");
  return EVMI_OK;
}

int evmi_mask_cols_f32(float* x_dev, const int* lens_dev, int C, int B, int T, void* stream) {
  EVMI_NONNULL(x_dev && lens_dev, "mask_cols");
  hipLaunchKernelGGL(mask_cols_kernel, grid1d((long long)C * B * T), dim3(256), 0, (hipStream_t)stream, x_dev, lens_dev, B, T, C);
  EVMI_LAUNCH_CHECK("mask_cols");
  return EVMI_OK;
}

int evmi_layernorm_cbt_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev, int C, long long n_cols,
                           float eps, void* stream) {
  EVMI_NONNULL(x_dev && gamma_dev && beta_dev && y_dev, "layernorm_cbt");
  if (C <= 0 || n_cols <= 0) return fail(EVMI_ERR_INVALID_ARG, "layernorm_cbt: shape");
  const dim3 grid((unsigned)((n_cols + 63) / 64));
  hipStream_t s = (hipStream_t)stream;
  if (C <= 64) hipLaunchKernelGGL(layernorm_cbt_kernel<16>, grid, dim3(256), 0, s, x_dev, gamma_dev, beta_dev, y_dev, C, n_cols, eps);
  else if (C <= 256) hipLaunchKernelGGL(layernorm_cbt_kernel<64>, grid, dim3(256), 0, s, x_dev, gamma_dev, beta_dev, y_dev, C, n_cols, eps);
  else if (C <= 1024) hipLaunchKernelGGL(layernorm_cbt_kernel<256>, grid, dim3(256), 0, s, x_dev, gamma_dev, beta_dev, y_dev, C, n_cols, eps);
  else return fail(EVMI_ERR_UNSUPPORTED, "layernorm_cbt: more than 1024 channels");
  EVMI_LAUNCH_CHECK("layernorm_cbt");
  return EVMI_OK;
}

int evmi_dwconv1d_cbt_f32(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int C, int B, int T, int k,
                          int pad, int act, void* stream) {
  EVMI_NONNULL(x_dev && w_dev && y_dev, "dwconv1d_cbt");
  if (C <= 0 || B <= 0 || T <= 0 || k <= 0 || act < 0 || act > 2) return fail(EVMI_ERR_INVALID_ARG, "dwconv1d_cbt: shape");
  hipLaunchKernelGGL(dwconv_cbt_kernel, grid1d((long long)C * B * T), dim3(256), 0, (hipStream_t)stream, x_dev, w_dev, bias_dev,
                     y_dev, C, B, T, k, pad, act);
  EVMI_LAUNCH_CHECK("dwconv1d_cbt");
  return EVMI_OK;
}

int evmi_fs2_bucket_embed_add_f32(float* x_dev, const float* values_dev, const float* bins_dev, const float* table_dev, int n_bins,
                                  int B, int L, int D, float control, void* stream) {
  EVMI_NONNULL(x_dev && values_dev && bins_dev && table_dev, "fs2_bucket_embed_add");
  if (n_bins < 2) return fail(EVMI_ERR_INVALID_ARG, "fs2_bucket_embed_add: n_bins");
  hipLaunchKernelGGL(fs2_bucket_embed_add_kernel, grid1d((long long)D * B * L), dim3(256), 0, (hipStream_t)stream, x_dev, values_dev,
                     bins_dev, table_dev, n_bins, B, L, D, control);
  EVMI_LAUNCH_CHECK("fs2_bucket_embed_add");
  return EVMI_OK;
}

int evmi_fs2_durations_i32(const float* log_d_dev, const int* lens_dev, int* dur_dev, int B, int L, float control, void* stream) {
  EVMI_NONNULL(log_d_dev && lens_dev && dur_dev, "fs2_durations");
  hipLaunchKernelGGL(fs2_durations_kernel, grid1d((long long)B * L), dim3(256), 0, (hipStream_t)stream, log_d_dev, lens_dev, dur_dev,
                     B, L, control);
  EVMI_LAUNCH_CHECK("fs2_durations");
  return EVMI_OK;
}

int evmi_length_regulate_cbt_f32(const float* x_dev, const int* cum_dev, float* out_dev, int C, int B, int L, int T, void* stream) {
  EVMI_NONNULL(x_dev && cum_dev && out_dev, "length_regulate_cbt");
  if (C <= 0 || B <= 0 || L <= 0 || T <= 0) return fail(EVMI_ERR_INVALID_ARG, "length_regulate_cbt: shape");
  hipLaunchKernelGGL(length_regulate_cbt_kernel, grid1d((long long)C * B * T), dim3(256), 0, (hipStream_t)stream, x_dev, cum_dev,
                     out_dev, C, B, L, T);
  EVMI_LAUNCH_CHECK("length_regulate_cbt");
  return EVMI_OK;
}

int evmi_attention_cbt_bf16(const float* qkv_dev, const int* lens_dev, float* out_dev, int B, int T, int D, int heads, void* stream) {
  EVMI_NONNULL(qkv_dev && lens_dev && out_dev, "attention_cbt_bf16");
  if (B <= 0 || T <= 0 || D <= 0 || heads <= 0 || D % heads) return fail(EVMI_ERR_INVALID_ARG, "attention_cbt_bf16: shape");
  if (B > 65535 || heads > 65535) return fail(EVMI_ERR_UNSUPPORTED, "attention_cbt_bf16: grid limits");
  if (launch_mha_fwd_bf16_plain(qkv_dev, lens_dev, out_dev, nullptr, B, T, D, heads, (hipStream_t)stream))
    return fail(EVMI_ERR_UNSUPPORTED, "attention_cbt_bf16: head dimension must be 32, 64 or 128");
  EVMI_LAUNCH_CHECK("attention_cbt_bf16");
  return EVMI_OK;
}

int evmi_attention_cbt_f32(const float* qkv_dev, const int* lens_dev, float* out_dev, int B, int T, int D, int heads, void* stream) {
  EVMI_NONNULL(qkv_dev && lens_dev && out_dev, "attention_cbt");
  if (B <= 0 || T <= 0 || D <= 0 || heads <= 0 || D % heads) return fail(EVMI_ERR_INVALID_ARG, "attention_cbt: shape");
  if (B > 65535 || heads > 65535) return fail(EVMI_ERR_UNSUPPORTED, "attention_cbt: grid limits");
  const int dh = D / heads;
  const float scale = 1.f / sqrtf((float)dh);
  const dim3 grid((T + 127) / 128, heads, B);
  hipStream_t s = (hipStream_t)stream;
  if (dh == 128) hipLaunchKernelGGL(attention_cbt_kernel<128>, grid, dim3(256), 0, s, qkv_dev, lens_dev, out_dev, B, T, D, scale);
  else if (dh == 64) hipLaunchKernelGGL(attention_cbt_kernel<64>, grid, dim3(256), 0, s, qkv_dev, lens_dev, out_dev, B, T, D, scale);
  else if (dh == 32) hipLaunchKernelGGL(attention_cbt_kernel<32>, grid, dim3(256), 0, s, qkv_dev, lens_dev, out_dev, B, T, D, scale);
  else return fail(EVMI_ERR_UNSUPPORTED, "attention_cbt: head dimension must be 32, 64 or 128");
  EVMI_LAUNCH_CHECK("attention_cbt");
  return EVMI_OK;
}

}  // extern "C"

// ---- beta-binomial attention prior (everyvoice/preprocessor/attention_prior.py:34-67), float64 -----------
namespace evmi {

__device__ inline double betaln_d(double x, double y) { return lgamma(x) + lgamma(y) - lgamma(x + y); }

// table[ti][li] = pmf of BetaBinomial(n = bw, a = li + 1, b = bh - li) at k = ti  (the reference's bank, transposed)
__device__ inline double prior_table(int ti, int li, int bw, int bh) {
  if (ti < 0 || ti >= bw || li < 0 || li >= bh) return 0.0;  // scipy.ndimage.zoom(mode='constant', cval=0)
  const double n = (double)bw, k = (double)ti, a = (double)(li + 1), b = (double)(bh - li);
  const double log_comb = lgamma(n + 1.0) - lgamma(k + 1.0) - lgamma(n - k + 1.0);
  return exp(log_comb + betaln_d(k + a, n - k + b) - betaln_d(a, b));
}

// out[t][l] = order-1 zoom of table [bw][bh] to [T][L]: output index o samples input coordinate o * (in-1)/(out-1)
__global__ void attention_prior_kernel(double* __restrict__ out, int T, int L, int bw, int bh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * L) return;
  const int t = i / L, l = i - t * L;
  const double zt = T > 1 ? (double)(bw - 1) / (double)(T - 1) : 0.0;
  const double zl = L > 1 ? (double)(bh - 1) / (double)(L - 1) : 0.0;
  const double ct = t * zt, cl = l * zl;
  const int ft = (int)floor(ct), fl = (int)floor(cl);
  const double wt = ct - ft, wl = cl - fl;
  double v = 0.0;
  v += (1.0 - wt) * (1.0 - wl) * prior_table(ft, fl, bw, bh);
  if (wl != 0.0) v += (1.0 - wt) * wl * prior_table(ft, fl + 1, bw, bh);
  if (wt != 0.0) v += wt * (1.0 - wl) * prior_table(ft + 1, fl, bw, bh);
  if (wt != 0.0 && wl != 0.0) v += wt * wl * prior_table(ft + 1, fl + 1, bw, bh);
  out[i] = v;
}

}  // namespace evmi

extern "C" int evmi_attention_prior_f64(double* out_dev, int T, int L, int grid_mel, int grid_text, void* stream) {
  if (!out_dev) return evmi::fail(EVMI_ERR_INVALID_ARG, "attention_prior: null pointer");
  if (T <= 0 || L <= 0 || grid_mel <= 0 || grid_text <= 0) return evmi::fail(EVMI_ERR_INVALID_ARG, "attention_prior: shape");
  hipLaunchKernelGGL(evmi::attention_prior_kernel, dim3((T * L + 255) / 256), dim3(256), 0, (hipStream_t)stream, out_dev, T, L, grid_mel,
                     grid_text);
  EVMI_LAUNCH_CHECK("attention_prior");
  return EVMI_OK;
}

// ---- monotonic alignment search (hard alignments; Glow-TTS maximum_path) -----------------------------
namespace evmi {

// One workgroup per item.  The DP runs row by row (frames), tokens in parallel, two rows of Q in LDS; the decisions
// (came from x - 1?) are kept as one bit per cell in global scratch for the read-back.  Same arithmetic and tie rule as the
// published kernel: Q[y][x] = value + max(prev, cur) in fp32, move left when x == y or Q[y-1][x] < Q[y-1][x-1].
__global__ __launch_bounds__(256) void mas_kernel(const float* __restrict__ value, const int* __restrict__ mel_lens,
                                                  const int* __restrict__ text_lens, int* __restrict__ path, int* __restrict__ dur,
                                                  unsigned char* __restrict__ left, int T, int L) {
  extern __shared__ float q[];  // [2][L]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int t_y = min(mel_lens[b], T), t_x = min(text_lens[b], L);
  const float* v = value + (long long)b * T * L;
  int* p = path + (long long)b * T * L;
  unsigned char* lf = left + (long long)b * T * L;
  const float MAX_NEG = -1e9f;
  for (int i = tid; i < T * L; i += 256) p[i] = 0;
  for (int x = tid; x < L; x += 256) dur[b * L + x] = 0;
  for (int y = 0; y < t_y; ++y) {
    float* cur = q + (y & 1) * L;
    const float* prev = q + ((y & 1) ^ 1) * L;
    const int lo = max(0, t_x + y - t_y), hi = min(t_x, y + 1);
    for (int x = lo + tid; x < hi; x += 256) {
      const float v_cur = x == y ? MAX_NEG : prev[x];
      const float v_prev = x == 0 ? (y == 0 ? 0.f : MAX_NEG) : prev[x - 1];
      cur[x] = v[(long long)y * L + x] + fmaxf(v_prev, v_cur);
      // the read-back at row y + 1 ... asks about row y: decide "x came from x - 1" now, while both values are here
      lf[(long long)y * L + x] = (x != 0 && (x == y || v_cur < v_prev)) ? 1 : 0;
    }
    __syncthreads();
  }
  if (tid == 0 && t_y > 0 && t_x > 0) {
    // read-back: at row y the published kernel tests Q[y-1][index] < Q[y-1][index-1] or index == y: exactly the bit stored
    // for cell (y, index) (v_cur / v_prev are those two values; x == y covers the diagonal)
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
      p[(long long)y * L + index] = 1;
      dur[b * L + index] += 1;
      if (index != 0 && lf[(long long)y * L + index]) --index;
    }
  }
}

}  // namespace evmi

extern "C" int evmi_monotonic_align_f32(const float* value_dev, const int* mel_lens_dev, const int* text_lens_dev, int* path_dev,
                                        int* dur_dev, unsigned char* scratch_dev, int B, int T, int L, void* stream) {
  if (!value_dev || !mel_lens_dev || !text_lens_dev || !path_dev || !dur_dev || !scratch_dev)
    return evmi::fail(EVMI_ERR_INVALID_ARG, "monotonic_align: null pointer");
  if (B <= 0 || T <= 0 || L <= 0) return evmi::fail(EVMI_ERR_INVALID_ARG, "monotonic_align: shape");
  if ((size_t)2 * L * sizeof(float) > 64 * 1024) return evmi::fail(EVMI_ERR_UNSUPPORTED, "monotonic_align: more than 8192 tokens");
  hipLaunchKernelGGL(evmi::mas_kernel, dim3(B), dim3(256), 2 * L * sizeof(float), (hipStream_t)stream, value_dev, mel_lens_dev,
                     text_lens_dev, path_dev, dur_dev, scratch_dev, T, L);
  EVMI_LAUNCH_CHECK("monotonic_align");
  return EVMI_OK;
}

// ---- per-item embeddings (speaker / language): x[c][b][l] += table[ids[b]][c] for l < lens[b] ---------------------------
namespace evmi {
__global__ void fs2_add_item_embedding_kernel(float* __restrict__ x, const int* __restrict__ ids, const int* __restrict__ lens,
                                              const float* __restrict__ table, int B, int L, int D) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)D * B * L) return;
  const int l = (int)(i % L), b = (int)((i / L) % B), c = (int)(i / ((long long)L * B));
  if (l < lens[b]) x[i] += table[(long long)ids[b] * D + c];
}
}  // namespace evmi

extern "C" int evmi_fs2_add_item_embedding_f32(float* x_dev, const int* ids_dev, const int* lens_dev, const float* table_dev, int B,
                                               int L, int D, void* stream) {
  if (!x_dev || !ids_dev || !lens_dev || !table_dev) return evmi::fail(EVMI_ERR_INVALID_ARG, "fs2_add_item_embedding: null pointer");
  hipLaunchKernelGGL(evmi::fs2_add_item_embedding_kernel, dim3((unsigned)(((long long)D * B * L + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, x_dev, ids_dev, lens_dev, table_dev, B, L, D);
  EVMI_LAUNCH_CHECK("fs2_add_item_embedding");
  return EVMI_OK;
}

// ---- alignment learning (Badlani et al. 2021 as implemented in FastPitch) ------------------------------
namespace evmi {

// One workgroup per (item, frame): scores over the tokens, log-softmax (+ log prior), masked softmax.
// q [A][B][T], k [A][B][L] channel-major; prior [B][T][L] float64 or nullptr; soft / logprob [B][T][L].
__global__ __launch_bounds__(256) void align_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                             const double* __restrict__ prior, const int* __restrict__ text_lens,
                                                             float* __restrict__ soft, float* __restrict__ logprob, int A, int B,
                                                             int T, int L, float temperature) {
  extern __shared__ float sm[];  // [A] query vector, [L] scores, [8] reduction scratch
  float* qv = sm;
  float* sc = sm + A;
  float* red = sc + L;
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < A; c += 256) qv[c] = q[((long long)c * B + b) * T + t];
  __syncthreads();
  for (int l = tid; l < L; l += 256) {
    float d = 0.f;
    for (int c = 0; c < A; ++c) {
      const float df = qv[c] - k[((long long)c * B + b) * L + l];
      d = fmaf(df, df, d);
    }
    sc[l] = -temperature * d;
  }
  __syncthreads();
  auto block_max = [&](float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    v = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    return v;
  };
  auto block_sum = [&](float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    v = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return v;
  };
  const long long row = ((long long)b * T + t) * L;
  if (prior) {  // log_softmax over ALL tokens (padding included, as the reference does), then the prior
    float m = -INFINITY;
    for (int l = tid; l < L; l += 256) m = fmaxf(m, sc[l]);
    m = block_max(m);
    float s = 0.f;
    for (int l = tid; l < L; l += 256) s += expf(sc[l] - m);
    s = block_sum(s);
    const float lse = m + logf(s);
    for (int l = tid; l < L; l += 256) sc[l] = sc[l] - lse + logf((float)prior[row + l] + 1e-8f);
    __syncthreads();
  }
  for (int l = tid; l < L; l += 256) logprob[row + l] = sc[l];
  const int len = min(text_lens[b], L);
  float m = -INFINITY;
  for (int l = tid; l < len; l += 256) m = fmaxf(m, sc[l]);
  m = block_max(m);
  float s = 0.f;
  for (int l = tid; l < len; l += 256) s += expf(sc[l] - m);
  s = block_sum(s);
  for (int l = tid; l < L; l += 256) soft[row + l] = l < len ? expf(sc[l] - m) / s : 0.f;
}

__device__ inline float logaddexp_f(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b);
  return m + log1pf(expf(-fabsf(a - b)));
}

// CTC forward-sum per item: targets 1..L_b, blank 0 prepended with a fixed log-probability; one workgroup per item,
// the extended target (2 L_b + 1 states) in parallel, frames in sequence.  loss[b] = -log p(target) / L_b.
__global__ __launch_bounds__(256) void forward_sum_kernel(const float* __restrict__ logprob, const int* __restrict__ text_lens,
                                                         const int* __restrict__ mel_lens, float* __restrict__ loss, int T, int L,
                                                         float blank_logprob) {
  extern __shared__ float sm[];  // [L + 1] normalised row, 2 x [2L + 1] alpha, [8] scratch
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Lb = min(text_lens[b], L), Tb = min(mel_lens[b], T), S = 2 * Lb + 1;
  float* lp = sm;
  float* alpha0 = lp + (L + 1);
  float* alpha1 = alpha0 + (2 * L + 1);
  float* red = alpha1 + (2 * L + 1);
  if (Lb <= 0 || Tb <= 0) { if (tid == 0) loss[b] = 0.f; return; }
  for (int t = 0; t < Tb; ++t) {
    const float* row = logprob + ((long long)b * T + t) * L;
    // log_softmax over [blank, tokens 0..Lb-1]
    float m = blank_logprob;
    for (int l = tid; l < Lb; l += 256) m = fmaxf(m, row[l]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = tid == 0 ? expf(blank_logprob - m) : 0.f;
    for (int l = tid; l < Lb; l += 256) s += expf(row[l] - m);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    const float lse = m + logf(red[0] + red[1] + red[2] + red[3]);
    __syncthreads();
    if (tid == 0) lp[0] = blank_logprob - lse;
    for (int l = tid; l < Lb; l += 256) lp[l + 1] = row[l] - lse;
    __syncthreads();
    float* cur = (t & 1) ? alpha1 : alpha0;
    const float* prev = (t & 1) ? alpha0 : alpha1;
    for (int st = tid; st < S; st += 256) {
      const int lab = (st & 1) ? (st + 1) / 2 : 0;  // extended target: blank, 1, blank, 2, ...
      float a;
      if (t == 0) a = st < 2 ? 0.f : -INFINITY;
      else {
        a = prev[st];
        if (st >= 1) a = logaddexp_f(a, prev[st - 1]);
        if ((st & 1) && st >= 3) a = logaddexp_f(a, prev[st - 2]);  // consecutive labels always differ
      }
      cur[st] = a + lp[lab];
    }
    __syncthreads();
  }
  if (tid == 0) {
    const float* fin = ((Tb - 1) & 1) ? alpha1 : alpha0;
    const float ll = S >= 2 ? logaddexp_f(fin[S - 1], fin[S - 2]) : fin[S - 1];
    loss[b] = ll == -INFINITY ? 0.f : -ll / (float)Lb;  // zero_infinity; reduction 'mean' divides by the target length
  }
}

// partial sums of log(max(soft, 1e-12)) over hard == 1 cells and of hard; finished on the host side of the wrapper
__global__ __launch_bounds__(256) void binarization_kernel(const int* __restrict__ hard, const float* __restrict__ soft, double* __restrict__ out,
                                                          long long n) {
  __shared__ double sh[2][4];
  double a = 0.0, c = 0.0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    if (hard[i] == 1) { a += (double)logf(fmaxf(soft[i], 1e-12f)); c += 1.0; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off, 64); c += __shfl_down(c, off, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = a; sh[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
    out[2 * blockIdx.x + 1] = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
}

}  // namespace evmi

extern "C" {

int evmi_align_attention_f32(const float* q_dev, const float* k_dev, const double* prior_dev, const int* text_lens_dev, float* soft_dev,
                             float* logprob_dev, int A, int B, int T, int L, float temperature, void* stream) {
  if (!q_dev || !k_dev || !text_lens_dev || !soft_dev || !logprob_dev) return evmi::fail(EVMI_ERR_INVALID_ARG, "align_attention: null pointer");
  if (A <= 0 || B <= 0 || T <= 0 || L <= 0 || B > 65535) return evmi::fail(EVMI_ERR_INVALID_ARG, "align_attention: shape");
  const size_t lds = (size_t)(A + L + 8) * sizeof(float);
  hipLaunchKernelGGL(evmi::align_attention_kernel, dim3(T, B), dim3(256), lds, (hipStream_t)stream, q_dev, k_dev, prior_dev, text_lens_dev,
                     soft_dev, logprob_dev, A, B, T, L, temperature);
  EVMI_LAUNCH_CHECK("align_attention");
  return EVMI_OK;
}

int evmi_forward_sum_loss_f32(const float* logprob_dev, const int* text_lens_dev, const int* mel_lens_dev, float* loss_per_item_dev, int B,
                              int T, int L, float blank_logprob, void* stream) {
  if (!logprob_dev || !text_lens_dev || !mel_lens_dev || !loss_per_item_dev) return evmi::fail(EVMI_ERR_INVALID_ARG, "forward_sum_loss: null pointer");
  if (B <= 0 || T <= 0 || L <= 0) return evmi::fail(EVMI_ERR_INVALID_ARG, "forward_sum_loss: shape");
  const size_t lds = (size_t)((L + 1) + 2 * (2 * L + 1) + 8) * sizeof(float);
  if (lds > 64 * 1024) return evmi::fail(EVMI_ERR_UNSUPPORTED, "forward_sum_loss: too many tokens");
  hipLaunchKernelGGL(evmi::forward_sum_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, logprob_dev, text_lens_dev, mel_lens_dev,
                     loss_per_item_dev, T, L, blank_logprob);
  EVMI_LAUNCH_CHECK("forward_sum_loss");
  return EVMI_OK;
}

int evmi_binarization_partials_f64(const int* hard_dev, const float* soft_dev, double* partials_dev, int n_blocks, long long n, void* stream) {
  if (!hard_dev || !soft_dev || !partials_dev || n_blocks <= 0) return evmi::fail(EVMI_ERR_INVALID_ARG, "binarization: arguments");
  hipLaunchKernelGGL(evmi::binarization_kernel, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, hard_dev, soft_dev, partials_dev, n);
  EVMI_LAUNCH_CHECK("binarization");
  return EVMI_OK;
}

}  // extern "C"
