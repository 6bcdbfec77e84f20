// FastSpeech2 training: the backward (and training-mode forward) operators that are not convolutions, channel-major
// fp32 layout x[c][b][t] (column n = b * T + t contiguous per channel row).  Dense layers reuse the fp32 matrix-core
// convolution kernels (forward / input gradient / weight gradient, k = 1 is a GEMM); attention in training mode keeps
// the probabilities (needed again by the backward and by attention dropout) and runs as batched GEMMs around the two
// row kernels here.
//
//   layernorm_bwd_cbt_kernel     dx of LayerNorm over channels + per-workgroup partial sums of d gamma / d beta
//   colsum_partials_kernel       fixed-order reduction of those partials (deterministic: no atomics on parameters
//                                except embedding tables, where rows collide by construction)
//   batchnorm_{stats,apply,bwd}  training-mode BatchNorm1d (batch statistics over every column, padded ones included:
//                                torchaudio's Conformer does not mask them) with SiLU / tanh fused behind it
//   dwconv_bwd_{dx,dw}           depthwise convolution backward
//   softmax_rows / softmax_bwd   key-masked softmax rows (+ dropout) and its backward, in place on [B][Tq][Tk]
//   glu_bwd, dropout             elementwise
//   embedding / bucket / item-embedding backward (scatter-add), length regulator backward (segment sums)
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "common.h"
#include "evmi.h"

namespace evmi {

// (the dropout generator `uniform01` lives in common.h: the attention kernels draw from the same streams)

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }  // (all 64 lanes live at every call site)

// ---- LayerNorm backward --------------------------------------------------------------------------------------------
// Workgroup = 64 columns x SL channel slices (one wave per slice): every thread keeps x and dy of its slice of one
// column in registers (one read of each, one write of dx).  part[blk][0][c] = sum over the 64 columns of dy * xhat,
// part[blk][1][c] = sum of dy: reduced afterwards in a fixed order.  SL = 8 keeps the slices at 32 channels for
// C = 256 (two 32-float register arrays: no spills, two workgroups per CU).
template <int CPT, int SL>
__global__ __launch_bounds__(64 * SL) void layernorm_bwd_cbt_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                   const float* __restrict__ dy, float* __restrict__ dx,
                                                                   float* __restrict__ part, int C, long long N, float eps,
                                                                   int accumulate) {
  __shared__ float red[4][SL][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long long n = (long long)blockIdx.x * 64 + lane;
  const bool live = n < N;
  const int per = (C + SL - 1) / SL, c0 = slice * per, c1 = min(C, c0 + per);
  float v[CPT], d[CPT];
  float s = 0.f;
  const float* xp = x + (long long)c0 * N + n;
  const float* dp = dy + (long long)c0 * N + n;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const bool on = live && c0 + i < c1;
    v[i] = on ? xp[(long long)i * N] : 0.f;
    d[i] = on ? dp[(long long)i * N] : 0.f;
    s += v[i];
  }
  red[0][slice][lane] = s;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int q = 0; q < SL; ++q) mean += red[0][q][lane];
  mean /= (float)C;
  float qs = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const float t = (c0 + i < c1) ? v[i] - mean : 0.f;
    qs = fmaf(t, t, qs);
  }
  red[1][slice][lane] = qs;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int q = 0; q < SL; ++q) var += red[1][q][lane];
  const float rstd = 1.f / sqrtf(var / (float)C + eps);
  float s1 = 0.f, s2 = 0.f;
  float* pg = part + (long long)blockIdx.x * 2 * C;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const int c = c0 + i;
    const bool on = c < c1;
    v[i] = on ? (v[i] - mean) * rstd : 0.f;  // xhat
    const float a = wave_sum(d[i] * v[i]), b = wave_sum(d[i]);
    if (on && lane == 0) {
      pg[c] = a;
      pg[C + c] = b;
    }
    d[i] = on ? d[i] * gamma[c] : 0.f;  // dy * gamma
    s1 += d[i];
    s2 = fmaf(d[i], v[i], s2);
  }
  red[2][slice][lane] = s1;
  red[3][slice][lane] = s2;
  // the gradient so far (accumulate): every row requested here, together, under the reduction below -- read inside the store loop
  // each would wait for the store in front of it (the compiler cannot tell rows N apart from each other: 32 dependent round trips)
  // the gradient so far (accumulate) is read in batches of eight rows below: read inside the store loop, one row at a time, each load
  // would wait for its own use behind the store in front of it -- 32 dependent round trips; all 32 rows requested up here cost 32
  // registers, the second workgroup of the CU and 40 % of the kernel's speed without accumulation (33.6 -> 47.7 us)
  float* op = dx + (long long)c0 * N + n;
  __syncthreads();
  float m1 = 0.f, m2 = 0.f;
#pragma unroll
  for (int q = 0; q < SL; ++q) {
    m1 += red[2][q][lane];
    m2 += red[3][q][lane];
  }
  m1 /= (float)C;
  m2 /= (float)C;
  if (!live) return;
  if (accumulate) {  // (wave-uniform)
    constexpr int EB = 8;
    static_assert(CPT % EB == 0, "batches of eight rows");
#pragma unroll
    for (int i0 = 0; i0 < CPT; i0 += EB) {
      float old[EB];
#pragma unroll
      for (int e = 0; e < EB; ++e) old[e] = op[(long long)min(i0 + e, c1 - c0 - 1) * N];  // (clamped: unconditional loads)
#pragma unroll
      for (int e = 0; e < EB; ++e) {
        const int i = i0 + e;
        if (c0 + i < c1) op[(long long)i * N] = old[e] + rstd * (d[i] - m1 - v[i] * m2);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    if (c0 + i < c1) op[(long long)i * N] = rstd * (d[i] - m1 - v[i] * m2);
  }
}

// out0[c] += sum_blk part[blk][0][c]; out1[c] += sum_blk part[blk][1][c]: 16 channels x 16 segments of the partial list per
// workgroup, segment sums combined in a fixed order
__global__ __launch_bounds__(256) void colsum_partials_kernel(const float* __restrict__ part, float* __restrict__ out0,
                                                             float* __restrict__ out1, int C, int nblk) {
  __shared__ double sh[16][17];
  const int ch = threadIdx.x & 15, seg = threadIdx.x >> 4;
  const int i = blockIdx.x * 16 + ch;
  const int per = (nblk + 15) / 16;
  double acc = 0.0;
  if (i < 2 * C)
    for (int b = seg * per; b < min(nblk, (seg + 1) * per); ++b) acc += (double)part[(long long)b * 2 * C + i];
  sh[seg][ch] = acc;
  __syncthreads();
  if (seg == 0 && i < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][ch];
    float* dst = i < C ? out0 + i : out1 + (i - C);
    *dst += (float)t;
  }
}

// The same for a batch of partial lists in ONE launch (a training step's LayerNorm backwards leave their partials in buffers of
// their own and the step reduces them together where its backward ends: one launch instead of one per LayerNorm on the chain).
constexpr int COLSUM_MAX_JOBS = 32;
struct ColsumBatch {
  const float* part[COLSUM_MAX_JOBS];
  float* out0[COLSUM_MAX_JOBS];
  float* out1[COLSUM_MAX_JOBS];
  int C[COLSUM_MAX_JOBS], nblk[COLSUM_MAX_JOBS];
  int blk_start[COLSUM_MAX_JOBS + 1];  // workgroups, prefix sums
  int n;
};
__global__ __launch_bounds__(256) void colsum_partials_batch_kernel(ColsumBatch q) {
  __shared__ double sh[16][17];
  int l = 0;
  while (l + 1 < q.n && (int)blockIdx.x >= q.blk_start[l + 1]) ++l;
  const float* __restrict__ part = q.part[l];
  const int C = q.C[l], nblk = q.nblk[l];
  const int ch = threadIdx.x & 15, seg = threadIdx.x >> 4;
  const int i = ((int)blockIdx.x - q.blk_start[l]) * 16 + ch;
  const int per = (nblk + 15) / 16;
  double acc = 0.0;
  if (i < 2 * C)
    for (int b = seg * per; b < min(nblk, (seg + 1) * per); ++b) acc += (double)part[(long long)b * 2 * C + i];
  sh[seg][ch] = acc;
  __syncthreads();
  if (seg == 0 && i < 2 * C) {
    double t = 0.0;
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) t += sh[qq][ch];
    float* dst = i < C ? q.out0[l] + i : q.out1[l] + (i - C);
    *dst += (float)t;
  }
}

// ---- BatchNorm1d, training mode --------------------------------------------------------------------------------------
// one workgroup per channel row (N = B * T contiguous values)
template <int NT = 256>
__device__ __forceinline__ double block_sum(double v, double* sh) {
  v = wave_sum_dpp_f64(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = sh[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) t += sh[w];  // (wave order: fixed)
  return t;
}

__device__ __forceinline__ float bn_act(float z, int act) {
  if (act == 2) return z * sigmoidf_(z);
  if (act == 4) return tanhf(z);
  return z;
}
__device__ __forceinline__ float bn_act_grad(float z, int act) {
  if (act == 2) { const float s = sigmoidf_(z); return s * (1.f + z * (1.f - s)); }
  if (act == 4) { const float t = tanhf(z); return 1.f - t * t; }
  return 1.f;
}

// An NT-thread workgroup walking ONE row of N values (thread t takes t, t + NT, ...): `body(i, u)`-style loops compile to one load ->
// wait -> use per trip (a memory round trip per NT values: 68 us for a 26 k-value BatchNorm row at 256 threads).  row_walk issues the
// loads of eight trips before the first use.  NT = 1024 for the long rows (BN_LONG_ROW): one workgroup per channel is one workgroup per
// CU at 256 channels, and four waves do not keep enough loads in flight to reach the CU's share of the bandwidth (34 / 48 us forward /
// backward at 26 k values; the sums are added in another -- still fixed -- order than with 256 threads).
template <int NT, class Load, class Use>
__device__ __forceinline__ void row_walk(long long N, Load load, Use use) {
  long long i = threadIdx.x;
  for (; i + 7 * NT < N; i += 8 * NT) {
    decltype(load(i)) v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = load(i + u * NT);
#pragma unroll
    for (int u = 0; u < 8; ++u) use(i + u * NT, v[u]);
  }
  for (; i < N; i += NT) use(i, load(i));
}
constexpr long long BN_LONG_ROW = 8192;

template <int NT>
__global__ __launch_bounds__(NT) void batchnorm_fwd_cbt_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               long long N, float eps, float momentum, int act) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x;
  const float* xr = x + (long long)c * N;
  if (momentum < 0.f) {  // evaluation mode: the running statistics normalise, nothing is updated
    const float mean = running_mean[c], rstd = 1.f / sqrtf(running_var[c] + eps);
    if (threadIdx.x == 0) { mean_out[c] = mean; rstd_out[c] = rstd; }
    const float ga = gamma[c] * rstd, be = beta[c];
    float* yr = y + (long long)c * N;
    row_walk<NT>(N, [&](long long i) { return xr[i]; }, [&](long long i, float v) { yr[i] = bn_act((v - mean) * ga + be, act); });
    return;
  }
  double s = 0.0;
  row_walk<NT>(N, [&](long long i) { return xr[i]; }, [&](long long, float v) { s += (double)v; });
  const float mean = (float)(block_sum<NT>(s, sh) / (double)N);
  double q = 0.0;
  row_walk<NT>(N, [&](long long i) { return xr[i]; }, [&](long long, float v) { const float d = v - mean; q += (double)(d * d); });
  const double ss = block_sum<NT>(q, sh);
  const float var = (float)(ss / (double)N);
  const float rstd = 1.f / sqrtf(var + eps);
  if (threadIdx.x == 0) {
    mean_out[c] = mean;
    rstd_out[c] = rstd;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(ss / (double)(N > 1 ? N - 1 : 1));
    }
  }
  const float ga = gamma[c] * rstd, be = beta[c];
  float* yr = y + (long long)c * N;
  row_walk<NT>(N, [&](long long i) { return xr[i]; }, [&](long long i, float v) { yr[i] = bn_act((v - mean) * ga + be, act); });
}

template <int NT>
__global__ __launch_bounds__(NT) void batchnorm_bwd_cbt_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const float* __restrict__ mean_in,
                                                               const float* __restrict__ rstd_in, const float* __restrict__ dy,
                                                               float* __restrict__ dx, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, long long N, int act) {
  __shared__ double sh[NT / 64];
  const int c = blockIdx.x;
  const float* xr = x + (long long)c * N;
  const float* dr = dy + (long long)c * N;
  const float mean = mean_in[c], rstd = rstd_in[c], ga = gamma[c], be = beta[c];
  double s1 = 0.0, s2 = 0.0;
  struct XD { float x, d; };
  row_walk<NT>(N, [&](long long i) { return XD{xr[i], dr[i]}; }, [&](long long, XD v) {
    const float xh = (v.x - mean) * rstd;
    const float dz = v.d * bn_act_grad(xh * ga + be, act);
    s1 += (double)dz;
    s2 += (double)(dz * xh);
  });
  const double t1 = block_sum<NT>(s1, sh);
  const double t2 = block_sum<NT>(s2, sh);
  if (threadIdx.x == 0) {
    dbeta[c] += (float)t1;
    dgamma[c] += (float)t2;
  }
  const float m1 = (float)(t1 / (double)N), m2 = (float)(t2 / (double)N);
  float* dxr = dx + (long long)c * N;
  row_walk<NT>(N, [&](long long i) { return XD{xr[i], dr[i]}; }, [&](long long i, XD v) {
    const float xh = (v.x - mean) * rstd;
    const float dz = v.d * bn_act_grad(xh * ga + be, act);
    dxr[i] = ga * rstd * (dz - m1 - xh * m2);
  });
}

// ---- depthwise convolution backward ----------------------------------------------------------------------------------
// dx[c][b][t] = sum_j w[c][j] * dy[c][b][t + pad - j]
__global__ void dwconv_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, int C, int B,
                                     int T, int k, int pad) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)C * B * T) return;
  const int t = (int)(i % T);
  const int c = (int)(i / ((long long)T * B));
  const float* dr = dy + (i - t);
  float v = 0.f;
  for (int j = 0; j < k; ++j) {
    const int to = t + pad - j;
    if (to >= 0 && to < T) v = fmaf(w[c * k + j], dr[to], v);
  }
  dx[i] = v;
}

constexpr int DW_KMAX = 32;
// part[c][b][j] = sum_t x[c][b][t + j - pad] * dy[c][b][t] (j < k), part[c][b][k] = sum_t dy : one workgroup per (c, b) row
// STAGED: the row of x sits in LDS, zero padded by the kernel's reach (each thread otherwise issues k conditional global loads per
// position: the anti-pattern of DESIGN.md 9.12; rows longer than the LDS take the direct form)
// KM: the taps the unrolled loops cover (k <= KM <= DW_KMAX: 4 for the variance predictors' k = 3, 12 for the Conformer's k = 9 -- the
// 32-tap form spent three quarters of its multiply-adds on taps that do not exist)
template <bool STAGED, int KM>
__global__ __launch_bounds__(256) void dwconv_bwd_dw_partial_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   float* __restrict__ part, int B, int T, int k, int pad) {
  extern __shared__ float xs[];  // STAGED: [T + k - 1], xs[i] = x[i - pad]
  const int b = blockIdx.x, c = blockIdx.y;
  const float* xr = x + ((long long)c * B + b) * T;
  const float* dr = dy + ((long long)c * B + b) * T;
  float acc[KM];
#pragma unroll
  for (int j = 0; j < KM; ++j) acc[j] = 0.f;
  float sb = 0.f;
  if (STAGED) {
    for (int i = threadIdx.x; i < T + k - 1; i += 256) {
      const int ti = i - pad;
      const float v = xr[min(max(ti, 0), T - 1)];  // (requested unconditionally, selected afterwards)
      xs[i] = (ti >= 0 && ti < T) ? v : 0.f;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += 256) {
      const float d = dr[t];
      sb += d;
#pragma unroll
      for (int j = 0; j < KM; ++j)
        if (j < k) acc[j] = fmaf(xs[t + j], d, acc[j]);  // (a padded position adds 0 * d: the sum's bits do not change)
    }
  } else {
    for (int t = threadIdx.x; t < T; t += 256) {
      const float d = dr[t];
      sb += d;
#pragma unroll
      for (int j = 0; j < KM; ++j) {
        const int ti = t + j - pad;
        if (j < k && ti >= 0 && ti < T) acc[j] = fmaf(xr[ti], d, acc[j]);
      }
    }
  }
  // k + 1 workgroup sums: the wave sums of all of them first (DPP adds: no LDS round trips -- as ds_bpermute butterflies on doubles these
  // were 12 crossbar trips per value, the larger part of the workgroup's time), ONE barrier, then the four wave sums of value j added
  // by thread j
  __shared__ double shw[4][DW_KMAX + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j <= KM; ++j) {
    if (j < k || j == KM) {
      const double v = wave_sum_dpp_f64(j == KM ? (double)sb : (double)acc[j < KM ? j : 0]);
      if (lane == 0) shw[wave][j == KM ? DW_KMAX : j] = v;
    }
  }
  __syncthreads();
  float* pr = part + ((long long)c * B + b) * (k + 1);
  if (threadIdx.x < k) pr[threadIdx.x] = (float)(shw[0][threadIdx.x] + shw[1][threadIdx.x] + shw[2][threadIdx.x] + shw[3][threadIdx.x]);
  if (threadIdx.x == 0) pr[k] = (float)(shw[0][DW_KMAX] + shw[1][DW_KMAX] + shw[2][DW_KMAX] + shw[3][DW_KMAX]);
}

// dw[c][j] += sum_b part[c][b][j] ; db[c] += sum_b part[c][b][k]   (fixed order)
__global__ void dwconv_bwd_dw_final_kernel(const float* __restrict__ part, float* __restrict__ dw, float* __restrict__ db, int C, int B,
                                           int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= C * (k + 1)) return;
  const int c = i / (k + 1), j = i - c * (k + 1);
  double acc = 0.0;
  for (int b = 0; b < B; ++b) acc += (double)part[((long long)c * B + b) * (k + 1) + j];
  if (j < k) dw[c * k + j] += (float)acc;
  else if (db) db[c] += (float)acc;
}

// ---- attention rows ------------------------------------------------------------------------------------------------
// S[b][tq][tk] (scores, already scaled) -> P = softmax over tk < len[b] (0 beyond); Pd = dropout(P) when p > 0
// one wave per row; grid = rows / 4
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ S, float* __restrict__ Pd, const int* __restrict__ lens,
                                                          int B, int Tq, int Tk, float p, unsigned long long seed) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * Tq) return;
  const int b = (int)(row / Tq);
  const int len = min(lens[b], Tk);
  float* sr = S + row * Tk;
  float m = -INFINITY;
  for (int i = lane; i < len; i += 64) m = fmaxf(m, sr[i]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = 0.f;
  for (int i = lane; i < len; i += 64) s += expf(sr[i] - m);
  s = wave_sum(s);
  const float inv = 1.f / s, keep = 1.f / (1.f - p);
  for (int i = lane; i < Tk; i += 64) {
    const float v = i < len ? expf(sr[i] - m) * inv : 0.f;
    sr[i] = v;
    if (Pd) Pd[row * Tk + i] = uniform01(seed, (unsigned long long)(row * Tk + i)) >= p ? v * keep : 0.f;
  }
}

// dS = scale * P * (dPm - sum_k P * dPm), dPm = dP * dropout mask / (1 - p); in place on dP
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP, long long rows,
                                                              int Tk, float scale, float p, unsigned long long seed) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* pr = P + row * Tk;
  float* dr = dP + row * Tk;
  const float keep = 1.f / (1.f - p);
  float s = 0.f;
  for (int i = lane; i < Tk; i += 64) {
    float d = dr[i];
    if (p > 0.f) d = uniform01(seed, (unsigned long long)(row * Tk + i)) >= p ? d * keep : 0.f;
    s = fmaf(pr[i], d, s);
  }
  s = wave_sum(s);
  for (int i = lane; i < Tk; i += 64) {
    float d = dr[i];
    if (p > 0.f) d = uniform01(seed, (unsigned long long)(row * Tk + i)) >= p ? d * keep : 0.f;
    dr[i] = scale * pr[i] * (d - s);
  }
}

// ---- elementwise ---------------------------------------------------------------------------------------------------
// p = [a; b] (two halves of n elements): y = a * sigmoid(b);  da = dy * sigmoid(b), db = dy * a * sigmoid(b) (1 - sigmoid(b))
__global__ void glu_bwd_kernel(const float* __restrict__ p, const float* __restrict__ dy, float* __restrict__ dp, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float a = p[i], s = sigmoidf_(p[n + i]), d = dy[i];
  dp[i] = d * s;
  dp[n + i] = d * a * s * (1.f - s);
}

__global__ void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float p, SeedArg seed_arg) {
  const unsigned long long seed = seed_arg.get();
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = uniform01(seed, (unsigned long long)i) >= p ? x[i] / (1.f - p) : 0.f;
}

// dropout with its neighbour in one pass (the Conformer block's `x + s * dropout(h)` and `dropout(silu(h))`, and their backwards):
//   MODE 1: y = b + scale * drop(a)      2: y = drop(silu(a))      3: y = drop(a) * silu'(b)      4: y = scale * drop(a)
// drop(v)[i] = keep(seed, i) ? v[i] / (1 - p) : 0, the same stream as dropout_kernel; four elements per thread.
template <int MODE>
__global__ __launch_bounds__(256) void dropout_fused_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                            long long n, float p, SeedArg seed_arg, float scale) {
  const unsigned long long seed = seed_arg.get();
  const long long i0 = ((long long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= n) return;
  float av[4], bv[4];
  const bool full = i0 + 3 < n;
  if (full) {
    const float4 t = *reinterpret_cast<const float4*>(a + i0);
    av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
    if (MODE == 1 || MODE == 3) {
      const float4 u = *reinterpret_cast<const float4*>(b + i0);
      bv[0] = u.x; bv[1] = u.y; bv[2] = u.z; bv[3] = u.w;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const long long i = min(i0 + e, n - 1);
      av[e] = a[i];
      bv[e] = (MODE == 1 || MODE == 3) ? b[i] : 0.f;
    }
  }
  const float inv_keep = 1.f - p;
  float r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool keep = uniform01(seed, (unsigned long long)(i0 + e)) >= p;
    float v = av[e];
    if (MODE == 2) v = silu_value(v);
    float d = keep ? v / inv_keep : 0.f;
    if (MODE == 1) d = bv[e] + scale * d;
    if (MODE == 3) d = d * silu_grad(bv[e]);
    if (MODE == 4) d = scale * d;
    r[e] = d;
  }
  if (full) {
    *reinterpret_cast<float4*>(y + i0) = make_float4(r[0], r[1], r[2], r[3]);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i0 + e < n) y[i0 + e] = r[e];
  }
}

// ---- embeddings, backward ------------------------------------------------------------------------------------------
// dtable[r][c] += sum of dx[c][n] over the tokens n that map to row r, added in a fixed order (four interleaved running sums in token
// order: bitwise reproducible, unlike a scatter with atomics).  text embedding: row = ids[b][l] for l < lens[b], ids != skip_id;  variance buckets: ids = the
// precomputed bucket index of every position, lens = NULL (the forward adds everywhere, pads included).
// One workgroup per CHANNEL (x 256 table rows): the channel's gradient row and the row indices go through LDS in chunks (coalesced
// reads, each once), and thread r walks the chunk's tokens for ITS table row -- every LDS read is a broadcast, there is no gather.
// (The form this replaces had one workgroup per table row gather its matches from the channel-major tensor: 64 cache lines per wave
// load, 105 us per call at 4.5 k positions.)
constexpr int TABLE_CHUNK = 4096;
__global__ __launch_bounds__(256) void fs2_table_bwd_kernel(const float* __restrict__ dx, const int* __restrict__ ids,
                                                           const int* __restrict__ lens, float* __restrict__ dtable, int B, int L,
                                                           int D, int skip_id, int rows) {
  __shared__ __attribute__((aligned(16))) int sid[TABLE_CHUNK];    // row index of every token of the chunk (-1: skipped)
  __shared__ __attribute__((aligned(16))) float val[TABLE_CHUNK];  // the channel's gradient at those tokens
  const int c = blockIdx.x;
  const int r = blockIdx.y * 256 + threadIdx.x;
  const int N = B * L;
  const float* row = dx + (long long)c * N;
  float acc4[4] = {0.f, 0.f, 0.f, 0.f};
  for (int base = 0; base < N; base += TABLE_CHUNK) {
    const int cnt = min(TABLE_CHUNK, N - base);
    __syncthreads();
    for (int i = threadIdx.x; i < TABLE_CHUNK; i += 256) {
      const int n = min(base + i, N - 1), b = n / L, l = n - b * L;
      const int id = ids[n];
      const float v = row[n];
      const bool live = i < cnt && !((lens && l >= lens[b]) || id == skip_id);
      sid[i] = live ? id : -1;
      val[i] = v;
    }
    __syncthreads();
    // four running sums (tokens 4 i + e): a fixed order -- bitwise reproducible -- without one chain of N dependent additions
    const int cnt4 = (cnt + 3) & ~3;  // (the chunk's tail is staged as skipped tokens)
#pragma unroll 4
    for (int i = 0; i < cnt4; i += 4) {
      const int4 s4 = *reinterpret_cast<const int4*>(&sid[i]);
      const float4 v4 = *reinterpret_cast<const float4*>(&val[i]);
      acc4[0] += s4.x == r ? v4.x : 0.f;
      acc4[1] += s4.y == r ? v4.y : 0.f;
      acc4[2] += s4.z == r ? v4.z : 0.f;
      acc4[3] += s4.w == r ? v4.w : 0.f;
    }
  }
  if (r < rows) dtable[(long long)r * D + c] += (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);
}

// bucket index of every token: first boundary >= value (torch.bucketize, right = False)
__global__ void fs2_bucket_index_kernel(const float* __restrict__ values, const float* __restrict__ bins, int* __restrict__ idx, int n,
                                        int n_bins, float control) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = values[i] * control;
  int lo = 0, hi = n_bins - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (bins[mid] < v) lo = mid + 1; else hi = mid;
  }
  idx[i] = lo;
}

// dtable[r][c] += sum over the items b with ids[b] == r of sum_{l < len[b]} dx[c][b][l]   (fixed order)
__global__ void fs2_item_embedding_bwd_kernel(const float* __restrict__ dx, const int* __restrict__ ids, const int* __restrict__ lens,
                                              float* __restrict__ dtable, int B, int L, int D, int rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= D * rows) return;
  const int c = i % D, r = i / D;
  float acc = 0.f;
  for (int b = 0; b < B; ++b) {
    if (ids[b] != r) continue;
    const float* p = dx + ((long long)c * B + b) * L;
    float s = 0.f;
    for (int l = 0; l < lens[b] && l < L; ++l) s += p[l];
    acc += s;
  }
  dtable[(long long)r * D + c] += acc;
}

// ---- length regulator backward: dx[c][b][l] = sum_{t in [cum[l-1], cum[l])} dframes[c][b][t] ----------------------------
__global__ void length_regulate_bwd_cbt_kernel(const float* __restrict__ dframes, const int* __restrict__ cum, float* __restrict__ dx,
                                               int C, int B, int L, int T) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)C * B * L) return;
  const int l = (int)(i % L), b = (int)((i / L) % B), c = (int)(i / ((long long)L * B));
  const int t0 = l ? cum[b * L + l - 1] : 0, t1 = min(cum[b * L + l], T);
  const float* r = dframes + ((long long)c * B + b) * T;
  float s = 0.f;
  for (int t = t0; t < t1; ++t) s += r[t];
  dx[i] = s;
}

static inline unsigned blocks_for(long long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

}  // namespace evmi

using namespace evmi;

extern "C" {

long long evmi_layernorm_bwd_cbt_f32_ws_elems(int C, long long n_cols) { return ((n_cols + 63) / 64) * 2ll * C; }

int evmi_layernorm_bwd_cbt_f32(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma, float* dbeta, float* ws,
                               long long ws_elems, int C, long long n_cols, float eps, int accumulate_dx, void* stream) {
  if (!x || !gamma || !dy || !dx || !ws || (dgamma == nullptr) != (dbeta == nullptr)) return fail(EVMI_ERR_INVALID_ARG, "layernorm_bwd: null pointer");
  if (C < 1 || C > 256 || n_cols < 1) return fail(EVMI_ERR_INVALID_ARG, "layernorm_bwd: 1 <= C <= 256 and n_cols >= 1 required");
  if (ws_elems < evmi_layernorm_bwd_cbt_f32_ws_elems(C, n_cols)) return fail(EVMI_ERR_INVALID_ARG, "layernorm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nblk = (unsigned)((n_cols + 63) / 64);
  if (C <= 64) hipLaunchKernelGGL((layernorm_bwd_cbt_kernel<16, 4>), dim3(nblk), dim3(256), 0, s, x, gamma, dy, dx, ws, C, n_cols, eps, accumulate_dx);
  // fewer workgroups than CUs (the encoder's 4.5 k columns): sixteen slices of 16 channels -- twice the waves on each of the few CUs
  // that work (21.6 -> 16.8 us); with every CU busy the eight-slice form is the faster one (35.9 vs 43.0 us at 26 k columns)
  else if (nblk <= 256) hipLaunchKernelGGL((layernorm_bwd_cbt_kernel<16, 16>), dim3(nblk), dim3(1024), 0, s, x, gamma, dy, dx, ws, C, n_cols, eps, accumulate_dx);
  else hipLaunchKernelGGL((layernorm_bwd_cbt_kernel<32, 8>), dim3(nblk), dim3(512), 0, s, x, gamma, dy, dx, ws, C, n_cols, eps, accumulate_dx);
  EVMI_LAUNCH_CHECK("layernorm_bwd_cbt");
  if (!dgamma) return EVMI_OK;  // (the partial sums stay in ws: evmi_layernorm_bwd_partials_reduce)
  hipLaunchKernelGGL(colsum_partials_kernel, dim3((2 * C + 15) / 16), dim3(256), 0, s, ws, dgamma, dbeta, C, (int)nblk);
  EVMI_LAUNCH_CHECK("colsum_partials");
  return EVMI_OK;
}

int evmi_layernorm_bwd_partials_reduce(int n_jobs, const evmi_ln_partials* jobs, void* stream) {
  if (n_jobs < 0 || (n_jobs && !jobs)) return fail(EVMI_ERR_INVALID_ARG, "layernorm_bwd_partials_reduce: bad job list");
  for (int j0 = 0; j0 < n_jobs; j0 += COLSUM_MAX_JOBS) {
    ColsumBatch q;
    q.n = std::min(COLSUM_MAX_JOBS, n_jobs - j0);
    q.blk_start[0] = 0;
    for (int j = 0; j < q.n; ++j) {
      const evmi_ln_partials& jb = jobs[j0 + j];
      if (!jb.ws || !jb.dgamma || !jb.dbeta || jb.C < 1 || jb.C > 256 || jb.n_cols < 1)
        return fail(EVMI_ERR_INVALID_ARG, "layernorm_bwd_partials_reduce: null pointer or bad shape");
      q.part[j] = jb.ws; q.out0[j] = jb.dgamma; q.out1[j] = jb.dbeta; q.C[j] = jb.C; q.nblk[j] = (int)((jb.n_cols + 63) / 64);
      q.blk_start[j + 1] = q.blk_start[j] + (2 * jb.C + 15) / 16;
    }
    hipLaunchKernelGGL(colsum_partials_batch_kernel, dim3((unsigned)q.blk_start[q.n]), dim3(256), 0, (hipStream_t)stream, q);
    EVMI_LAUNCH_CHECK("colsum_partials_batch");
  }
  return EVMI_OK;
}

int evmi_batchnorm_fwd_cbt_f32(const float* x, const float* gamma, const float* beta, float* y, float* mean_out, float* rstd_out,
                               float* running_mean, float* running_var, int C, long long n_cols, float eps, float momentum, int act,
                               void* stream) {
  if (!x || !gamma || !beta || !y || !mean_out || !rstd_out) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_fwd: null pointer");
  if ((running_mean == nullptr) != (running_var == nullptr)) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_fwd: running_mean and running_var go together");
  if (C < 1 || n_cols < 1) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_fwd: empty input");
  if (act != 0 && act != 2 && act != 4) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_fwd: act must be 0 (none), 2 (SiLU) or 4 (tanh)");
  if (momentum < 0.f && (!running_mean || !running_var)) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_fwd: evaluation mode (momentum < 0) needs the running statistics");
  if (n_cols >= BN_LONG_ROW)
    hipLaunchKernelGGL(batchnorm_fwd_cbt_kernel<1024>, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta, y, mean_out, rstd_out,
                       running_mean, running_var, n_cols, eps, momentum, act);
  else
    hipLaunchKernelGGL(batchnorm_fwd_cbt_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean_out, rstd_out,
                       running_mean, running_var, n_cols, eps, momentum, act);
  EVMI_LAUNCH_CHECK("batchnorm_fwd_cbt");
  return EVMI_OK;
}

int evmi_batchnorm_bwd_cbt_f32(const float* x, const float* gamma, const float* beta, const float* mean, const float* rstd,
                               const float* dy, float* dx, float* dgamma, float* dbeta, int C, long long n_cols, int act, void* stream) {
  if (!x || !gamma || !beta || !mean || !rstd || !dy || !dx || !dgamma || !dbeta) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_bwd: null pointer");
  if (C < 1 || n_cols < 1) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_bwd: empty input");
  if (act != 0 && act != 2 && act != 4) return fail(EVMI_ERR_INVALID_ARG, "batchnorm_bwd: act must be 0 (none), 2 (SiLU) or 4 (tanh)");
  if (n_cols >= BN_LONG_ROW)
    hipLaunchKernelGGL(batchnorm_bwd_cbt_kernel<1024>, dim3(C), dim3(1024), 0, (hipStream_t)stream, x, gamma, beta, mean, rstd, dy, dx, dgamma,
                       dbeta, n_cols, act);
  else
    hipLaunchKernelGGL(batchnorm_bwd_cbt_kernel<256>, dim3(C), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, mean, rstd, dy, dx, dgamma,
                       dbeta, n_cols, act);
  EVMI_LAUNCH_CHECK("batchnorm_bwd_cbt");
  return EVMI_OK;
}

long long evmi_dwconv1d_bwd_cbt_f32_ws_elems(int C, int B, int k) { return (long long)C * B * (k + 1); }

int evmi_dwconv1d_bwd_cbt_f32(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, float* ws,
                              long long ws_elems, int C, int B, int T, int k, int pad, void* stream) {
  if (!x || !w || !dy) return fail(EVMI_ERR_INVALID_ARG, "dwconv_bwd: null pointer");
  if (C < 1 || B < 1 || T < 1 || k < 1 || k > DW_KMAX) return fail(EVMI_ERR_INVALID_ARG, "dwconv_bwd: 1 <= k <= 32 and a non-empty input required");
  hipStream_t s = (hipStream_t)stream;
  if (dx) {
    hipLaunchKernelGGL(dwconv_bwd_dx_kernel, dim3(blocks_for((long long)C * B * T)), dim3(256), 0, s, dy, w, dx, C, B, T, k, pad);
    EVMI_LAUNCH_CHECK("dwconv_bwd_dx");
  }
  if (dw) {
    if (!ws || ws_elems < evmi_dwconv1d_bwd_cbt_f32_ws_elems(C, B, k)) return fail(EVMI_ERR_INVALID_ARG, "dwconv_bwd: workspace missing or too small");
    const size_t lds = (size_t)(T + k - 1) * sizeof(float);
#define EVMI_DW_PARTIAL(KM)                                                                                                          \
  {                                                                                                                                  \
    if (lds <= 40 * 1024) hipLaunchKernelGGL((dwconv_bwd_dw_partial_kernel<true, KM>), dim3(B, C), dim3(256), lds, s, x, dy, ws, B, T, k, pad); \
    else hipLaunchKernelGGL((dwconv_bwd_dw_partial_kernel<false, KM>), dim3(B, C), dim3(256), 0, s, x, dy, ws, B, T, k, pad);        \
  }
    if (k <= 4) EVMI_DW_PARTIAL(4)
    else if (k <= 12) EVMI_DW_PARTIAL(12)
    else EVMI_DW_PARTIAL(DW_KMAX)
#undef EVMI_DW_PARTIAL
    EVMI_LAUNCH_CHECK("dwconv_bwd_dw_partial");
    hipLaunchKernelGGL(dwconv_bwd_dw_final_kernel, dim3(blocks_for((long long)C * (k + 1))), dim3(256), 0, s, ws, dw, db, C, B, k);
    EVMI_LAUNCH_CHECK("dwconv_bwd_dw_final");
  }
  return EVMI_OK;
}

int evmi_softmax_rows_f32(float* scores, float* dropped, const int* lens, int B, int Tq, int Tk, float p, unsigned long long seed,
                          void* stream) {
  if (!scores || !lens) return fail(EVMI_ERR_INVALID_ARG, "softmax_rows: null pointer");
  if (B < 1 || Tq < 1 || Tk < 1) return fail(EVMI_ERR_INVALID_ARG, "softmax_rows: empty input");
  if (p < 0.f || p >= 1.f || (p > 0.f && !dropped)) return fail(EVMI_ERR_INVALID_ARG, "softmax_rows: 0 <= p < 1, and p > 0 needs the second buffer");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(blocks_for((long long)B * Tq, 4)), dim3(256), 0, (hipStream_t)stream, scores,
                     p > 0.f ? dropped : nullptr, lens, B, Tq, Tk, p, seed);
  EVMI_LAUNCH_CHECK("softmax_rows");
  return EVMI_OK;
}

int evmi_softmax_bwd_rows_f32(const float* probs, float* dprobs, long long rows, int Tk, float scale, float p, unsigned long long seed,
                              void* stream) {
  if (!probs || !dprobs) return fail(EVMI_ERR_INVALID_ARG, "softmax_bwd_rows: null pointer");
  if (rows < 1 || Tk < 1 || p < 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "softmax_bwd_rows: bad sizes");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3(blocks_for(rows, 4)), dim3(256), 0, (hipStream_t)stream, probs, dprobs, rows, Tk,
                     scale, p, seed);
  EVMI_LAUNCH_CHECK("softmax_bwd_rows");
  return EVMI_OK;
}

int evmi_glu_bwd_f32(const float* p, const float* dy, float* dp, long long n_half, void* stream) {
  if (!p || !dy || !dp || n_half < 1) return fail(EVMI_ERR_INVALID_ARG, "glu_bwd: null pointer or empty input");
  hipLaunchKernelGGL(glu_bwd_kernel, dim3(blocks_for(n_half)), dim3(256), 0, (hipStream_t)stream, p, dy, dp, n_half);
  EVMI_LAUNCH_CHECK("glu_bwd");
  return EVMI_OK;
}

int evmi_dropout_fused_f32(int mode, const float* a, const float* b, float* y, long long n, float p, unsigned long long seed_value,
                           const unsigned long long* seed_base_dev, float scale, void* stream) {
  const SeedArg seed{seed_value, seed_base_dev};
  if (!a || !y || n < 1 || p < 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "dropout_fused: null pointer, empty input or p outside [0, 1)");
  if (mode < 1 || mode > 4 || ((mode == 1 || mode == 3) && !b)) return fail(EVMI_ERR_INVALID_ARG, "dropout_fused: mode 1..4 (1 and 3 take a second operand)");
  if (((uintptr_t)a | (uintptr_t)y | (uintptr_t)b) & 15) return fail(EVMI_ERR_INVALID_ARG, "dropout_fused: operands must be 16-byte aligned");
  const dim3 grid(blocks_for((n + 3) / 4));
  hipStream_t s = (hipStream_t)stream;
  if (mode == 1) hipLaunchKernelGGL(dropout_fused_kernel<1>, grid, dim3(256), 0, s, a, b, y, n, p, seed, scale);
  else if (mode == 2) hipLaunchKernelGGL(dropout_fused_kernel<2>, grid, dim3(256), 0, s, a, b, y, n, p, seed, scale);
  else if (mode == 3) hipLaunchKernelGGL(dropout_fused_kernel<3>, grid, dim3(256), 0, s, a, b, y, n, p, seed, scale);
  else hipLaunchKernelGGL(dropout_fused_kernel<4>, grid, dim3(256), 0, s, a, b, y, n, p, seed, scale);
  EVMI_LAUNCH_CHECK("dropout_fused");
  return EVMI_OK;
}

int evmi_dropout_f32(const float* x, float* y, long long n, float p, unsigned long long seed_value, const unsigned long long* seed_base_dev,
                     void* stream) {
  const SeedArg seed{seed_value, seed_base_dev};
  if (!x || !y || n < 1 || p < 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "dropout: null pointer, empty input or p outside [0, 1)");
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, p, seed);
  EVMI_LAUNCH_CHECK("dropout");
  return EVMI_OK;
}

int evmi_fs2_embed_bwd_f32(const float* dx, const int* ids, const int* lens, float* dtable, int rows, int B, int L, int D, int skip_id,
                           void* stream) {
  if (!dx || !ids || !lens || !dtable || rows < 1 || B < 1 || L < 1 || D < 1) return fail(EVMI_ERR_INVALID_ARG, "fs2_embed_bwd: bad arguments");
  hipLaunchKernelGGL(fs2_table_bwd_kernel, dim3(D, (rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, dx, ids, lens, dtable, B, L, D,
                     skip_id, rows);
  EVMI_LAUNCH_CHECK("fs2_embed_bwd");
  return EVMI_OK;
}

int evmi_fs2_bucket_embed_bwd_f32(const float* dx, const float* values, const float* bins, float* dtable, int* idx_ws, int n_bins, int B,
                                  int L, int D, float control, void* stream) {
  if (!dx || !values || !bins || !dtable || !idx_ws || n_bins < 2 || B < 1 || L < 1 || D < 1)
    return fail(EVMI_ERR_INVALID_ARG, "fs2_bucket_embed_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(fs2_bucket_index_kernel, dim3(blocks_for((long long)B * L)), dim3(256), 0, s, values, bins, idx_ws, B * L, n_bins, control);
  EVMI_LAUNCH_CHECK("fs2_bucket_index");
  // the text-embedding walk over precomputed indices: every position counts (lens = NULL -> full rows), no skipped id
  hipLaunchKernelGGL(fs2_table_bwd_kernel, dim3(D, (n_bins + 255) / 256), dim3(256), 0, s, dx, idx_ws, nullptr, dtable, B, L, D, -1, n_bins);
  EVMI_LAUNCH_CHECK("fs2_bucket_embed_bwd");
  return EVMI_OK;
}

int evmi_fs2_item_embedding_bwd_f32(const float* dx, const int* ids, const int* lens, float* dtable, int rows, int B, int L, int D,
                                    void* stream) {
  if (!dx || !ids || !lens || !dtable || rows < 1 || B < 1 || L < 1 || D < 1) return fail(EVMI_ERR_INVALID_ARG, "fs2_item_embedding_bwd: bad arguments");
  hipLaunchKernelGGL(fs2_item_embedding_bwd_kernel, dim3(blocks_for((long long)D * rows)), dim3(256), 0, (hipStream_t)stream, dx, ids, lens,
                     dtable, B, L, D, rows);
  EVMI_LAUNCH_CHECK("fs2_item_embedding_bwd");
  return EVMI_OK;
}

int evmi_length_regulate_bwd_cbt_f32(const float* dframes, const int* cum, float* dx, int C, int B, int L, int T, void* stream) {
  if (!dframes || !cum || !dx || C < 1 || B < 1 || L < 1 || T < 1) return fail(EVMI_ERR_INVALID_ARG, "length_regulate_bwd_cbt: bad arguments");
  hipLaunchKernelGGL(length_regulate_bwd_cbt_kernel, dim3(blocks_for((long long)C * B * L)), dim3(256), 0, (hipStream_t)stream, dframes,
                     cum, dx, C, B, L, T);
  EVMI_LAUNCH_CHECK("length_regulate_bwd_cbt");
  return EVMI_OK;
}

}  // extern "C"
