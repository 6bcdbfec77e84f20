// Training-side entry points for TIME-MAJOR bf16 activations -- the inference layout [item][row][channel] -- so that the generator's
// residual stacks train on the kernels its inference runs on (conv_tc_dma_kernel.h / conv_tc_kernel.h: 24-40 % of the dense bf16
// MFMA peak at these shapes, against 3-7 % for the packed channel-major kernels of conv_cbt_bf16_pk.hip) and no layout pass sits
// between two convolutions (prep_pk / pack2 were 27 % of the GAN step's HBM bytes, zero FLOPs).
//
// A TM tensor is bf16 [B][Tp][C] with the T valid rows of every item at rows [PL, PL + T) and ZERO rows around them; `base`
// pointers address row 0 of item 0 and the allocation extends a guard of zero rows to both sides (the weight-gradient kernel
// walks the flat row index across items and reads tap-shifted rows).  Kernels write valid rows only, so the zeros persist.
//
//   forward            y = conv_tc(x; W, b; leaky-ReLU on load / in the epilogue; + residual)
//   input gradient     dx = conv_tc(dy; W^T with taps reversed) * lrelu'(mask) + residual      (mask: conv_tc_mfma.h)
//   weight gradient    conv_wgrad_bf16_pk.hip, TM instantiation (units gathered from the rows by the LDS-direct loads)
//   bias gradient      column sums of dy (two passes, fixed order)
// plus the layout changes at the ends of a stack (fp32 [C][B][T] of the rest of the training graph <-> TM bf16) and the
// per-step weight re-layout into the tiles the convolution kernels stream (on the device: weights change every step).
#include <cstdint>
#include <string>

#include "common.h"
#include "conv_tc_mfma.h"
#include "evmi.h"

namespace evmi {

// dst (bf16, the layout relayout_conv of generator.hip builds on the host): [mtile][chunk][tap][BM][KC], wlayout 1: the eight
// 16-byte channel vectors of a row permuted (slot p of row m holds vector p ^ ((m >> 1) & 7)).
// mode 0: W[m][c][j] = w[m][c][j] (w [c_out][c_in][ks]);  mode 1 (input gradient of the convolution with w [c_in'][c_out'][ks],
// c_in' = this launch's c_out rows... see below): W[m][c][j] = w[c][m][ks - 1 - j] with w [c_in][c_out][ks] read as [c][m][.].
__global__ void relayout_tc_kernel(const float* __restrict__ w, bf16_t* __restrict__ dst, int c_out, int c_in, int ks, int BM, int KC, int wlayout,
                                   int mode, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int nch = c_in / KC;
  long long r = idx;
  const int cs = (int)(r % KC); r /= KC;
  const int mi = (int)(r % BM); r /= BM;
  const int j = (int)(r % ks); r /= ks;
  const int chn = (int)(r % nch); r /= nch;
  const int mt = (int)r;
  const int ci = wlayout == 1 ? ((((cs >> 3) ^ ((mi >> 1) & 7)) << 3) | (cs & 7)) : cs;
  const int m = mt * BM + mi, c = chn * KC + ci;
  const float v = mode == 0 ? w[((long long)m * c_in + c) * ks + j] : w[((long long)c * c_out + m) * ks + (ks - 1 - j)];
  dst[idx] = (bf16_t)v;
}

// x [C][B][T] fp32 -> tm [B][Tp][C] bf16 rows PL..PL+T-1: v = leaky_relu(x, slope) * scale.  One thread per (row, octet).
__global__ void cbt_to_tm_kernel(const float* __restrict__ x, bf16_t* __restrict__ tm, int C, int B, int T, int Tp, int PL, float slope, float scale,
                                 long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int octs = C >> 3;
  const int o = (int)(i % octs);
  const long long row = i / octs;  // b * T + t
  const int t = (int)(row % T), b = (int)(row / T);
  const long long cs = (long long)B * T;
  const float* src = x + (long long)(o * 8) * cs + (long long)b * T + t;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = src[e * cs];
  bf16x8 out;
#pragma unroll
  for (int e = 0; e < 8; ++e) out[e] = (bf16_t)(lrelu(v[e], slope) * scale);
  *reinterpret_cast<bf16x8*>(tm + ((long long)b * Tp + PL + t) * C + o * 8) = out;
}

// out [C][B][T] fp32 = scale * (a + b + c) of up to three TM tensors (b, c may be null).  One thread per (octet, row), rows fastest
// within a block of 64 so that the fp32 stores of a channel are 256-byte runs.
__global__ void tm_to_cbt_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b2, const bf16_t* __restrict__ c2, float* __restrict__ out,
                                 int C, int B, int T, int Tp, int PL, float scale, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long long rows = (long long)B * T;
  const long long blk = i >> 6;
  const int lane = (int)(i & 63);
  const long long rblocks = (rows + 63) >> 6;
  const int o = (int)(blk / rblocks);
  const long long row = (blk % rblocks) * 64 + lane;
  if (row >= rows) return;
  const int t = (int)(row % T), b = (int)(row / T);
  const long long off = ((long long)b * Tp + PL + t) * C + o * 8;
  const bf16x8 va = *reinterpret_cast<const bf16x8*>(a + off);
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (float)va[e];
  if (b2) {
    const bf16x8 vb = *reinterpret_cast<const bf16x8*>(b2 + off);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)vb[e];
  }
  if (c2) {
    const bf16x8 vc = *reinterpret_cast<const bf16x8*>(c2 + off);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += (float)vc[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) out[(long long)(o * 8 + e) * rows + row] = v[e] * scale;
}

__global__ void tm_lrelu_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long n_vec, float slope) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_vec) return;
  const bf16x8 v = reinterpret_cast<const bf16x8*>(x)[i];
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (bf16_t)lrelu((float)v[e], slope);
  reinterpret_cast<bf16x8*>(y)[i] = o;
}

// column sums of dy [rows][C] in two fixed-order passes: part[blk][C] over COLSUM_ROWS rows each, then db[c] (+)= sum_blk.
// 512 rows per workgroup: ~265 workgroups at 16 x 8192 rows (the first version gave every workgroup 2048 rows -- 66 workgroups on
// 256 CUs, 38 us per launch, 2.7 ms per GAN step; 256 rows made the second pass walk 66 partial rows per thread: 11 us).
constexpr int COLSUM_ROWS = 512;
__global__ __launch_bounds__(256) void tm_colsum_partial_kernel(const bf16_t* __restrict__ dy, float* __restrict__ part, long long rows, int C) {
  // thread = (row lane rl, octet o): 256 threads cover 256 / octs rows per pass
  const int octs = C >> 3;
  const int o = threadIdx.x % octs, rl = threadIdx.x / octs, rstep = 256 / octs;
  const long long r0 = (long long)blockIdx.x * COLSUM_ROWS;
  const long long r1 = min(rows, r0 + COLSUM_ROWS);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < rstep) {
    bf16x8 v[4];
    long long r = r0 + rl;
    for (; r + 3 * rstep < r1; r += 4 * rstep) {  // four rows in flight per thread
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(dy + (r + (long long)u * rstep) * C + o * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
    }
    for (; r < r1; r += rstep) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(dy + r * C + o * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)w[e];
    }
  }
  __shared__ float sh[256 * 8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < C) {  // channel c = o * 8 + e: add its rstep row lanes in order
    const int c = threadIdx.x, oo = c >> 3, e = c & 7;
    float s = 0.f;
    for (int q = 0; q < rstep; ++q) s += sh[(q * octs + oo) * 8 + e];
    part[(long long)blockIdx.x * C + c] = s;
  }
}
// db[c] (+)= sum over the nblk partial rows: one workgroup per 32 channels, 8 slices of the partial rows per channel summed in
// parallel (each in block order), the 8 slice sums added in slice order
__global__ __launch_bounds__(256) void tm_colsum_final_kernel(const float* __restrict__ part, float* __restrict__ db, int nblk, int C, int accumulate) {
  __shared__ float sh[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int per = (nblk + 7) / 8;
  float s = 0.f;
  const int lo = sl * per, hi = min(nblk, (sl + 1) * per);
  // (eight loads in flight: the plain loop is one L2 round trip per partial row, 11 us per launch x 72 launches per GAN step)
  if (c < C && hi > lo) s = ordered_sum_strided(part + (long long)lo * C + c, C, hi - lo);
  sh[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sh[q][cl];
    db[c] = accumulate ? db[c] + t : t;
  }
}

// The same for up to TM_COLSUM_JOBS tensors of one shape in one launch pair (the bias gradients of a whole residual stack: 18 tensors per
// stage -- 144 launches of 5-7 us on the backward chains of a GAN step as single calls): grid.y = tensor
constexpr int TM_COLSUM_JOBS = 24;
struct ColsumBatch {
  const bf16_t* dy[TM_COLSUM_JOBS];
  float* db[TM_COLSUM_JOBS];
};
__global__ __launch_bounds__(256) void tm_colsum_partial_batch_kernel(ColsumBatch b, float* __restrict__ part, long long rows, int C, int nblk) {
  const bf16_t* dy = b.dy[blockIdx.y];
  part += (long long)blockIdx.y * nblk * C;
  const int octs = C >> 3;
  const int o = threadIdx.x % octs, rl = threadIdx.x / octs, rstep = 256 / octs;
  const long long r0 = (long long)blockIdx.x * COLSUM_ROWS;
  const long long r1 = min(rows, r0 + COLSUM_ROWS);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (rl < rstep) {
    bf16x8 v[4];
    long long r = r0 + rl;
    for (; r + 3 * rstep < r1; r += 4 * rstep) {
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const bf16x8*>(dy + (r + (long long)u * rstep) * C + o * 8);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += (float)v[u][e];
    }
    for (; r < r1; r += rstep) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(dy + r * C + o * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += (float)w[e];
    }
  }
  __shared__ float sh[256 * 8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sh[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  if (threadIdx.x < C) {
    const int c = threadIdx.x, oo = c >> 3, e = c & 7;
    float s = 0.f;
    for (int q = 0; q < rstep; ++q) s += sh[(q * octs + oo) * 8 + e];
    part[(long long)blockIdx.x * C + c] = s;
  }
}
__global__ __launch_bounds__(256) void tm_colsum_final_batch_kernel(ColsumBatch b, const float* __restrict__ part, int nblk, int C, int accumulate) {
  __shared__ float sh[8][32];
  float* db = b.db[blockIdx.y];
  part += (long long)blockIdx.y * nblk * C;
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int c = blockIdx.x * 32 + cl;
  const int per = (nblk + 7) / 8;
  float s = 0.f;
  const int lo = sl * per, hi = min(nblk, (sl + 1) * per);
  if (c < C && hi > lo) s = ordered_sum_strided(part + (long long)lo * C + c, C, hi - lo);
  sh[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += sh[q][cl];
    db[c] = accumulate ? db[c] + t : t;
  }
}

// Every weight of a residual stack into its convolution kernel's tile layout in ONE launch: table[l] = {src offset (floats, from
// w_base), dst offset (bf16 elements, from dst_base), ks, BM, KC, wlayout, mode, 0}; grid (ceil(C * C * ks_max / 256), layers)
__global__ void relayout_tc_batched_kernel(const float* __restrict__ w_base, bf16_t* __restrict__ dst_base, const long long* __restrict__ table, int C) {
  const long long* e = table + (long long)blockIdx.y * 8;
  const int ks = (int)e[2], BM = (int)e[3], KC = (int)e[4], wlayout = (int)e[5], mode = (int)e[6];
  const long long n = (long long)C * C * ks;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int nch = C / KC;
  long long r = idx;
  const int cs = (int)(r % KC); r /= KC;
  const int mi = (int)(r % BM); r /= BM;
  const int j = (int)(r % ks); r /= ks;
  const int chn = (int)(r % nch); r /= nch;
  const int mt = (int)r;
  const int ci = wlayout == 1 ? ((((cs >> 3) ^ ((mi >> 1) & 7)) << 3) | (cs & 7)) : cs;
  const int m = mt * BM + mi, c = chn * KC + ci;
  const float* w = w_base + e[0];
  const float v = mode == 0 ? w[((long long)m * C + c) * ks + j] : w[((long long)c * C + m) * ks + (ks - 1 - j)];
  dst_base[e[1] + idx] = (bf16_t)v;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

int evmi_conv_tc_supported(int c_in, int c_out, int ks, int dil) { return find_conv_tc(c_in, c_out, ks, dil) != nullptr ? 1 : 0; }

int evmi_conv_tc_relayout_f32(const float* w_dev, void* dst_bf16_dev, int c_in, int c_out, int ks, int dil, int transpose, void* stream) {
  if (!w_dev || !dst_bf16_dev) return fail(EVMI_ERR_INVALID_ARG, "conv_tc_relayout: null pointer");
  const ConvTcLaunch* L = find_conv_tc(c_in, c_out, ks, dil);
  if (!L) return fail(EVMI_ERR_UNSUPPORTED, "conv_tc_relayout: no time-major convolution kernel for this shape");
  if (c_in % L->kc || c_out % L->bm) return fail(EVMI_ERR_UNSUPPORTED, "conv_tc_relayout: channels do not tile");
  const long long n = (long long)c_out * c_in * ks;
  hipLaunchKernelGGL(relayout_tc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w_dev,
                     reinterpret_cast<bf16_t*>(dst_bf16_dev), c_out, c_in, ks, L->bm, L->kc, L->wlayout, transpose ? 1 : 0, n);
  EVMI_LAUNCH_CHECK("relayout_tc");
  return EVMI_OK;
}

int evmi_conv_tc_tile_layout(int c_in, int c_out, int ks, int dil, int* bm, int* kc, int* wlayout) {
  const ConvTcLaunch* L = find_conv_tc(c_in, c_out, ks, dil);
  if (!L || !bm || !kc || !wlayout) return fail(EVMI_ERR_UNSUPPORTED, "conv_tc_tile_layout: no time-major convolution kernel for this shape");
  *bm = L->bm; *kc = L->kc; *wlayout = L->wlayout;
  return EVMI_OK;
}

int evmi_conv_tc_relayout_batched_f32(const float* w_base_dev, void* dst_base_bf16_dev, const long long* table_dev, int n_layers, int C, int ks_max,
                                      void* stream) {
  if (!w_base_dev || !dst_base_bf16_dev || !table_dev || n_layers <= 0 || C <= 0 || ks_max <= 0) return fail(EVMI_ERR_INVALID_ARG, "conv_tc_relayout_batched: bad arguments");
  const long long n = (long long)C * C * ks_max;
  hipLaunchKernelGGL(relayout_tc_batched_kernel, dim3((unsigned)((n + 255) / 256), n_layers), dim3(256), 0, (hipStream_t)stream, w_base_dev,
                     reinterpret_cast<bf16_t*>(dst_base_bf16_dev), table_dev, C);
  EVMI_LAUNCH_CHECK("relayout_tc_batched");
  return EVMI_OK;
}

int evmi_conv_tc_tm_bf16(const void* x_tm, const void* w_laid, const float* bias_dev, const void* res_tm, const void* mask_tm, void* out_tm, int B,
                         int T, int Tp, int PL, int c_in, int c_out, int ks, int dil, float pre_slope, float post_slope, float mask_slope,
                         float out_scale, void* stream) {
  if (!x_tm || !w_laid || !bias_dev || !out_tm) return fail(EVMI_ERR_INVALID_ARG, "conv_tc_tm: null pointer");
  if (B <= 0 || T <= 0 || Tp < T + PL || PL < 0) return fail(EVMI_ERR_INVALID_ARG, "conv_tc_tm: shape");
  const ConvTcLaunch* L = find_conv_tc(c_in, c_out, ks, dil);
  if (!L) return fail(EVMI_ERR_UNSUPPORTED, "conv_tc_tm: no time-major convolution kernel for this shape");
  ConvTcArgs a = {};
  a.x = reinterpret_cast<const bf16_t*>(x_tm) + (long long)PL * c_in;
  a.w = reinterpret_cast<const bf16_t*>(w_laid);
  a.bias = bias_dev;
  a.res = res_tm ? reinterpret_cast<const bf16_t*>(res_tm) + (long long)PL * c_out : nullptr;
  a.mask = mask_tm ? reinterpret_cast<const bf16_t*>(mask_tm) + (long long)PL * c_out : nullptr;
  a.out = reinterpret_cast<bf16_t*>(out_tm) + (long long)PL * c_out;
  a.t_in = T; a.n_rows = T; a.c_out = c_out; a.dil = dil; a.pad = dil * (ks - 1) / 2;
  a.x_batch_stride = (long long)Tp * c_in;
  a.out_batch_stride = (long long)Tp * c_out;
  a.out_row_stride = c_out; a.out_shift = 0; a.out_limit = (long long)T * c_out;
  a.pre_slope = pre_slope; a.post_slope = post_slope; a.out_scale = out_scale; a.accumulate = 0;
  a.mask_slope = mask_slope;
  return launch_conv_tc(L, a, B, (hipStream_t)stream);
}

int evmi_cbt_f32_to_tm_bf16(const float* x_dev, void* tm_dev, int C, int B, int T, int Tp, int PL, float slope, float scale, void* stream) {
  if (!x_dev || !tm_dev || C <= 0 || (C & 7) || B <= 0 || T <= 0 || Tp < PL + T) return fail(EVMI_ERR_INVALID_ARG, "cbt_f32_to_tm_bf16: bad arguments");
  const long long n = (long long)B * T * (C >> 3);
  hipLaunchKernelGGL(cbt_to_tm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x_dev, reinterpret_cast<bf16_t*>(tm_dev), C, B, T,
                     Tp, PL, slope, scale, n);
  EVMI_LAUNCH_CHECK("cbt_to_tm");
  return EVMI_OK;
}

int evmi_tm_bf16_to_cbt_f32(const void* a_tm, const void* b_tm, const void* c_tm, float* out_dev, int C, int B, int T, int Tp, int PL, float scale,
                            void* stream) {
  if (!a_tm || !out_dev || C <= 0 || (C & 7) || B <= 0 || T <= 0 || Tp < PL + T) return fail(EVMI_ERR_INVALID_ARG, "tm_bf16_to_cbt_f32: bad arguments");
  const long long rblocks = ((long long)B * T + 63) / 64;
  const long long n = rblocks * 64 * (C >> 3);
  hipLaunchKernelGGL(tm_to_cbt_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16_t*>(a_tm),
                     reinterpret_cast<const bf16_t*>(b_tm), reinterpret_cast<const bf16_t*>(c_tm), out_dev, C, B, T, Tp, PL, scale, n);
  EVMI_LAUNCH_CHECK("tm_to_cbt");
  return EVMI_OK;
}

int evmi_tm_lrelu_bf16(const void* x_tm, void* y_tm, long long n_elems, float slope, void* stream) {
  if (!x_tm || !y_tm || n_elems <= 0 || (n_elems & 7)) return fail(EVMI_ERR_INVALID_ARG, "tm_lrelu_bf16: bad arguments");
  const long long nv = n_elems >> 3;
  hipLaunchKernelGGL(tm_lrelu_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const bf16_t*>(x_tm),
                     reinterpret_cast<bf16_t*>(y_tm), nv, slope);
  EVMI_LAUNCH_CHECK("tm_lrelu");
  return EVMI_OK;
}

long long evmi_tm_colsum_bf16_ws_elems(long long rows, int C) { return ((rows + COLSUM_ROWS - 1) / COLSUM_ROWS) * C; }

int evmi_tm_colsum_bf16(const void* dy_tm, float* db_dev, float* ws_dev, long long ws_elems, long long rows, int C, int accumulate, void* stream) {
  if (!dy_tm || !db_dev || !ws_dev || rows <= 0 || C <= 0 || (C & 7) || C > 256) return fail(EVMI_ERR_INVALID_ARG, "tm_colsum_bf16: bad arguments (C: multiple of 8, <= 256)");
  const long long nblk = (rows + COLSUM_ROWS - 1) / COLSUM_ROWS;
  if (ws_elems < nblk * C) return fail(EVMI_ERR_INVALID_ARG, "tm_colsum_bf16: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(tm_colsum_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, s, reinterpret_cast<const bf16_t*>(dy_tm), ws_dev, rows, C);
  EVMI_LAUNCH_CHECK("tm_colsum_partial");
  hipLaunchKernelGGL(tm_colsum_final_kernel, dim3((C + 31) / 32), dim3(256), 0, s, ws_dev, db_dev, (int)nblk, C, accumulate);
  EVMI_LAUNCH_CHECK("tm_colsum_final");
  return EVMI_OK;
}

/* The same for n (<= 24) tensors of one shape, one launch pair: dy_tm[i] -> db_dev[i]; ws: n * evmi_tm_colsum_bf16_ws_elems floats. */
int evmi_tm_colsum_batch_bf16(int n, const void* const* dy_tm, float* const* db_dev, float* ws_dev, long long ws_elems, long long rows, int C, int accumulate,
                              void* stream) {
  if (n <= 0 || n > TM_COLSUM_JOBS || !dy_tm || !db_dev || !ws_dev || rows <= 0 || C <= 0 || (C & 7) || C > 256)
    return fail(EVMI_ERR_INVALID_ARG, "tm_colsum_batch_bf16: bad arguments (1..24 tensors, C: multiple of 8, <= 256)");
  const long long nblk = (rows + COLSUM_ROWS - 1) / COLSUM_ROWS;
  if (ws_elems < (long long)n * nblk * C) return fail(EVMI_ERR_INVALID_ARG, "tm_colsum_batch_bf16: workspace too small");
  ColsumBatch b;
  for (int i = 0; i < n; ++i) {
    if (!dy_tm[i] || !db_dev[i]) return fail(EVMI_ERR_INVALID_ARG, "tm_colsum_batch_bf16: null tensor");
    b.dy[i] = reinterpret_cast<const bf16_t*>(dy_tm[i]);
    b.db[i] = db_dev[i];
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(tm_colsum_partial_batch_kernel, dim3((unsigned)nblk, n), dim3(256), 0, s, b, ws_dev, rows, C, (int)nblk);
  EVMI_LAUNCH_CHECK("tm_colsum_partial_batch");
  hipLaunchKernelGGL(tm_colsum_final_batch_kernel, dim3((C + 31) / 32, n), dim3(256), 0, s, b, ws_dev, (int)nblk, C, accumulate);
  EVMI_LAUNCH_CHECK("tm_colsum_final_batch");
  return EVMI_OK;
}

}  // extern "C"
