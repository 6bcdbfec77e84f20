// Edge kernels of the discriminator chains (train/disc_chain.py): the layers of MPD / MSD (jik876 hifi-gan models.py DiscriminatorP /
// DiscriminatorS; reached by the reference through hfgl.model.HiFiGAN.training_step) whose shapes are not
// matrix-core shapes -- the one-input-channel first layers and the one-output-channel logit layers -- and the reductions around the
// chain (feature-matching L1, bias gradients), all on FLAT PACKED tensors:
//
//   a flat packed tensor: bf16, [C / 8 octet rows][units], a unit = the 8 channels of one position (16 bytes); n_items items laid
//   end to end T units apart, the first `valid` units of an item are data, everything else -- the gap behind every item, a front
//   guard and a tail guard around the row -- is zero and stays zero (kernels write valid units only).  The gap is the zero padding
//   of the NEXT convolution (right side of this item, left side of the next), so a convolution over the flat row needs no item
//   arithmetic: csrc/conv_cbt_bf16_pk.hip (evmi_conv_pkflat_*), csrc/conv_wgrad_bf16_pk.hip (evmi_conv_pkflat_wgrad).
//
// Every reduction here has a fixed summation order (partials per workgroup, added in index order by a second pass).
#include <algorithm>
#include <cstdint>

#include "common.h"
#include "conv_pk_common.h"

namespace evmi {

constexpr int DC_KMAX = 16;   // taps of a first layer (5 / 15 in the model)
constexpr int DC_PKMAX = 8;   // taps of a logit layer (3 in the model)

__device__ __forceinline__ void unpack8(const uint4& u, float* v) {
  v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
  v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w);
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 o;
  o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
  return o;
}

// the convolution input of item (b, c) of the period view at position h (reflection past the end of the waveform; 0 outside [0, H))
struct AudioView {
  const float* audio;
  int t_audio, period, H;
  __device__ __forceinline__ float at(int item, int h) const {
    if (h < 0 || h >= H) return 0.f;
    const int b = item / period, c = item - b * period;
    int t = h * period + c;
    if (t >= t_audio) t = 2 * (t_audio - 1) - t;
    return audio[(long long)b * t_audio + t];
  }
};

// ---- first layer, forward: y[o][item * Ts + to] = lrelu(bias + sum_j w[co][j] x(item, to * s + j - pad)) ----------------------------
struct FirstFwdArgs {
  AudioView xv;
  const float* w;
  const float* bias;
  uint4* y;
  long long plane;
  int Ts, n_out, n_items, c_out, k, stride, pad;
  float slope;
};
__global__ __launch_bounds__(256) void disc_first_fwd_kernel(FirstFwdArgs a) {
  extern __shared__ float wl[];  // [c_out][DC_KMAX] + [c_out]
  float* bl = wl + a.c_out * DC_KMAX;
  for (int v = threadIdx.x; v < a.c_out * DC_KMAX; v += 256) {
    const int co = v / DC_KMAX, j = v - co * DC_KMAX;
    wl[v] = j < a.k ? a.w[co * a.k + j] : 0.f;
  }
  for (int v = threadIdx.x; v < a.c_out; v += 256) bl[v] = a.bias ? a.bias[v] : 0.f;
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= a.n_items * a.n_out) return;
  const int item = n / a.n_out, to = n - item * a.n_out;
  float x[DC_KMAX];
#pragma unroll
  for (int j = 0; j < DC_KMAX; ++j) x[j] = j < a.k ? a.xv.at(item, to * a.stride + j - a.pad) : 0.f;
  uint4* dst = a.y + (long long)item * a.Ts + to;
  for (int o = 0; o < a.c_out / 8; ++o) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float s = bl[o * 8 + e];
      const float* wr = wl + (o * 8 + e) * DC_KMAX;
#pragma unroll
      for (int j = 0; j < DC_KMAX; ++j) s = fmaf(wr[j], x[j], s);
      v[e] = s > 0.f ? s : s * a.slope;
    }
    dst[(long long)o * a.plane] = pack8(v);
  }
}

// ---- first layer, weight + bias gradient: partial[blk][co][k + 1] (last column: sum of dy) -----------------------------------------
// columns per thread: chosen per call so that the launch is ~512 workgroups -- every workgroup ends with a fixed-order reduction of its
// 8 x (taps + 1) accumulators over its 256 threads (~12 us), which at 4 columns per thread was half of the kernel's time
static int first_wgrad_cols(long long n_total, int c_out) {
  const long long want = (n_total * (c_out / 8) + 256LL * 512 - 1) / (256LL * 512);
  return (int)std::min<long long>(32, std::max<long long>(4, want));
}
struct FirstWgradArgs {
  AudioView xv;
  const uint4* dy;
  long long plane;
  int T, n_out, n_items, c_out, k, stride, pad, wcols;
  float* partial;
};
template <int KT>  // taps held in registers (8: the period discriminators' k = 5; 16: the scale discriminators' k = 15)
__global__ __launch_bounds__(256) void disc_first_wgrad_kernel(FirstWgradArgs a) {
  // (The workgroup's 8 x (KT + 1) sums: wave sums by DPP, then the four waves in order.  The transposing reduction through
  // [8 * (KT + 1)][257] floats of LDS this replaces -- 140 KB at KT = 16 -- left ONE workgroup of four waves per CU for a loop of
  // gathered audio reads: 128 us per launch on the scale discriminators' first layers.)
  __shared__ float red[4][8 * (KT + 1)];
  const int tid = threadIdx.x, o = blockIdx.y;
  const int n_total = a.n_items * a.n_out;
  const int n0 = blockIdx.x * (256 * a.wcols);
  float acc[8][KT + 1];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j <= KT; ++j) acc[e][j] = 0.f;
  for (int i = 0; i < a.wcols; ++i) {
    const int n = n0 + i * 256 + tid;
    const bool live = n < n_total;
    const int nc = live ? n : n_total - 1;
    const int item = nc / a.n_out, to = nc - item * a.n_out;
    const uint4 du = a.dy[(long long)o * a.plane + (long long)item * a.T + to];
    float x[KT], dv[8];
#pragma unroll
    for (int j = 0; j < KT; ++j) x[j] = (live && j < a.k) ? a.xv.at(item, to * a.stride + j - a.pad) : 0.f;
    unpack8(du, dv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float d = live ? dv[e] : 0.f;
#pragma unroll
      for (int j = 0; j < KT; ++j) acc[e][j] = fmaf(d, x[j], acc[e][j]);
      acc[e][KT] += d;
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j <= KT; ++j) {
      const float v = wave_sum_dpp(acc[e][j]);
      if (lane == 0) red[wave][e * (KT + 1) + j] = v;
    }
  __syncthreads();
  if (tid < 8 * (KT + 1)) {
    const int e = tid / (KT + 1), j = tid - e * (KT + 1);
    if (j < a.k || j == KT) {
      const float v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];  // (wave order: fixed)
      a.partial[((long long)blockIdx.x * a.c_out + o * 8 + e) * (a.k + 1) + (j == KT ? a.k : j)] = v;
    }
  }
}
__global__ void disc_first_wgrad_final_kernel(const float* __restrict__ partial, float* __restrict__ dw, float* __restrict__ db, int c_out, int k,
                                              int nblk, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c_out * (k + 1)) return;
  const int co = i / (k + 1), j = i - co * (k + 1);
  if (j == k) {
    if (db) db[co] = ordered_sum_strided(partial + i, (long long)c_out * (k + 1), nblk, accumulate ? db[co] : 0.f);
  } else {
    dw[co * k + j] = ordered_sum_strided(partial + i, (long long)c_out * (k + 1), nblk, accumulate ? dw[co * k + j] : 0.f);
  }
}

// ---- first layer, input gradient (generator step): dx[item][h] = sum_co sum_j w[co][j] dy[co][item][(h + pad - j) / s] ------------
struct FirstDgradArgs {
  const uint4* dy;
  long long plane;
  int T, n_out, n_items, H, c_out, k, stride, pad;
  const float* w;
  float* dx;
};
__global__ __launch_bounds__(256) void disc_first_dgrad_kernel(FirstDgradArgs a) {
  extern __shared__ float wl[];  // [k][c_out]
  for (int v = threadIdx.x; v < a.c_out * a.k; v += 256) {
    const int co = v / a.k, j = v - co * a.k;
    wl[j * a.c_out + co] = a.w[v];
  }
  __syncthreads();
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= a.n_items * a.H) return;
  const int item = n / a.H, h = n - item * a.H;
  float acc = 0.f;
  for (int j = 0; j < a.k; ++j) {
    const int num = h + a.pad - j;
    if (num < 0 || num % a.stride) continue;
    const int to = num / a.stride;
    if (to >= a.n_out) continue;
    const uint4* src = a.dy + (long long)item * a.T + to;
    const float* wr = wl + j * a.c_out;
    for (int o = 0; o < a.c_out / 8; o += 4) {  // four units in flight
      uint4 u[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) u[q] = src[(long long)min(o + q, a.c_out / 8 - 1) * a.plane];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (o + q >= a.c_out / 8) break;
        float dv[8];
        unpack8(u[q], dv);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(wr[(o + q) * 8 + e], dv[e], acc);
      }
    }
  }
  a.dx[n] = acc;
}

// ---- logit layer (c_out = 1), forward: partial[chunk][n] = sum over the chunk's channels and the taps -----------------------------
struct PostFwdArgs {
  const uint4* x;
  long long plane;
  int T, n, n_items, C, k, pad, octs_per_chunk, nchunks;
  const float* w;     // [1][C][k]
  const float* bias;  // [1] or null
  float* partial;     // [nchunks][n_items * n]
  float* logits;      // [n_items][n]
};
__global__ __launch_bounds__(256) void disc_post_fwd_kernel(PostFwdArgs a) {
  extern __shared__ float lds[];
  float* wl = lds;                                      // [octs_per_chunk * 8][DC_PKMAX]
  float* red = lds + a.octs_per_chunk * 8 * DC_PKMAX;   // [4][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = blockIdx.y * a.octs_per_chunk;
  const int no = min(a.octs_per_chunk, a.C / 8 - o0);
  for (int v = tid; v < a.octs_per_chunk * 8 * DC_PKMAX; v += 256) {
    const int c = v / DC_PKMAX, j = v - c * DC_PKMAX;
    wl[v] = (c < no * 8 && j < a.k) ? a.w[(long long)(o0 * 8 + c) * a.k + j] : 0.f;
  }
  __syncthreads();
  const int n_total = a.n_items * a.n;
  const int n = blockIdx.x * 64 + lane;
  const bool live = n < n_total;
  const int nc = live ? n : 0;
  const int item = nc / a.n, to = nc - item * a.n;
  const uint4* src = a.x + (long long)item * a.T + to - a.pad;  // (units left of item 0 / right of the item: zero gaps and guards)
  float acc = 0.f;
  for (int o = wave; o < no; o += 4) {
    uint4 u[DC_PKMAX];
#pragma unroll
    for (int j = 0; j < DC_PKMAX; ++j) u[j] = src[(long long)(o0 + o) * a.plane + min(j, a.k - 1)];
#pragma unroll
    for (int j = 0; j < DC_PKMAX; ++j) {
      if (j >= a.k) break;
      float xv[8];
      unpack8(u[j], xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf(wl[(o * 8 + e) * DC_PKMAX + j], xv[e], acc);
    }
  }
  red[wave * 64 + lane] = acc;
  __syncthreads();
  if (wave == 0 && live) a.partial[(long long)blockIdx.y * n_total + n] = red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane];
}
__global__ void disc_post_fwd_final_kernel(PostFwdArgs a) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  const int n_total = a.n_items * a.n;
  if (n >= n_total) return;
  a.logits[n] = ordered_sum_strided(a.partial + n, n_total, a.nchunks, a.bias ? a.bias[0] : 0.f);
}

// ---- logit layer, input gradient: dx[c][item][u] = (sum_j w[c][j] dl[item][u + pad - j] + fm term) * lrelu'(mask) ------------------
struct PostDgradArgs {
  const float* dl;  // [n_items][n]
  const float* w;
  uint4* dx;
  long long plane;
  int Ts, n, n_items, C, k, pad;
  const uint4* mask;
  const uint4* fm;
  long long mplane;
  int Tm;
  float mask_slope, fm_scale;
};
__global__ __launch_bounds__(256) void disc_post_dgrad_kernel(PostDgradArgs a) {
  const int n = blockIdx.x * 256 + threadIdx.x, o = blockIdx.y;
  if (n >= a.n_items * a.n) return;
  const int item = n / a.n, u = n - item * a.n;
  float d[DC_PKMAX];
#pragma unroll
  for (int j = 0; j < DC_PKMAX; ++j) {
    const int to = u + a.pad - j;
    d[j] = (j < a.k && to >= 0 && to < a.n) ? a.dl[(long long)item * a.n + to] : 0.f;
  }
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float s = 0.f;
    const float* wr = a.w + (long long)(o * 8 + e) * a.k;
#pragma unroll
    for (int j = 0; j < DC_PKMAX; ++j)
      if (j < a.k) s = fmaf(wr[j], d[j], s);
    v[e] = s;
  }
  if (a.mask) {
    const long long mu = (long long)o * a.mplane + (long long)item * a.Tm + u;
    const uint4 mk = a.mask[mu];
    const uint4 fr = a.fm ? a.fm[mu] : mk;
    pk_flat_tail4(v, make_uint2(mk.x, mk.y), make_uint2(fr.x, fr.y), a.fm_scale, a.mask_slope);
    pk_flat_tail4(v + 4, make_uint2(mk.z, mk.w), make_uint2(fr.z, fr.w), a.fm_scale, a.mask_slope);
  }
  a.dx[(long long)o * a.plane + (long long)item * a.Ts + u] = pack8(v);
}

// ---- logit layer, weight gradient: partial[chunk][c][k] ----------------------------------------------------------------------------
constexpr int DC_PWCOLS = 4;
struct PostWgradArgs {
  const uint4* x;
  long long plane;
  int T, n, n_items, C, k, pad;
  const float* dl;
  float* partial;
};
__global__ __launch_bounds__(256) void disc_post_wgrad_kernel(PostWgradArgs a) {
  __shared__ float red[4][8 * DC_PKMAX];  // (wave sums by DPP, then the four waves in order: see disc_first_wgrad_kernel)
  const int tid = threadIdx.x, o = blockIdx.y;
  const int n_total = a.n_items * a.n;
  const int n0 = blockIdx.x * (256 * DC_PWCOLS);
  float acc[8][DC_PKMAX];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < DC_PKMAX; ++j) acc[e][j] = 0.f;
  for (int i = 0; i < DC_PWCOLS; ++i) {
    const int n = n0 + i * 256 + tid;
    const bool live = n < n_total;
    const int nc = live ? n : 0;
    const int item = nc / a.n, to = nc - item * a.n;
    const float d = live ? a.dl[nc] : 0.f;
    const uint4* src = a.x + (long long)o * a.plane + (long long)item * a.T + to - a.pad;
    uint4 u[DC_PKMAX];
#pragma unroll
    for (int j = 0; j < DC_PKMAX; ++j) u[j] = src[min(j, a.k - 1)];
#pragma unroll
    for (int j = 0; j < DC_PKMAX; ++j) {
      if (j >= a.k) break;
      float xv[8];
      unpack8(u[j], xv);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e][j] = fmaf(d, xv[e], acc[e][j]);
    }
  }
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int j = 0; j < DC_PKMAX; ++j) {
      const float v = wave_sum_dpp(acc[e][j]);
      if (lane == 0) red[wave][e * DC_PKMAX + j] = v;
    }
  __syncthreads();
  if (tid < 8 * DC_PKMAX) {
    const int e = tid / DC_PKMAX, j = tid - e * DC_PKMAX;
    if (j < a.k) a.partial[((long long)blockIdx.x * a.C + o * 8 + e) * a.k + j] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  }
}
__global__ void disc_ordered_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int n, int nblk, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = ordered_sum_strided(partial + i, n, nblk, accumulate ? out[i] : 0.f);
}

// ---- feature-matching L1 over pairs of packed tensors: out[0] += sum_l scale_l * sum |a_l - b_l| -----------------------------------
constexpr int DC_MAX_PAIRS = 12;
constexpr int DC_ABS_BLOCKS = 64;  // workgroups per pair
struct AbsdiffBatch {
  const uint4* a[DC_MAX_PAIRS];
  const uint4* b[DC_MAX_PAIRS];
  long long units[DC_MAX_PAIRS];
  long long plane[DC_MAX_PAIRS];
  int rows[DC_MAX_PAIRS];
  float scale[DC_MAX_PAIRS];
  int n;
  double* partial;  // [n][DC_ABS_BLOCKS]
  float* out;
};
__global__ __launch_bounds__(256) void pkflat_absdiff_kernel(AbsdiffBatch q) {
  __shared__ double sh[4];
  const int l = blockIdx.y;
  const long long n = q.units[l], step = (long long)DC_ABS_BLOCKS * 256;
  double acc = 0.0;
  for (int row = 0; row < q.rows[l]; ++row) {
    const uint4* pa = q.a[l] + (long long)row * q.plane[l];
    const uint4* pb = q.b[l] + (long long)row * q.plane[l];
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * step < n; i += 4 * step) {
      uint4 ua[4], ub[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { ua[u] = pa[i + u * step]; ub[u] = pb[i + u * step]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float x[8], y[8];
        unpack8(ua[u], x);
        unpack8(ub[u], y);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) s += fabsf(x[e] - y[e]);
        acc += (double)s;
      }
    }
    for (; i < n; i += step) {
      float x[8], y[8];
      unpack8(pa[i], x);
      unpack8(pb[i], y);
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += fabsf(x[e] - y[e]);
      acc += (double)s;
    }
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) q.partial[l * DC_ABS_BLOCKS + blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void pkflat_absdiff_final_kernel(AbsdiffBatch q) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double tot = 0.0;
  for (int l = 0; l < q.n; ++l) {
    double s = 0.0;
    for (int b = 0; b < DC_ABS_BLOCKS; ++b) s += q.partial[l * DC_ABS_BLOCKS + b];
    tot += s * (double)q.scale[l];
  }
  q.out[0] += (float)tot;
}

// ---- bias gradients: db[c] += sum over the units of row c / 8 --------------------------------------------------------------------
constexpr int DC_MAX_ROWJOBS = 12;
constexpr int DC_ROW_CHUNKS = 16;
struct RowsumBatch {
  const uint4* dy[DC_MAX_ROWJOBS];
  long long plane[DC_MAX_ROWJOBS];
  long long units[DC_MAX_ROWJOBS];  // per row (n_items * T)
  float* db[DC_MAX_ROWJOBS];
  int row_start[DC_MAX_ROWJOBS + 1];  // octet rows, prefix sums
  int n;
  float* partial;  // [total rows][DC_ROW_CHUNKS][8]
};
__global__ __launch_bounds__(256) void pkflat_rowsum_kernel(RowsumBatch q) {
  __shared__ float sh[4][8];
  const int row = blockIdx.y, chunk = blockIdx.x;
  int l = 0;
  while (l + 1 < q.n && row >= q.row_start[l + 1]) ++l;
  const int o = row - q.row_start[l];
  const long long n = q.units[l];
  const long long per = (n + DC_ROW_CHUNKS - 1) / DC_ROW_CHUNKS;
  const long long lo = chunk * per, hi = min(n, lo + per);
  const uint4* src = q.dy[l] + (long long)o * q.plane[l];
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  long long i = lo + threadIdx.x;
  for (; i + 3 * 256 < hi; i += 4 * 256) {
    uint4 u[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) u[k] = src[i + k * 256];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v[8];
      unpack8(u[k], v);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += v[e];
    }
  }
  for (; i < hi; i += 256) {
    float v[8];
    unpack8(src[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] += v[e];
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = acc[e];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][e] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) q.partial[((long long)row * DC_ROW_CHUNKS + chunk) * 8 + threadIdx.x] =
      sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}
__global__ void pkflat_rowsum_final_kernel(RowsumBatch q) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;  // (row, e)
  const int rows = q.row_start[q.n];
  if (i >= rows * 8) return;
  const int row = i >> 3, e = i & 7;
  int l = 0;
  while (l + 1 < q.n && row >= q.row_start[l + 1]) ++l;
  const int c = (row - q.row_start[l]) * 8 + e;
  q.db[l][c] = ordered_sum_strided(q.partial + (long long)row * DC_ROW_CHUNKS * 8 + e, 8, DC_ROW_CHUNKS, q.db[l][c]);
}

// zero-fill (a packed tensor's buffer, once at allocation; also the library's own fill for small scratch)
__global__ void pkflat_zero_kernel(uint4* p, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = make_uint4(0u, 0u, 0u, 0u);
}

}  // namespace evmi

using namespace evmi;

extern "C" {

int evmi_pkflat_zero(void* buf, long long n_units, void* stream) {
  if (!buf || n_units < 0) return fail(EVMI_ERR_INVALID_ARG, "pkflat_zero: bad arguments");
  if (n_units) hipLaunchKernelGGL(pkflat_zero_kernel, dim3((unsigned)((n_units + 255) / 256)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<uint4*>(buf), n_units);
  EVMI_LAUNCH_CHECK("pkflat_zero");
  return EVMI_OK;
}

static bool first_shape_ok(int c_out, int k, int period, int stride) { return c_out > 0 && c_out % 8 == 0 && c_out <= 512 && k > 0 && k <= DC_KMAX && period > 0 && stride > 0; }

int evmi_disc_first_fwd(const float* audio_dev, int n_audio, int t_audio, int period, const float* w_dev, const float* bias_dev, void* y_pk,
                        long long y_plane, int T_store, int n_out, int c_out, int k, int stride, int pad, float slope, void* stream) {
  if (!audio_dev || !w_dev || !y_pk) return fail(EVMI_ERR_INVALID_ARG, "disc_first_fwd: null pointer");
  if (!first_shape_ok(c_out, k, period, stride)) return fail(EVMI_ERR_UNSUPPORTED, "disc_first_fwd: shape (c_out a multiple of 8 up to 512, k <= 16)");
  FirstFwdArgs a;
  a.xv.audio = audio_dev; a.xv.t_audio = t_audio; a.xv.period = period; a.xv.H = (t_audio + period - 1) / period;
  a.w = w_dev; a.bias = bias_dev; a.y = reinterpret_cast<uint4*>(y_pk); a.plane = y_plane; a.Ts = T_store; a.n_out = n_out;
  a.n_items = n_audio * period; a.c_out = c_out; a.k = k; a.stride = stride; a.pad = pad; a.slope = slope;
  const long long n = (long long)a.n_items * n_out;
  const size_t lds = (size_t)c_out * (DC_KMAX + 1) * sizeof(float);
  hipLaunchKernelGGL(disc_first_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), lds, (hipStream_t)stream, a);
  EVMI_LAUNCH_CHECK("disc_first_fwd");
  return EVMI_OK;
}

long long evmi_disc_first_wgrad_ws_elems(int n_items, int n_out, int c_out, int k) {
  const long long n = (long long)n_items * n_out;
  const long long per = 256LL * first_wgrad_cols(n, c_out);
  return (n + per - 1) / per * c_out * (k + 1);
}

int evmi_disc_first_wgrad(const float* audio_dev, int n_audio, int t_audio, int period, const void* dy_pk, long long dy_plane, int T_dy, int n_out,
                          float* dw_dev, float* db_dev, float* ws_dev, long long ws_elems, int c_out, int k, int stride, int pad, int accumulate,
                          void* stream) {
  if (!audio_dev || !dy_pk || !dw_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "disc_first_wgrad: null pointer");
  if (!first_shape_ok(c_out, k, period, stride)) return fail(EVMI_ERR_UNSUPPORTED, "disc_first_wgrad: shape");
  const int n_items = n_audio * period;
  if (ws_elems < evmi_disc_first_wgrad_ws_elems(n_items, n_out, c_out, k)) return fail(EVMI_ERR_INVALID_ARG, "disc_first_wgrad: workspace too small");
  FirstWgradArgs a;
  a.xv.audio = audio_dev; a.xv.t_audio = t_audio; a.xv.period = period; a.xv.H = (t_audio + period - 1) / period;
  a.dy = reinterpret_cast<const uint4*>(dy_pk); a.plane = dy_plane; a.T = T_dy; a.n_out = n_out; a.n_items = n_items; a.c_out = c_out; a.k = k;
  a.stride = stride; a.pad = pad; a.partial = ws_dev;
  const long long n = (long long)n_items * n_out;
  a.wcols = first_wgrad_cols(n, c_out);
  const int nblk = (int)((n + 256LL * a.wcols - 1) / (256LL * a.wcols));
  const bool k8 = k <= 8;
  if (k8) hipLaunchKernelGGL(disc_first_wgrad_kernel<8>, dim3(nblk, c_out / 8), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(disc_first_wgrad_kernel<DC_KMAX>, dim3(nblk, c_out / 8), dim3(256), 0, (hipStream_t)stream, a);
  EVMI_LAUNCH_CHECK("disc_first_wgrad");
  const int nf = c_out * (k + 1);
  hipLaunchKernelGGL(disc_first_wgrad_final_kernel, dim3((nf + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws_dev, dw_dev, db_dev, c_out, k, nblk, accumulate);
  EVMI_LAUNCH_CHECK("disc_first_wgrad_final");
  return EVMI_OK;
}

int evmi_disc_first_dgrad(const void* dy_pk, long long dy_plane, int T_dy, int n_out, const float* w_dev, float* dx_dev, int n_items, int H, int c_out,
                          int k, int stride, int pad, void* stream) {
  if (!dy_pk || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "disc_first_dgrad: null pointer");
  if (!first_shape_ok(c_out, k, 1, stride)) return fail(EVMI_ERR_UNSUPPORTED, "disc_first_dgrad: shape");
  FirstDgradArgs a;
  a.dy = reinterpret_cast<const uint4*>(dy_pk); a.plane = dy_plane; a.T = T_dy; a.n_out = n_out; a.n_items = n_items; a.H = H; a.c_out = c_out; a.k = k;
  a.stride = stride; a.pad = pad; a.w = w_dev; a.dx = dx_dev;
  const long long n = (long long)n_items * H;
  hipLaunchKernelGGL(disc_first_dgrad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), (size_t)c_out * k * sizeof(float), (hipStream_t)stream, a);
  EVMI_LAUNCH_CHECK("disc_first_dgrad");
  return EVMI_OK;
}

static int post_chunks(int n_total, int C, int& octs_per_chunk) {
  const int nblk = (n_total + 63) / 64;
  int want = std::max(1, std::min((512 + nblk - 1) / nblk, C / 32));
  octs_per_chunk = ((C / 8) + want - 1) / want;
  octs_per_chunk = (octs_per_chunk + 3) & ~3;
  return ((C / 8) + octs_per_chunk - 1) / octs_per_chunk;
}

long long evmi_disc_post_fwd_ws_elems(int n_items, int n, int C) {
  int opc;
  const int nch = post_chunks(n_items * n, C, opc);
  return (long long)nch * n_items * n;
}

int evmi_disc_post_fwd(const void* x_pk, long long x_plane, int T_x, int n_items, int n, const float* w_dev, const float* bias_dev, float* logits_dev,
                       float* ws_dev, long long ws_elems, int C, int k, int pad, void* stream) {
  if (!x_pk || !w_dev || !logits_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "disc_post_fwd: null pointer");
  if (C % 8 || k <= 0 || k > DC_PKMAX || n_items <= 0 || n <= 0) return fail(EVMI_ERR_UNSUPPORTED, "disc_post_fwd: shape (C a multiple of 8, k <= 8)");
  PostFwdArgs a;
  a.x = reinterpret_cast<const uint4*>(x_pk); a.plane = x_plane; a.T = T_x; a.n = n; a.n_items = n_items; a.C = C; a.k = k; a.pad = pad;
  a.nchunks = post_chunks(n_items * n, C, a.octs_per_chunk);
  a.w = w_dev; a.bias = bias_dev; a.partial = ws_dev; a.logits = logits_dev;
  const int n_total = n_items * n;
  if (ws_elems < (long long)a.nchunks * n_total) return fail(EVMI_ERR_INVALID_ARG, "disc_post_fwd: workspace too small");
  const size_t lds = ((size_t)a.octs_per_chunk * 8 * DC_PKMAX + 256) * sizeof(float);
  hipLaunchKernelGGL(disc_post_fwd_kernel, dim3((n_total + 63) / 64, a.nchunks), dim3(256), lds, (hipStream_t)stream, a);
  EVMI_LAUNCH_CHECK("disc_post_fwd");
  hipLaunchKernelGGL(disc_post_fwd_final_kernel, dim3((n_total + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
  EVMI_LAUNCH_CHECK("disc_post_fwd_final");
  return EVMI_OK;
}

int evmi_disc_post_dgrad(const float* dlogits_dev, const float* w_dev, void* dx_pk, long long dx_plane, int T_store, int n_items, int n, int C, int k,
                         int pad, const void* mask_pk, const void* fm_pk, long long mask_plane, int T_mask, float mask_slope, float fm_scale,
                         void* stream) {
  if (!dlogits_dev || !w_dev || !dx_pk) return fail(EVMI_ERR_INVALID_ARG, "disc_post_dgrad: null pointer");
  if (C % 8 || k <= 0 || k > DC_PKMAX) return fail(EVMI_ERR_UNSUPPORTED, "disc_post_dgrad: shape");
  if (fm_pk && !mask_pk) return fail(EVMI_ERR_INVALID_ARG, "disc_post_dgrad: the feature-matching reference needs the mask tensor");
  PostDgradArgs a;
  a.dl = dlogits_dev; a.w = w_dev; a.dx = reinterpret_cast<uint4*>(dx_pk); a.plane = dx_plane; a.Ts = T_store; a.n = n; a.n_items = n_items; a.C = C;
  a.k = k; a.pad = pad; a.mask = reinterpret_cast<const uint4*>(mask_pk); a.fm = reinterpret_cast<const uint4*>(fm_pk); a.mplane = mask_plane;
  a.Tm = T_mask; a.mask_slope = mask_slope; a.fm_scale = fm_scale;
  hipLaunchKernelGGL(disc_post_dgrad_kernel, dim3((n_items * n + 255) / 256, C / 8), dim3(256), 0, (hipStream_t)stream, a);
  EVMI_LAUNCH_CHECK("disc_post_dgrad");
  return EVMI_OK;
}

long long evmi_disc_post_wgrad_ws_elems(int n_items, int n, int C, int k) {
  const long long nt = (long long)n_items * n;
  return (nt + 256 * DC_PWCOLS - 1) / (256 * DC_PWCOLS) * C * k;
}

int evmi_disc_post_wgrad(const void* x_pk, long long x_plane, int T_x, int n_items, int n, const float* dlogits_dev, float* dw_dev, float* ws_dev,
                         long long ws_elems, int C, int k, int pad, int accumulate, void* stream) {
  if (!x_pk || !dlogits_dev || !dw_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "disc_post_wgrad: null pointer");
  if (C % 8 || k <= 0 || k > DC_PKMAX) return fail(EVMI_ERR_UNSUPPORTED, "disc_post_wgrad: shape");
  if (ws_elems < evmi_disc_post_wgrad_ws_elems(n_items, n, C, k)) return fail(EVMI_ERR_INVALID_ARG, "disc_post_wgrad: workspace too small");
  PostWgradArgs a;
  a.x = reinterpret_cast<const uint4*>(x_pk); a.plane = x_plane; a.T = T_x; a.n = n; a.n_items = n_items; a.C = C; a.k = k; a.pad = pad;
  a.dl = dlogits_dev; a.partial = ws_dev;
  const int nblk = (n_items * n + 256 * DC_PWCOLS - 1) / (256 * DC_PWCOLS);
  hipLaunchKernelGGL(disc_post_wgrad_kernel, dim3(nblk, C / 8), dim3(256), 0, (hipStream_t)stream, a);
  EVMI_LAUNCH_CHECK("disc_post_wgrad");
  hipLaunchKernelGGL(disc_ordered_final_kernel, dim3((C * k + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws_dev, dw_dev, C * k, nblk, accumulate);
  EVMI_LAUNCH_CHECK("disc_post_wgrad_final");
  return EVMI_OK;
}

long long evmi_pkflat_absdiff_ws_elems(int n_pairs) { return (long long)n_pairs * DC_ABS_BLOCKS * 2; }

int evmi_pkflat_absdiff(int n_pairs, const evmi_pkflat_pair* pairs, float* out_dev, float* ws_dev, long long ws_elems, void* stream) {
  if (n_pairs <= 0 || n_pairs > DC_MAX_PAIRS || !pairs || !out_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "pkflat_absdiff: bad arguments (at most 12 pairs)");
  if (ws_elems < evmi_pkflat_absdiff_ws_elems(n_pairs) || (reinterpret_cast<uintptr_t>(ws_dev) & 7)) return fail(EVMI_ERR_INVALID_ARG, "pkflat_absdiff: workspace too small or unaligned");
  AbsdiffBatch q;
  q.n = n_pairs; q.partial = reinterpret_cast<double*>(ws_dev); q.out = out_dev;
  for (int l = 0; l < n_pairs; ++l) {
    if (!pairs[l].a || !pairs[l].b || pairs[l].units < 0) return fail(EVMI_ERR_INVALID_ARG, "pkflat_absdiff: null tensor");
    q.a[l] = reinterpret_cast<const uint4*>(pairs[l].a); q.b[l] = reinterpret_cast<const uint4*>(pairs[l].b);
    q.units[l] = pairs[l].units; q.scale[l] = pairs[l].scale; q.plane[l] = pairs[l].plane; q.rows[l] = pairs[l].rows > 0 ? pairs[l].rows : 1;
  }
  hipLaunchKernelGGL(pkflat_absdiff_kernel, dim3(DC_ABS_BLOCKS, n_pairs), dim3(256), 0, (hipStream_t)stream, q);
  EVMI_LAUNCH_CHECK("pkflat_absdiff");
  hipLaunchKernelGGL(pkflat_absdiff_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, q);
  EVMI_LAUNCH_CHECK("pkflat_absdiff_final");
  return EVMI_OK;
}

long long evmi_pkflat_rowsum_ws_elems(int n_jobs, const evmi_pkflat_rows* jobs) {
  long long rows = 0;
  for (int l = 0; l < n_jobs; ++l) rows += jobs[l].C / 8;
  return rows * DC_ROW_CHUNKS * 8;
}

int evmi_pkflat_rowsum(int n_jobs, const evmi_pkflat_rows* jobs, float* ws_dev, long long ws_elems, void* stream) {
  if (n_jobs <= 0 || n_jobs > DC_MAX_ROWJOBS || !jobs || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "pkflat_rowsum: bad arguments (at most 12 tensors)");
  if (ws_elems < evmi_pkflat_rowsum_ws_elems(n_jobs, jobs)) return fail(EVMI_ERR_INVALID_ARG, "pkflat_rowsum: workspace too small");
  RowsumBatch q;
  q.n = n_jobs; q.partial = ws_dev; q.row_start[0] = 0;
  for (int l = 0; l < n_jobs; ++l) {
    if (!jobs[l].dy || !jobs[l].db || jobs[l].C % 8 || jobs[l].C <= 0) return fail(EVMI_ERR_INVALID_ARG, "pkflat_rowsum: bad tensor");
    q.dy[l] = reinterpret_cast<const uint4*>(jobs[l].dy); q.plane[l] = jobs[l].plane; q.units[l] = jobs[l].units; q.db[l] = jobs[l].db;
    q.row_start[l + 1] = q.row_start[l] + jobs[l].C / 8;
  }
  const int rows = q.row_start[n_jobs];
  hipLaunchKernelGGL(pkflat_rowsum_kernel, dim3(DC_ROW_CHUNKS, rows), dim3(256), 0, (hipStream_t)stream, q);
  EVMI_LAUNCH_CHECK("pkflat_rowsum");
  hipLaunchKernelGGL(pkflat_rowsum_final_kernel, dim3((rows * 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream, q);
  EVMI_LAUNCH_CHECK("pkflat_rowsum_final");
  return EVMI_OK;
}

}  // extern "C"
