// Direct (non matrix-core) fp32 convolutions for the two degenerate GEMM shapes of the discriminators in the
// channel-major training layout x[c][b][t]:
//
//   * few output channels (c_out <= 4: the logit convolutions 1024 -> 1, and the input gradients of every
//     first layer): a GEMV -- per column a dot product over (channel, tap).  Bound by reading x once.
//   * one input channel (c_in == 1: the first layers 1 -> 32 / 1 -> 128, and the input gradients of the logit
//     convolutions): an outer product -- per column k multiplies per output channel.  Bound by writing y once.
//
// Same contract as the matrix-core kernel (evmi_conv1d_cbt_f32): columns are the flattened (b, to) index,
// outputs may be placed on a strided grid (out_stride / out_offset) and accumulated.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "conv_cbt_direct.h"

namespace evmi {

__device__ __forceinline__ float direct_act(float v, int act, float p) {
  if (act == 1) return v > 0.f ? v : v * p;
  if (act == 2) return v / (1.f + expf(-v));
  if (act == 3) return fmaxf(v, 0.f);
  if (act == 4) return tanhf(v);
  return v;
}

constexpr int SMALLCO_MAX = 4;

// grid (ceil(N / 64), nchunks), 256 threads: lane = column, wave = quarter of the workgroup's channels
__global__ __launch_bounds__(256) void conv_smallco_kernel(ConvDirectArgs a) {
  extern __shared__ float lds[];
  float* wl = lds;                                   // [c_out][cc][k]
  float* red = lds + a.c_out * a.cc * a.k;           // [4][SMALLCO_MAX][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c0 = blockIdx.y * a.cc;
  const int cc = min(a.cc, a.c_in - c0);
  for (int v = tid; v < a.c_out * a.cc * a.k; v += 256) {
    const int co = v / (a.cc * a.k), r = v - co * (a.cc * a.k), c = r / a.k, j = r - c * a.k;
    wl[v] = c < cc ? a.w[((long long)co * a.c_in + c0 + c) * a.k + j] : 0.f;
  }
  __syncthreads();
  const long long n_total = (long long)a.B * a.n_out;
  const long long n = (long long)blockIdx.x * 64 + lane;
  const bool live = n < n_total;
  const int b = live ? (int)(n / a.n_out) : 0;
  const int to = live ? (int)(n - (long long)b * a.n_out) : 0;
  const int ti0 = to * a.stride - a.pad;
  float acc[SMALLCO_MAX] = {0.f, 0.f, 0.f, 0.f};
  const int per_wave = (cc + 3) >> 2;
  const int cb = wave * per_wave, ce = min(cc, cb + per_wave);
  // four channels per trip, their reads unconditional from clamped positions and issued together (a read under `live && in range`
  // is a branch + vmcnt(0) drain per (channel, tap): 48 dependent round trips for 16 channels x 3 taps)
  const long long cs = (long long)a.B * a.t_in;
  const float* xb = a.x + ((long long)c0 * a.B + b) * a.t_in;
  for (int c = cb; c < ce; c += 4) {
    for (int j = 0; j < a.k; ++j) {
      const int ti = ti0 + j * a.dil;
      const bool ok = live && ti >= 0 && ti < a.t_in;
      const int tic = min(max(ti, 0), a.t_in - 1);
      float xv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) xv[u] = xb[(long long)min(c + u, ce - 1) * cs + tic];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float x1 = (ok && c + u < ce) ? xv[u] : 0.f;
        const int cw = min(c + u, ce - 1);
#pragma unroll
        for (int co = 0; co < SMALLCO_MAX; ++co)
          if (co < a.c_out) acc[co] = fmaf(wl[(co * a.cc + cw) * a.k + j], x1, acc[co]);
      }
    }
  }
#pragma unroll
  for (int co = 0; co < SMALLCO_MAX; ++co) red[(wave * SMALLCO_MAX + co) * 64 + lane] = acc[co];
  __syncthreads();
  if (wave != 0 || !live) return;
  for (int co = 0; co < a.c_out; ++co) {
    float v = red[co * 64 + lane] + red[(SMALLCO_MAX + co) * 64 + lane] + red[(2 * SMALLCO_MAX + co) * 64 + lane] +
              red[(3 * SMALLCO_MAX + co) * 64 + lane];
    if (a.nchunks > 1) {
      a.partial[((long long)blockIdx.y * a.c_out + co) * n_total + n] = v;
    } else {
      if (a.bias) v += a.bias[co];
      v = direct_act(v, a.act, a.act_param);
      float* dst = a.y + ((long long)co * a.B + b) * a.t_out_total + (long long)to * a.out_stride + a.out_offset;
      *dst = a.accumulate ? *dst + v : v;
    }
  }
}

// y = bias + sum over channel chunks (fixed order: deterministic)
__global__ void conv_smallco_reduce_kernel(ConvDirectArgs a) {
  const long long n_total = (long long)a.B * a.n_out;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_total * a.c_out) return;
  const int co = (int)(i / n_total);
  const long long n = i - (long long)co * n_total;
  float v = ordered_sum_strided(a.partial + (long long)co * n_total + n, (long long)a.c_out * n_total, a.nchunks, a.bias ? a.bias[co] : 0.f);
  v = direct_act(v, a.act, a.act_param);
  const int b = (int)(n / a.n_out);
  const int to = (int)(n - (long long)b * a.n_out);
  float* dst = a.y + ((long long)co * a.B + b) * a.t_out_total + (long long)to * a.out_stride + a.out_offset;
  *dst = a.accumulate ? *dst + v : v;
}

constexpr int CIN1_KMAX = 16;
constexpr int CIN1_CO = 32;  // output channels per workgroup

// grid (ceil(N / 256), ceil(c_out / 32)): thread = column; its k inputs stay in registers for 32 output channels
__global__ __launch_bounds__(256) void conv_cin1_kernel(ConvDirectArgs a) {
  __shared__ float wl[CIN1_CO * CIN1_KMAX];
  __shared__ float bl[CIN1_CO];
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * CIN1_CO;
  const int nco = min(CIN1_CO, a.c_out - co0);
  for (int v = tid; v < CIN1_CO * CIN1_KMAX; v += 256) {
    const int co = v / CIN1_KMAX, j = v - co * CIN1_KMAX;
    wl[v] = (co < nco && j < a.k) ? a.w[(long long)(co0 + co) * a.k + j] : 0.f;
  }
  if (tid < CIN1_CO) bl[tid] = (a.bias && tid < nco) ? a.bias[co0 + tid] : 0.f;
  __syncthreads();
  const long long n_total = (long long)a.B * a.n_out;
  const long long n = (long long)blockIdx.x * 256 + tid;
  if (n >= n_total) return;
  const int b = (int)(n / a.n_out);
  const int to = (int)(n - (long long)b * a.n_out);
  const float* xr = a.x + (long long)b * a.t_in;
  const int ti0 = to * a.stride - a.pad;
  float xv[CIN1_KMAX];
#pragma unroll
  for (int j = 0; j < CIN1_KMAX; ++j) xv[j] = xr[min(max(ti0 + min(j, a.k - 1) * a.dil, 0), a.t_in - 1)];  // all requested, then masked
#pragma unroll
  for (int j = 0; j < CIN1_KMAX; ++j) {
    const int ti = ti0 + j * a.dil;
    xv[j] = (j < a.k && ti >= 0 && ti < a.t_in) ? xv[j] : 0.f;
  }
  float* dst = a.y + ((long long)co0 * a.B + b) * a.t_out_total + (long long)to * a.out_stride + a.out_offset;
  const long long co_stride = (long long)a.B * a.t_out_total;
  for (int co = 0; co < nco; co += 8) {  // previous values of 8 rows per trip (accumulate), from clamped rows
    float prev[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) prev[e] = -0.f;
    if (a.accumulate) {
#pragma unroll
      for (int e = 0; e < 8; ++e) prev[e] = dst[(long long)min(co + e, nco - 1) * co_stride];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (co + e >= nco) break;
      float v = bl[co + e];
#pragma unroll
      for (int j = 0; j < CIN1_KMAX; ++j) v = fmaf(wl[(co + e) * CIN1_KMAX + j], xv[j], v);
      v = direct_act(v, a.act, a.act_param);
      dst[(long long)(co + e) * co_stride] = prev[e] + v;
    }
  }
}

// ---- weight gradient of the one-input-channel layers (1 -> 32 k5 s3 of every period discriminator, 1 -> 128 k15 of every scale
// discriminator):  dw[co][j] = sum_n dy[co][n] * x[b(n)][to(n) * stride + j * dil - pad].  As a GEMM this is 128 x 15 outputs over a
// 131 k-long contraction: one tile, so the matrix-core kernel splits it 64 ways at most, copies dy into its padded layout first and
// takes 150-340 us for 67 MB of dy.  Here: thread = column (coalesced rows of dy), 8 output channels x k taps of accumulators per
// thread, the k inputs of a column loaded once for the 8 channels; the workgroup's accumulators meet in the LDS
// ([value][thread], one padded row per value) and leave as one partial per workgroup; a second pass adds the partials in index order.
constexpr int WCIN1_CO = 8;        // output channels per workgroup
constexpr int WCIN1_COLS = 16;     // columns per thread

struct WgradCin1Args {
  const float* x;    // [B][t_in]
  const float* dy;   // [c_out][B][n_out]
  float* partial;    // [nblk][c_out][k]
  int B, t_in, n_out, c_out, k, stride, dil, pad;
};

__global__ __launch_bounds__(256) void wgrad_cin1_kernel(WgradCin1Args a) {
  extern __shared__ float red[];  // [WCIN1_CO * CIN1_KMAX][257]
  const int tid = threadIdx.x;
  const int co0 = blockIdx.y * WCIN1_CO;
  const long long n_total = (long long)a.B * a.n_out;
  const long long n0 = (long long)blockIdx.x * (256 * WCIN1_COLS);
  float acc[WCIN1_CO][CIN1_KMAX];
#pragma unroll
  for (int e = 0; e < WCIN1_CO; ++e)
#pragma unroll
    for (int j = 0; j < CIN1_KMAX; ++j) acc[e][j] = 0.f;
  const long long row = n_total;  // elements between channels of dy
  for (int i = 0; i < WCIN1_COLS; ++i) {
    const long long n = n0 + (long long)i * 256 + tid;
    const bool live = n < n_total;
    const long long nc = live ? n : n_total - 1;
    const int b = (int)(nc / a.n_out);
    const int to = (int)(nc - (long long)b * a.n_out);
    const float* xr = a.x + (long long)b * a.t_in;
    const int ti0 = to * a.stride - a.pad;
    float xv[CIN1_KMAX], dv[WCIN1_CO];
#pragma unroll
    for (int j = 0; j < CIN1_KMAX; ++j) xv[j] = xr[min(max(ti0 + min(j, a.k - 1) * a.dil, 0), a.t_in - 1)];  // requested together, masked below
#pragma unroll
    for (int e = 0; e < WCIN1_CO; ++e) dv[e] = a.dy[(long long)min(co0 + e, a.c_out - 1) * row + nc];
#pragma unroll
    for (int j = 0; j < CIN1_KMAX; ++j) {
      const int ti = ti0 + j * a.dil;
      xv[j] = (live && j < a.k && ti >= 0 && ti < a.t_in) ? xv[j] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < WCIN1_CO; ++e)
#pragma unroll
      for (int j = 0; j < CIN1_KMAX; ++j) acc[e][j] = fmaf(dv[e], xv[j], acc[e][j]);
  }
#pragma unroll
  for (int e = 0; e < WCIN1_CO; ++e)
#pragma unroll
    for (int j = 0; j < CIN1_KMAX; ++j) red[(e * CIN1_KMAX + j) * 257 + tid] = acc[e][j];
  __syncthreads();
  if (tid < WCIN1_CO * CIN1_KMAX) {
    const int e = tid / CIN1_KMAX, j = tid - e * CIN1_KMAX;
    if (co0 + e < a.c_out && j < a.k) {
      const float* r = red + tid * 257;
      float v = 0.f;
      for (int t = 0; t < 256; ++t) v += r[t];  // fixed order
      a.partial[((long long)blockIdx.x * a.c_out + co0 + e) * a.k + j] = v;
    }
  }
}

__global__ void wgrad_cin1_final_kernel(const float* __restrict__ partial, float* __restrict__ dw, int n, int nblk, int accumulate) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dw[i] = ordered_sum_strided(partial + i, n, nblk, accumulate ? dw[i] : 0.f);
}

long long wgrad_cin1_plan(int B, int c_in, int n_out, int c_out, int k, int groups) {
  static const bool disabled = getenv("EVMI_NO_DIRECT_CONV") != nullptr;
  if (disabled || c_in != 1 || groups != 1 || k > CIN1_KMAX || c_out > 4096) return 0;
  const long long n_total = (long long)B * n_out;
  const long long nblk = (n_total + 256 * WCIN1_COLS - 1) / (256 * WCIN1_COLS);
  return nblk * c_out * k;
}

int launch_wgrad_cin1(const float* x, const float* dy, float* dw, float* ws, long long ws_elems, int B, int t_in, int n_out, int c_out, int k,
                      int stride, int pad, int dil, int accumulate, hipStream_t stream) {
  const long long need = wgrad_cin1_plan(B, 1, n_out, c_out, k, 1);
  if (need == 0) return fail(EVMI_ERR_UNSUPPORTED, "wgrad_cin1: not a one-input-channel shape");
  if (!ws || ws_elems < need) return fail(EVMI_ERR_INVALID_ARG, "wgrad_cin1: workspace missing or too small");
  const long long n_total = (long long)B * n_out;
  const int nblk = (int)((n_total + 256 * WCIN1_COLS - 1) / (256 * WCIN1_COLS));
  WgradCin1Args a{x, dy, ws, B, t_in, n_out, c_out, k, stride, dil, pad};
  const size_t lds = (size_t)WCIN1_CO * CIN1_KMAX * 257 * sizeof(float);
  static thread_local bool configured[kMaxDevices] = {};
  if (!configured[device_slot()]) {
    EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_cin1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    configured[device_slot()] = true;
  }
  hipLaunchKernelGGL(wgrad_cin1_kernel, dim3(nblk, (c_out + WCIN1_CO - 1) / WCIN1_CO), dim3(256), lds, stream, a);
  EVMI_LAUNCH_CHECK("wgrad_cin1");
  const int n = c_out * k;
  hipLaunchKernelGGL(wgrad_cin1_final_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, ws, dw, n, nblk, accumulate);
  EVMI_LAUNCH_CHECK("wgrad_cin1_final");
  return EVMI_OK;
}

long long conv_direct_plan(const ConvDirectArgs& in, int groups, int& cc, int& nchunks) {
  cc = nchunks = 0;
  static const bool disabled = getenv("EVMI_NO_DIRECT_CONV") != nullptr;  // A/B against the matrix-core kernel
  if (groups != 1 || disabled) return 0;
  if (in.c_in == 1 && in.k <= CIN1_KMAX) return 1;
  if (in.c_out > SMALLCO_MAX) return 0;
  const long long n_total = (long long)in.B * in.n_out;
  const long long nblk = (n_total + 63) / 64;
  // enough workgroups for the chip: split the channels when there are few columns; weights of a chunk fit 32 KB of LDS
  long long want = std::max<long long>(1, std::min<long long>((512 + nblk - 1) / nblk, (in.c_in + 15) / 16));
  cc = (int)((in.c_in + want - 1) / want);
  const int cc_lds = std::max(4, 8192 / (in.k * in.c_out));
  cc = std::max(4, std::min(cc, cc_lds));
  cc = (cc + 3) & ~3;
  nchunks = (in.c_in + cc - 1) / cc;
  return 1 + (nchunks > 1 ? (long long)nchunks * in.c_out * n_total : 0);
}

int launch_conv_direct(ConvDirectArgs a, int groups, float* ws, long long ws_elems, hipStream_t stream) {
  int cc, nchunks;
  const long long need = conv_direct_plan(a, groups, cc, nchunks);
  if (need == 0) return fail(EVMI_ERR_UNSUPPORTED, "conv_direct: not a direct-kernel shape");
  const long long n_total = (long long)a.B * a.n_out;
  if (a.c_in == 1 && a.k <= CIN1_KMAX) {
    dim3 grid((unsigned)((n_total + 255) / 256), (a.c_out + CIN1_CO - 1) / CIN1_CO);
    hipLaunchKernelGGL(conv_cin1_kernel, grid, dim3(256), 0, stream, a);
    EVMI_LAUNCH_CHECK("conv_cin1");
    return EVMI_OK;
  }
  a.cc = cc;
  a.nchunks = nchunks;
  a.partial = ws;
  if (nchunks > 1 && (!ws || ws_elems < need - 1)) return fail(EVMI_ERR_INVALID_ARG, "conv_direct: workspace missing or too small");
  const size_t lds = ((size_t)a.c_out * cc * a.k + 4 * SMALLCO_MAX * 64) * sizeof(float);
  dim3 grid((unsigned)((n_total + 63) / 64), nchunks);
  hipLaunchKernelGGL(conv_smallco_kernel, grid, dim3(256), lds, stream, a);
  EVMI_LAUNCH_CHECK("conv_smallco");
  if (nchunks > 1) {
    const long long total = n_total * a.c_out;
    hipLaunchKernelGGL(conv_smallco_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, a);
    EVMI_LAUNCH_CHECK("conv_smallco_reduce");
  }
  return EVMI_OK;
}

}  // namespace evmi
