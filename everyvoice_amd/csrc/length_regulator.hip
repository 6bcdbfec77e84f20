// FastSpeech2 length regulator: out[b, t, :] = values[b, i(t), :] where token i is repeated
// max(0, d_i) times — the batched form of everyvoice/utils/heavy.py:12-21 `expand`.
// Pure integer/byte work, HBM-bound: one inclusive scan of the durations per batch item, then a
// gather in which each wavefront copies whole rows with 16-byte lanes (a [256] fp32 row is one
// 1-KiB wave instruction).  Results are bit-exact by construction (no arithmetic on the payload).
#include "common.h"

namespace evmi {

constexpr int SCAN_THREADS = 256;

// cum[b][i] = sum_{i' <= i} max(0, d[b][i'])   (int32; totals are bounded by t_max in practice)
__global__ __launch_bounds__(SCAN_THREADS) void lr_scan_kernel(const int64_t* __restrict__ dur,
                                                               int* __restrict__ cum,
                                                               int64_t* __restrict__ lens, int L,
                                                               int t_max) {
  __shared__ int wave_tot[SCAN_THREADS / 64];
  __shared__ int carry_s;
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int base = 0; base < L; base += SCAN_THREADS) {
    const int i = base + tid;
    long long d = 0;
    if (i < L) d = dur[(long long)b * L + i];
    int v = d > 0 ? (d > 0x3fffffff ? 0x3fffffff : (int)d) : 0;
    // inclusive wave scan
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      int n = __shfl_up(v, off, 64);
      if (lane >= off) v += n;
    }
    if (lane == 63) wave_tot[wave] = v;
    __syncthreads();
    int prefix = carry_s;
    for (int w = 0; w < wave; ++w) prefix += wave_tot[w];
    if (i < L) cum[(long long)b * L + i] = v + prefix;
    __syncthreads();
    if (tid == SCAN_THREADS - 1) carry_s = v + prefix;
    __syncthreads();
  }
  if (tid == 0 && lens) {
    const int total = carry_s;
    lens[b] = total < t_max ? total : t_max;
  }
}

// first i with cum[i] > t  (cum is non-decreasing); returns L if none
__device__ __forceinline__ int upper_bound(const int* __restrict__ cum, int L, int t) {
  int lo = 0, hi = L;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cum[mid] > t) hi = mid; else lo = mid + 1;
  }
  return lo;
}

template <typename VecT>
__global__ __launch_bounds__(256) void lr_gather_kernel(const VecT* __restrict__ values,
                                                        const int* __restrict__ cum,
                                                        VecT* __restrict__ out,
                                                        int* __restrict__ index, int L, int t_max,
                                                        int row_vecs, int frames_per_block) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int* cb = cum + (long long)b * L;
  const int t0 = blockIdx.x * frames_per_block;
  for (int f = wave; f < frames_per_block; f += 4) {
    const int t = t0 + f;
    if (t >= t_max) break;
    const int src = upper_bound(cb, L, t);
    VecT* dst = out + ((long long)b * t_max + t) * row_vecs;
    if (src < L) {
      const VecT* s = values + ((long long)b * L + src) * row_vecs;
      for (int v = lane; v < row_vecs; v += 64) dst[v] = s[v];
    } else {
      VecT z;
      __builtin_memset(&z, 0, sizeof(VecT));
      for (int v = lane; v < row_vecs; v += 64) dst[v] = z;
    }
    if (index && lane == 0) index[(long long)b * t_max + t] = src < L ? src : -1;
  }
}

// grad_values[b, i, :] = sum_{t in [cum[i-1], min(cum[i], t_max))} grad_out[b, t, :]
__global__ __launch_bounds__(256) void lr_bwd_kernel(const float* __restrict__ go,
                                                     const int* __restrict__ cum,
                                                     float* __restrict__ gv, int L, int D, int t_max) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + wave;
  if (i >= L) return;
  const int* cb = cum + (long long)b * L;
  int lo = i > 0 ? cb[i - 1] : 0;
  int hi = cb[i];
  if (hi > t_max) hi = t_max;
  for (int c = lane; c < D; c += 64) {
    float acc = 0.f;
    for (int t = lo; t < hi; ++t) acc += go[((long long)b * t_max + t) * D + c];
    gv[((long long)b * L + i) * D + c] = acc;
  }
}

struct LrWorkspace {
  int* cum = nullptr;
  size_t cap = 0;
};

static int ensure_cum(LrWorkspace& ws, size_t n) {
  if (ws.cap >= n) return EVMI_OK;
  if (ws.cum) (void)hipFree(ws.cum);
  ws.cum = nullptr;
  ws.cap = 0;
  EVMI_HIP_CHECK(hipMalloc((void**)&ws.cum, n * sizeof(int)));
  ws.cap = n;
  return EVMI_OK;
}

static thread_local LrWorkspace g_lr_ws_dev[kMaxDevices];  // (scratch belongs to the device it was allocated on)
#define g_lr_ws g_lr_ws_dev[device_slot()]

int length_regulate(const void* values, const int64_t* dur, void* out, int64_t* out_lens,
                    int32_t* index, int B, int L, int D, int t_max, int elem_bytes, hipStream_t s) {
  if (B < 0 || L < 0 || D < 0 || t_max < 0 || (elem_bytes != 2 && elem_bytes != 4))
    return fail(EVMI_ERR_INVALID_ARG, "length_regulate: bad shape / elem_bytes");
  if (B == 0 || t_max == 0 || D == 0) {
    if (out_lens && B > 0) EVMI_HIP_CHECK(hipMemsetAsync(out_lens, 0, sizeof(int64_t) * B, s));
    return EVMI_OK;
  }
  if (L == 0) {
    EVMI_HIP_CHECK(hipMemsetAsync(out, 0, (size_t)B * t_max * D * elem_bytes, s));
    if (out_lens) EVMI_HIP_CHECK(hipMemsetAsync(out_lens, 0, sizeof(int64_t) * B, s));
    if (index) EVMI_HIP_CHECK(hipMemsetAsync(index, 0xff, sizeof(int32_t) * (size_t)B * t_max, s));
    return EVMI_OK;
  }
  int rc = ensure_cum(g_lr_ws, (size_t)B * L);
  if (rc) return rc;
  hipLaunchKernelGGL(lr_scan_kernel, dim3(B), dim3(SCAN_THREADS), 0, s, dur, g_lr_ws.cum, out_lens, L, t_max);
  EVMI_LAUNCH_CHECK("lr_scan");
  const size_t row_bytes = (size_t)D * elem_bytes;
  const int fpb = 16;
  dim3 grid((t_max + fpb - 1) / fpb, B);
  const bool a16 = row_bytes % 16 == 0 && ((uintptr_t)values % 16 == 0) && ((uintptr_t)out % 16 == 0);
  if (a16) {
    hipLaunchKernelGGL(lr_gather_kernel<uint4>, grid, dim3(256), 0, s, (const uint4*)values, g_lr_ws.cum,
                       (uint4*)out, index, L, t_max, (int)(row_bytes / 16), fpb);
  } else if (row_bytes % 4 == 0) {
    hipLaunchKernelGGL(lr_gather_kernel<uint32_t>, grid, dim3(256), 0, s, (const uint32_t*)values,
                       g_lr_ws.cum, (uint32_t*)out, index, L, t_max, (int)(row_bytes / 4), fpb);
  } else {
    hipLaunchKernelGGL(lr_gather_kernel<uint16_t>, grid, dim3(256), 0, s, (const uint16_t*)values,
                       g_lr_ws.cum, (uint16_t*)out, index, L, t_max, (int)(row_bytes / 2), fpb);
  }
  EVMI_LAUNCH_CHECK("lr_gather");
  return EVMI_OK;
}

int length_regulate_bwd_f32(const float* go, const int64_t* dur, float* gv, int B, int L, int D,
                            int t_max, hipStream_t s) {
  if (B < 0 || L < 0 || D < 0 || t_max < 0) return fail(EVMI_ERR_INVALID_ARG, "length_regulate_bwd: bad shape");
  if (B == 0 || L == 0 || D == 0) return EVMI_OK;
  int rc = ensure_cum(g_lr_ws, (size_t)B * L);
  if (rc) return rc;
  hipLaunchKernelGGL(lr_scan_kernel, dim3(B), dim3(SCAN_THREADS), 0, s, dur, g_lr_ws.cum,
                     (int64_t*)nullptr, L, t_max);
  EVMI_LAUNCH_CHECK("lr_scan");
  hipLaunchKernelGGL(lr_bwd_kernel, dim3((L + 3) / 4, B), dim3(256), 0, s, go, g_lr_ws.cum, gv, L, D, t_max);
  EVMI_LAUNCH_CHECK("lr_bwd");
  return EVMI_OK;
}

}  // namespace evmi
