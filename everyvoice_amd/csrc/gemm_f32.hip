// General fp32 GEMM on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains), row-major:
//
//     C[b][M][N] = alpha * op(A[b]) . op(B[b]) + beta * C[b]        op(A) is M x K, op(B) is K x N, any leading dimensions,
//                                                                   `batch` problems at fixed element strides
//
// The plain-GEMM shapes of the training paths that have no specialised kernel -- the mel / multi-resolution STFT loss DFTs and
// mel projections with their adjoints, the dense fp32 weight-gradient fallback (dY . col^T, split over K by the caller), the
// iSTFT head's transposed convolution, the aligner's per-item products -- run here; nothing in libevmi_hip links a BLAS.
//
// Tile: 64 x 64 outputs per 256-thread workgroup, one 32 x 32 accumulator per wave, K in chunks of 32 staged through LDS as
// As[k][m] / Bs[k][n] (so both MFMA operands are conflict-free row reads), the next chunk's global loads issued into registers
// before the current chunk's MFMAs.  Loads are 16-byte vectors wherever the source is aligned and inside the matrix, guarded
// scalars (zero fill) otherwise, so every shape and every sub-matrix view is accepted.
#include "common.h"

namespace evmi {

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K, lda, ldb, ldc;
  long long sa, sb, sc;
  float alpha, beta;
};

constexpr int GEMM_KC = 32, GEMM_LD = 68;

// Source with the contraction index contiguous (A not transposed: A[r * ld + k]; B transposed: B[r * ld + k]):
// thread -> row r = tid / 4, 8 consecutive k
__device__ __forceinline__ void gemm_load_kcontig(const float* __restrict__ src, int ld, int r0, int rows, int k0, int K, int tid,
                                                  float (&v)[8]) {
  const int r = r0 + (tid >> 2), k = k0 + (tid & 3) * 8;
  const float* p = src + (long long)r * ld + k;
  if (r < rows && k + 8 <= K && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (r < rows && k + e < K) ? p[e] : 0.f;
  }
}
__device__ __forceinline__ void gemm_store_kcontig(float* __restrict__ Xs, int tid, const float (&v)[8]) {
  const int r = tid >> 2, k = (tid & 3) * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) Xs[(k + e) * GEMM_LD + r] = v[e];
}
// Source with the output index contiguous (A transposed: A[k * ld + c]; B not transposed: B[k * ld + c]):
// thread -> k = tid / 8, 8 consecutive columns
__device__ __forceinline__ void gemm_load_ccontig(const float* __restrict__ src, int ld, int c0, int cols, int k0, int K, int tid,
                                                  float (&v)[8]) {
  const int k = k0 + (tid >> 3), c = c0 + (tid & 7) * 8;
  const float* p = src + (long long)k * ld + c;
  if (k < K && c + 8 <= cols && ((reinterpret_cast<uintptr_t>(p) & 15) == 0)) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (k < K && c + e < cols) ? p[e] : 0.f;
  }
}
__device__ __forceinline__ void gemm_store_ccontig(float* __restrict__ Xs, int tid, const float (&v)[8]) {
  float* d = Xs + (tid >> 3) * GEMM_LD + (tid & 7) * 8;
  *reinterpret_cast<float4*>(d) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(d + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmArgs g) {
  __shared__ __attribute__((aligned(16))) float As[GEMM_KC * GEMM_LD];
  __shared__ __attribute__((aligned(16))) float Bs[GEMM_KC * GEMM_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int i = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
  const float* A = g.A + (long long)blockIdx.z * g.sa;
  const float* B = g.B + (long long)blockIdx.z * g.sb;
  float* C = g.C + (long long)blockIdx.z * g.sc;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float va[8], vb[8];
  auto fetch = [&](int k0) {
    if (TA) gemm_load_ccontig(A, g.lda, m0, g.M, k0, g.K, tid, va);
    else gemm_load_kcontig(A, g.lda, m0, g.M, k0, g.K, tid, va);
    if (TB) gemm_load_kcontig(B, g.ldb, n0, g.N, k0, g.K, tid, vb);
    else gemm_load_ccontig(B, g.ldb, n0, g.N, k0, g.K, tid, vb);
  };
  if (g.K > 0) fetch(0);
  for (int k0 = 0; k0 < g.K; k0 += GEMM_KC) {
    __syncthreads();  // the previous chunk's fragment reads are done
    if (TA) gemm_store_ccontig(As, tid, va); else gemm_store_kcontig(As, tid, va);
    if (TB) gemm_store_kcontig(Bs, tid, vb); else gemm_store_ccontig(Bs, tid, vb);
    __syncthreads();
    if (k0 + GEMM_KC < g.K) fetch(k0 + GEMM_KC);  // in flight during the MFMAs below
    const float* ar = As + kh * GEMM_LD + wm * 32 + i;
    const float* br = Bs + kh * GEMM_LD + wn * 32 + i;
#pragma unroll
    for (int kk = 0; kk < GEMM_KC; kk += 2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ar[kk * GEMM_LD], br[kk * GEMM_LD], acc, 0, 0, 0);
  }
  // D layout: lane column = n, registers = rows (r & 3) + 8 * (r >> 2) + 4 * kh
  const int n = n0 + wn * 32 + i;
  if (n < g.N) {
    // the 16 previous values (beta != 0) are requested together from clamped rows, not one drained read per element
    float* dst[16];
    float prev[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      dst[r] = C + (long long)min(m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, g.M - 1) * g.ldc + n;
      prev[r] = 0.f;
    }
    if (g.beta != 0.f) {
#pragma unroll
      for (int r = 0; r < 16; ++r) prev[r] = *dst[r];
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = g.alpha * acc[r];
      if (m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh < g.M) *dst[r] = g.beta == 0.f ? v : g.beta * prev[r] + v;
    }
  }
}

int launch_gemm_f32(bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, long long sa, const float* B, int ldb,
                    long long sb, float beta, float* C, int ldc, long long sc, int batch, hipStream_t s) {
  if (M <= 0 || N <= 0 || batch <= 0) return EVMI_OK;
  if (batch > 65535 || (M + 63) / 64 > 65535) return fail(EVMI_ERR_UNSUPPORTED, "gemm_f32: grid limits");
  GemmArgs g{A, B, C, M, N, K, lda, ldb, ldc, sa, sb, sc, alpha, beta};
  const dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
  if (ta && tb) hipLaunchKernelGGL((gemm_f32_mfma_kernel<true, true>), grid, dim3(256), 0, s, g);
  else if (ta) hipLaunchKernelGGL((gemm_f32_mfma_kernel<true, false>), grid, dim3(256), 0, s, g);
  else if (tb) hipLaunchKernelGGL((gemm_f32_mfma_kernel<false, true>), grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL((gemm_f32_mfma_kernel<false, false>), grid, dim3(256), 0, s, g);
  EVMI_LAUNCH_CHECK("gemm_f32_mfma");
  return EVMI_OK;
}

}  // namespace evmi
