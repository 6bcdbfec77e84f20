// Building blocks of the GAN training step (HiFiGAN generator + MPD + MSD + losses), fp32.
//
// Layout: every activation is channel-major "CBT" — x[c][b][t], i.e. a [C][B*T] row-major matrix.
// A convolution (any stride / dilation / groups) is then ONE plain GEMM against the unfolded input:
//     fwd     Y[C_out][B*T_out]   = W[C_out][C_in*k] . col[C_in*k][B*T_out]
//     dgrad   dcol[C_in*k][B*T_out] = W^T . dY          -> fold back (col2im)
//     wgrad   dW[C_out][C_in*k]   = dY . col^T          (the batch is part of the GEMM's K dimension)
// and audio [B,1,T] / logits are the same bytes in CBT and in torch's BCT.  The plain GEMMs that remain (the A/B "gemm"
// convolution backend, loss DFTs, the fp32 dense weight-gradient fallback) run on this library's own fp32 matrix-core GEMM
// (gemm_f32.hip); everything around them (unfold / fold, activations, pooling, padding, losses, weight / spectral norm, the
// optimisers) is hand-written here.  No BLAS library is linked.
#include <algorithm>
#include <map>

#include "common.h"

namespace evmi {

int launch_gemm_f32(bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, long long sa, const float* B, int ldb,
                    long long sb, float beta, float* C, int ldc, long long sc, int batch, hipStream_t s);

// out[i] = beta * out[i] + sum_s part[s][i]   (split-K partial sums; out is [M][ldc], part is [S][M][N])
__global__ void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int M, int N, int ldc, int S,
                                     float beta) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)M * N) return;
  const int m = (int)(idx / N), n = (int)(idx % N);
  const float acc = ordered_sum_strided(part + idx, (long long)M * N, S);
  float* o = out + (long long)m * ldc + n;
  *o = beta == 0.f ? acc : beta * *o + acc;
}

// Scratch for partial sums, one buffer per (thread, stream): launches on different streams may overlap on the device.  It only
// grows (hipMalloc outside stream order): warm every shape up before capturing a HIP graph.
struct SplitKWs { float* ptr = nullptr; size_t cap = 0; };
static thread_local std::map<hipStream_t, SplitKWs> g_splitk;
static int splitk_ws(hipStream_t s, size_t need, float** out) {
  SplitKWs& w = g_splitk[s];
  if (need > w.cap) {
    if (w.ptr) (void)hipFree(w.ptr);
    w.ptr = nullptr;
    w.cap = 0;
    EVMI_HIP_CHECK(hipMalloc((void**)&w.ptr, need * sizeof(float)));
    w.cap = need;
  }
  *out = w.ptr;
  return EVMI_OK;
}

// Matrix-vector products of the spectral-norm power iteration (W up to 1024 x 5120): a library GEMM with N = 1 takes 36-51 us
// for them; these read W once at memory speed.  y[r] = sum_c W[r][c] x[c]: one workgroup per row.
__global__ __launch_bounds__(256) void gemv_rows_kernel(const float* __restrict__ W, const float* __restrict__ x, float* __restrict__ y,
                                                        int K, int lda) {
  __shared__ float part[4];
  const float* row = W + (long long)blockIdx.x * lda;
  float acc = 0.f;
  for (int i = threadIdx.x; i < K; i += 256) acc = fmaf(row[i], x[i], acc);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) y[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// y[c] = sum_r W[r][c] u[r]: grid (column blocks of 256, row chunks of GEMV_RC rows) -> partial sums, added in chunk order by
// the second kernel (fixed order: reproducible)
constexpr int GEMV_RC = 32;
__global__ __launch_bounds__(256) void gemv_cols_partial_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                                float* __restrict__ part, int rows, int cols, int lda) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int r0 = blockIdx.y * GEMV_RC, r1 = min(rows, r0 + GEMV_RC);
  float acc = 0.f;
  for (int r = r0; r < r1; ++r) acc = fmaf(W[(long long)r * lda + c], u[r], acc);
  part[(long long)blockIdx.y * cols + c] = acc;
}
__global__ __launch_bounds__(256) void gemv_cols_final_kernel(const float* __restrict__ part, float* __restrict__ y, int chunks, int cols) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  float acc = 0.f;
  for (int i = 0; i < chunks; ++i) acc += part[(long long)i * cols + c];
  y[c] = acc;
}

// C[M][N] = alpha * a b^T + beta * C (the rank-one term of the spectral-norm gradient; a library GEMM with K = 1 took 36 us)
__global__ __launch_bounds__(256) void rank1_update_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b, float* __restrict__ C,
                                                           int N, int ldc, float alpha, float beta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= N) return;
  float* dst = C + (long long)blockIdx.y * ldc + c;
  const float t = alpha * a[(long long)blockIdx.y * lda] * b[c];
  *dst = beta == 0.f ? t : beta * *dst + t;
}

// Row-major C[M][N] = alpha * op(A) . op(B) + beta * C;  op(A) is M x K, op(B) is K x N.
int gemm_rm(bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
            float beta, float* C, int ldc, hipStream_t s) {
  if (N == 1 && alpha == 1.f && beta == 0.f && ldb == 1 && ldc == 1 && (long long)M * K >= 4096) {  // matrix-vector products
    if (!ta) {
      hipLaunchKernelGGL(gemv_rows_kernel, dim3(M), dim3(256), 0, s, A, B, C, K, lda);
      EVMI_LAUNCH_CHECK("gemv_rows");
      return EVMI_OK;
    }
    const int chunks = (K + GEMV_RC - 1) / GEMV_RC;  // op(A) = A^T: A is [K][M]
    float* ws = nullptr;
    if (int rc = splitk_ws(s, (size_t)chunks * M, &ws)) return rc;
    hipLaunchKernelGGL(gemv_cols_partial_kernel, dim3((M + 255) / 256, chunks), dim3(256), 0, s, A, B, ws, K, M, lda);
    hipLaunchKernelGGL(gemv_cols_final_kernel, dim3((M + 255) / 256), dim3(256), 0, s, ws, C, chunks, M);
    EVMI_LAUNCH_CHECK("gemv_cols");
    return EVMI_OK;
  }
  if (K == 1 && !ta && !tb && M <= 65535 && (long long)M * N >= 4096) {  // outer product
    hipLaunchKernelGGL(rank1_update_kernel, dim3((N + 255) / 256, M), dim3(256), 0, s, A, lda, B, C, N, ldc, alpha, beta);
    EVMI_LAUNCH_CHECK("rank1_update");
    return EVMI_OK;
  }
  // Weight-gradient shapes: a small [M][N] output reduced over a very long K (= batch * time).  A single GEMM
  // puts that on a handful of workgroups; split K into equal slabs (a batched GEMM into partial tiles) and add the slabs in order.
  if (K >= 8192 && (long long)M * N <= (1ll << 21)) {
    int S = K / 4096;
    if (S > 128) S = 128;
    const int kc = K / S, tail = K - kc * S;
    const int slabs = S + (tail ? 1 : 0);
    float* ws = nullptr;
    if (int rc = splitk_ws(s, (size_t)slabs * M * N, &ws)) return rc;
    const long long sa = ta ? (long long)kc * lda : kc;  // advance of op(A) along K
    const long long sb = tb ? kc : (long long)kc * ldb;
    if (int rc = launch_gemm_f32(ta, tb, M, N, kc, alpha, A, lda, sa, B, ldb, sb, 0.f, ws, N, (long long)M * N, S, s)) return rc;
    if (tail)
      if (int rc = launch_gemm_f32(ta, tb, M, N, tail, alpha, A + sa * S, lda, 0, B + sb * S, ldb, 0, 0.f, ws + (size_t)S * M * N, N, 0, 1, s)) return rc;
    const long long n = (long long)M * N;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, ws, C, M, N, ldc, slabs, beta);
    EVMI_LAUNCH_CHECK("splitk_reduce");
    return EVMI_OK;
  }
  return launch_gemm_f32(ta, tb, M, N, K, alpha, A, lda, 0, B, ldb, 0, beta, C, ldc, 0, 1, s);
}

// Same, `batch` independent problems at fixed element strides (grouped convolutions: one problem per group).
int gemm_rm_batched(bool ta, bool tb, int M, int N, int K, float alpha, const float* A, int lda, long long sa,
                    const float* B, int ldb, long long sb, float beta, float* C, int ldc, long long sc, int batch,
                    hipStream_t s) {
  if (batch == 1 || K >= 8192) {  // long-K problems take the split-K path one by one
    for (int g = 0; g < batch; ++g) {
      int rc = gemm_rm(ta, tb, M, N, K, alpha, A + g * sa, lda, B + g * sb, ldb, beta, C + g * sc, ldc, s);
      if (rc) return rc;
    }
    return EVMI_OK;
  }
  return launch_gemm_f32(ta, tb, M, N, K, alpha, A, lda, sa, B, ldb, sb, beta, C, ldc, sc, batch, s);
}

// ---- unfold / fold ------------------------------------------------------------------------------------
// col[(c*k + j)][b][to] = x[c][b][to*stride + j*dil - pad]  (0 outside)
// grid (T_out tiles, B, C*k): the row decomposition is per workgroup (scalar), threads walk contiguous `to`
__global__ __launch_bounds__(256) void unfold_cbt_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int t_in,
                                                         int t_out, int k, int stride, int pad, int dil) {
  const int row = blockIdx.z, b = blockIdx.y;
  const int c = row / k, j = row - c * k;
  const float* xr = x + ((long long)c * B + b) * t_in;
  float* cr = col + ((long long)row * B + b) * t_out;
  const int off = j * dil - pad;
  for (int to = blockIdx.x * 1024 + threadIdx.x; to < min(t_out, (int)(blockIdx.x + 1) * 1024); to += 256) {
    const int ti = to * stride + off;
    cr[to] = (ti >= 0 && ti < t_in) ? xr[ti] : 0.f;
  }
}

// dx[c][b][ti] = sum_j dcol[(c*k + j)][b][(ti + pad - j*dil) / stride]   (terms that divide evenly and are in range)
__global__ __launch_bounds__(256) void fold_cbt_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int t_in,
                                                       int t_out, int k, int stride, int pad, int dil, int accumulate) {
  const int c = blockIdx.z, b = blockIdx.y;
  float* xr = dx + ((long long)c * B + b) * t_in;
  const float* cbase = dcol + ((long long)c * k * B + b) * t_out;
  const long long jstride = (long long)B * t_out;
  for (int ti = blockIdx.x * 1024 + threadIdx.x; ti < min(t_in, (int)(blockIdx.x + 1) * 1024); ti += 256) {
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
      const int num = ti + pad - j * dil;
      if (num < 0) break;  // num only decreases with j
      if (stride > 1 && num % stride) continue;
      const int to = stride > 1 ? num / stride : num;
      if (to < t_out) acc += cbase[j * jstride + to];
    }
    xr[ti] = accumulate ? xr[ti] + acc : acc;
  }
}

// flat variants (one element per thread) for short rows, where a per-row grid would be mostly empty workgroups
__global__ void unfold_cbt_flat_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int t_in, int t_out, int k,
                                       int stride, int pad, int dil, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int to = (int)(idx % t_out);
  long long r = idx / t_out;
  const int b = (int)(r % B);
  r /= B;
  const int j = (int)(r % k);
  const long long c = r / k;
  const int ti = to * stride + j * dil - pad;
  col[idx] = (ti >= 0 && ti < t_in) ? x[(c * B + b) * t_in + ti] : 0.f;
}
__global__ void fold_cbt_flat_kernel(const float* __restrict__ dcol, float* __restrict__ dx, int B, int t_in, int t_out, int k,
                                     int stride, int pad, int dil, long long n, int accumulate) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int ti = (int)(idx % t_in);
  long long r = idx / t_in;
  const int b = (int)(r % B);
  const long long c = r / B;
  float acc = 0.f;
  for (int j = 0; j < k; ++j) {
    const int num = ti + pad - j * dil;
    if (num < 0) break;
    if (num % stride) continue;
    const int to = num / stride;
    if (to < t_out) acc += dcol[((c * k + j) * B + b) * t_out + to];
  }
  dx[idx] = accumulate ? dx[idx] + acc : acc;
}

// Weights of the convolution that computes an input gradient.  For phase `phi` of a stride-s convolution
// (stride 1: the single phase 0) with taps j = phi + s*m, m < M:
//   wt[g*cin_g + ci][co_l][m'] = w[g*cout_g + co_l][ci][phi + s*(M - 1 - m')]
// so that dx_phi = conv1d(dy, wt, groups) (see everyvoice_amd/train/ops.py: conv1d_bwd_data_mfma).
__global__ void dgrad_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int cin_g, int cout_g, int k,
                                     int stride, int phi, int M, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int mp = (int)(idx % M);
  long long r = idx / M;
  const int co_l = (int)(r % cout_g);
  const long long ci_all = r / cout_g;  // g*cin_g + ci
  const int g = (int)(ci_all / cin_g), ci = (int)(ci_all % cin_g);
  const int j = phi + stride * (M - 1 - mp);
  wt[idx] = w[((long long)(g * cout_g + co_l) * cin_g + ci) * k + j];
}

// ---- row-wise helpers on [R][N] matrices ---------------------------------------------------------------
__global__ void bias_add_rows_kernel(float* __restrict__ y, const float* __restrict__ bias, long long N, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < n) y[idx] += bias[idx / N];
}

// out[r] (+)= scale * sum_n f(a[r][n], b[r][n]);  MODE 0: a ; 1: a*b ; 2: a*a
// grid (rows, nseg): workgroup (r, sg) sums its segment of the row; nseg == 1 writes out[r], otherwise part[r][sg]
// and row_reduce_final_kernel adds the segments in a fixed order (reproducible).
template <int MODE>
__global__ __launch_bounds__(256) void row_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, float* __restrict__ part_out, long long N,
                                                         long long seg, float scale, int accumulate) {
  __shared__ float part[4];
  const long long r = blockIdx.x;
  const long long lo = (long long)blockIdx.y * seg, hi = min(N, lo + seg);
  const float* ar = a + r * N;
  const float* br = b ? b + r * N : nullptr;
  float acc = 0.f;
  long long i = lo + threadIdx.x;
  for (; i + 3 * 256 < hi; i += 4 * 256) {  // four independent reads per trip (a one-read loop drains vmcnt behind every element)
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = ar[i + u * 256];
      bv[u] = MODE == 1 ? br[i + u * 256] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) acc += MODE == 0 ? av[u] : (MODE == 1 ? av[u] * bv[u] : av[u] * av[u]);
  }
  for (; i < hi; i += 256) {
    const float av = ar[i];
    acc += MODE == 0 ? av : (MODE == 1 ? av * br[i] : av * av);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = part[0] + part[1] + part[2] + part[3];
    if (gridDim.y == 1) out[r] = accumulate ? out[r] + t * scale : t * scale;
    else part_out[r * gridDim.y + blockIdx.y] = t;
  }
}

// Backward of leaky_relu(conv(x)) in one pass over dy: dpre = dy * (y > 0 ? 1 : slope) is written for the convolution's gradient
// kernels and its row sums (the bias gradient) come out of the same read (same two-stage scheme as row_reduce_kernel).
__global__ __launch_bounds__(256) void lrelu_bwd_rowsum_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                               float* __restrict__ dpre, float* __restrict__ out, float* __restrict__ part_out,
                                                               long long N, long long seg, float slope, int accumulate) {
  __shared__ float part[4];
  const long long r = blockIdx.x;
  const long long lo = (long long)blockIdx.y * seg, hi = min(N, lo + seg);
  float acc = 0.f;
  long long i = lo + threadIdx.x;
  for (; i + 3 * 256 < hi; i += 4 * 256) {
    float dv[4], yv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dv[u] = dy[r * N + i + u * 256];
      yv[u] = y[r * N + i + u * 256];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float v = dv[u] * (yv[u] > 0.f ? 1.f : slope);
      dpre[r * N + i + u * 256] = v;
      acc += v;
    }
  }
  for (; i < hi; i += 256) {
    const float v = dy[r * N + i] * (y[r * N + i] > 0.f ? 1.f : slope);
    dpre[r * N + i] = v;
    acc += v;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float t = part[0] + part[1] + part[2] + part[3];
    if (gridDim.y == 1) out[r] = accumulate ? out[r] + t : t;
    else part_out[r * gridDim.y + blockIdx.y] = t;
  }
}

__global__ void row_reduce_final_kernel(const float* __restrict__ part, float* __restrict__ out, int rows, int nseg,
                                        float scale, int accumulate) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const float t = ordered_sum_strided(part + (long long)r * nseg, 1, nseg);
  out[r] = accumulate ? out[r] + t * scale : t * scale;
}

// ---- elementwise -----------------------------------------------------------------------------------------
// OP 0: y = lrelu(x, p0)                     1: dx = dy * (x > 0 ? 1 : p0)          (a = dy, b = x)
//    2: y = tanh(x)                          3: dx = dy * (1 - y*y)                 (a = dy, b = y)
//    4: y = p0 * a + p1 * b                  5: y = p0 * a                          6: y = a * b
//    7: y = sign(a - b) * p0                 8: y = 2 * (a - p1) * p0               (d/da of p0 * (a - p1)^2)
//    9: y = log(max(a, p0))                 10: y = b > p0 ? a / b : 0              (dmel from dlogmel, b = mel)
//   11: y = sqrt(a*a + b*b + p0)            12: y = a * b / c  (dre = dmag * re / mag)
//   13: y = silu(a)                         14: y = relu(a)                        15: y = a * sigmoid(b)  (GLU)
//   20: y = a * min(1, p0 / (sqrt(c[0]) + 1e-6))   (gradient-norm clipping, c[0] = sum of squares on the device)
//   18: y = a * silu'(b)                    19: y = b > 0 ? a : 0  (ReLU backward from the output)
//   23: y = p0 (fill)
//   24: y = p0 * a / c[0]  (device scalar: a count that changes from batch to batch without changing a captured graph)
//   21: y = a / c[0]  (device scalar)        22: op 17 with p0 / (sqrt(c[0]) sqrt(c[1])) as the first coefficient (device scalars)
//   16: y = log(a)                          17: y = p0 * (a - b) + p1 * sign(a - b) / a   (d/da of the two STFT-loss terms, a = |Y^|, b = |Y|)
template <int OP>
__global__ void ew_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                          float* __restrict__ y, long long n, float p0, float p1) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float r;
  if (OP == 0) { const float v = a[i]; r = v > 0.f ? v : v * p0; }
  else if (OP == 1) r = a[i] * (b[i] > 0.f ? 1.f : p0);
  else if (OP == 2) r = tanhf(a[i]);
  else if (OP == 3) { const float t = b[i]; r = a[i] * (1.f - t * t); }
  else if (OP == 4) r = p0 * a[i] + p1 * b[i];
  else if (OP == 5) r = p0 * a[i];
  else if (OP == 6) r = a[i] * b[i];
  else if (OP == 7) { const float d = a[i] - b[i]; r = d > 0.f ? p0 : (d < 0.f ? -p0 : 0.f); }
  else if (OP == 8) r = 2.f * (a[i] - p1) * p0;
  else if (OP == 9) r = logf(fmaxf(a[i], p0));
  else if (OP == 10) r = b[i] > p0 ? a[i] / b[i] : 0.f;
  else if (OP == 11) r = sqrtf(a[i] * a[i] + b[i] * b[i] + p0);
  else if (OP == 12) r = a[i] * b[i] / c[i];
  else if (OP == 13) { const float v = a[i]; r = v / (1.f + expf(-v)); }
  else if (OP == 14) r = fmaxf(a[i], 0.f);
  else if (OP == 15) r = a[i] / (1.f + expf(-b[i]));
  else if (OP == 16) r = logf(a[i]);
  else if (OP == 18) { const float z = b[i], sg = 1.f / (1.f + expf(-z)); r = a[i] * sg * (1.f + z * (1.f - sg)); }
  else if (OP == 19) r = b[i] > 0.f ? a[i] : 0.f;
  else if (OP == 20) r = a[i] * fminf(1.f, p0 / (sqrtf(c[0]) + 1e-6f));
  else if (OP == 21) r = a[i] / c[0];
  else if (OP == 23) r = p0;
  else if (OP == 24) r = p0 * a[i] / c[0];
  else if (OP == 22) {  // op 17 with its first coefficient p0 / (||a - b|| ||b||) formed from the squared norms c[0], c[1] on the device
    const float nd = sqrtf(c[0]), ny = sqrtf(c[1]);
    const float k0 = nd > 0.f ? p0 / (nd * ny) : 0.f;
    r = k0 * (a[i] - b[i]) + p1 * (a[i] > b[i] ? 1.f : (a[i] < b[i] ? -1.f : 0.f)) / a[i];
  }
  else r = p0 * (a[i] - b[i]) + p1 * (a[i] > b[i] ? 1.f : (a[i] < b[i] ? -1.f : 0.f)) / a[i];
  y[i] = r;
}

// deterministic scalar reductions: out[0] (+)= scale * sum f;  MODE 0: |a-b| ; 1: (a-p)^2 ; 2: a
// two fixed-shape passes (grid of partial sums in double, then one workgroup): bitwise reproducible
template <int MODE>
__global__ __launch_bounds__(256) void scalar_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             double* __restrict__ part, long long n, float p) {
  __shared__ double sh[4];
  double acc = 0.0;
  const long long step = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + 3 * step < n; i += 4 * step) {
    float av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      av[u] = a[i + u * step];
      bv[u] = MODE == 0 ? b[i + u * step] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      acc += MODE == 0 ? (double)fabsf(av[u] - bv[u]) : (MODE == 1 ? (double)((av[u] - p) * (av[u] - p)) : (double)av[u]);
  }
  for (; i < n; i += step) {
    const float av = a[i];
    acc += MODE == 0 ? (double)fabsf(av - b[i]) : (MODE == 1 ? (double)((av - p) * (av - p)) : (double)av);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void scalar_final_kernel(const double* __restrict__ part, int n_part, float* __restrict__ out,
                                                           float scale, int accumulate) {
  __shared__ double sh[4];
  double acc = 0.0;
  for (int i = threadIdx.x; i < n_part; i += 256) acc += part[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const float r = (float)((sh[0] + sh[1] + sh[2] + sh[3]) * scale);
    out[0] = accumulate ? out[0] + r : r;
  }
}

// ---- pooling / padding / period view ------------------------------------------------------------------------
// AvgPool1d(4, 2, padding=2), count_include_pad: y[r][to] = (sum_{j<4} x[r][2 to + j - 2]) / 4
__global__ void avgpool4s2_kernel(const float* __restrict__ x, float* __restrict__ y, int t_in, int t_out, long long n, int bwd) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  if (!bwd) {
    const int to = (int)(idx % t_out);
    const long long r = idx / t_out;
    float acc = 0.f;
    for (int j = 0; j < 4; ++j) {
      const int ti = 2 * to + j - 2;
      if (ti >= 0 && ti < t_in) acc += x[r * t_in + ti];
    }
    y[idx] = 0.25f * acc;
  } else {  // x = dy [rows][t_out], y = dx [rows][t_in]
    const int ti = (int)(idx % t_in);
    const long long r = idx / t_in;
    float acc = 0.f;
    for (int j = 0; j < 4; ++j) {
      const int num = ti + 2 - j;
      if (num < 0 || (num & 1)) continue;
      const int to = num >> 1;
      if (to < t_out) acc += x[r * t_out + to];
    }
    y[idx] = 0.25f * acc;
  }
}

// MPD view: audio x[b][t] (t < T), reflect-padded on the right to T' = H * p, then x2[(b*p + w)][h] = xpad[b][h*p + w]
__global__ void period_view_kernel(const float* __restrict__ x, float* __restrict__ x2, int T, int H, int p, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int h = (int)(idx % H);
  const long long r = idx / H;
  const int w = (int)(r % p);
  const long long b = r / p;
  int t = h * p + w;
  if (t >= T) t = 2 * (T - 1) - t;  // reflect
  x2[idx] = x[b * T + t];
}
// adjoint: dx[b][t] = sum over the padded positions that map to t
__global__ void period_view_bwd_kernel(const float* __restrict__ dx2, float* __restrict__ dx, int T, int H, int p, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int t = (int)(idx % T);
  const long long b = idx / T;
  float acc = dx2[(b * p + t % p) * H + t / p];
  const int tr = 2 * (T - 1) - t;  // the padded position whose reflection is t
  if (tr >= T && tr < H * p) acc += dx2[(b * p + tr % p) * H + tr / p];
  dx[idx] = acc;
}

// ---- STFT framing for the mel loss ------------------------------------------------------------------------------
// frames[k][b][f] = xpad[b][f*hop + k - n_fft/2] (reflect)   -> [n_fft][B*F]; the window lives in the DFT basis
__global__ void frame_kernel(const float* __restrict__ x, float* __restrict__ fr, int T, int F, int n_fft, int hop, long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int f = (int)(idx % F);
  long long r = idx / F;
  const long long nb = n / ((long long)n_fft * F);
  const int b = (int)(r % nb);
  const int k = (int)(r / nb);
  int t = f * hop + k - n_fft / 2;
  if (t < 0) t = -t;
  if (t >= T) t = 2 * (T - 1) - t;
  fr[idx] = x[(long long)b * T + t];
}
// adjoint of the framing: dx[b][t] = sum over (k, f) whose (reflected) source position is t.  One thread per
// (b, t): positions u in the padded signal that read t are u = t, and the mirror images -t and 2(T-1)-t
__global__ void frame_bwd_kernel(const float* __restrict__ dfr, float* __restrict__ dx, int B, int T, int F, int n_fft, int hop,
                                 long long n) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const int t = (int)(idx % T);
  const int b = (int)(idx / T);
  const int pad = n_fft / 2;
  float acc = 0.f;
  const int cand[3] = {t, -t, 2 * (T - 1) - t};
  for (int ci = 0; ci < 3; ++ci) {
    const int u = cand[ci];  // un-reflected position (may lie in the padding)
    if (ci == 1 && (t == 0 || u < -pad)) continue;
    if (ci == 2 && (t == T - 1 || u > T - 1 + pad)) continue;
    // frames f with 0 <= u + pad - f*hop < n_fft
    const int s = u + pad;
    int f_hi = s / hop;
    if (f_hi > F - 1) f_hi = F - 1;
    int f_lo = (s - n_fft + hop) / hop;  // ceil((s - n_fft + 1) / hop) for s - n_fft + 1 possibly negative
    if (s - n_fft + 1 <= 0) f_lo = 0;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int k = s - f * hop;
      if (k >= 0 && k < n_fft) acc += dfr[((long long)k * B + b) * F + f];
    }
  }
  dx[idx] = acc;
}

// ---- weight norm / spectral norm / optimiser ------------------------------------------------------------------------
// w[r][:] = g[r] * v[r][:] / ||v[r]||        (torch.nn.utils.weight_norm, dim=0)
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                              float* __restrict__ w, float* __restrict__ norm, int N) {
  __shared__ float part[4];
  __shared__ float nrm;
  const long long r = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) { const float t = v[r * N + i]; acc += t * t; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { nrm = sqrtf(part[0] + part[1] + part[2] + part[3]); norm[r] = nrm; }
  __syncthreads();
  const float sc = g[r] / nrm;
  for (int i = threadIdx.x; i < N; i += 256) w[r * N + i] = v[r * N + i] * sc;
}
// dg[r] = <dw, v> / ||v|| ;  dv = g/||v|| * (dw - v * <dw, v> / ||v||^2)
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ v,
                                                              const float* __restrict__ norm, const float* __restrict__ dw,
                                                              float* __restrict__ dg, float* __restrict__ dv, int N) {
  __shared__ float part[4];
  __shared__ float dot;
  const long long r = blockIdx.x;
  float acc = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) acc += dw[r * N + i] * v[r * N + i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { dot = part[0] + part[1] + part[2] + part[3]; dg[r] = dot / norm[r]; }
  __syncthreads();
  const float nr = norm[r], sc = g[r] / nr, k = dot / (nr * nr);
  for (int i = threadIdx.x; i < N; i += 256) dv[r * N + i] = sc * (dw[r * N + i] - v[r * N + i] * k);
}

// Weight norm of MANY layers in one launch (one optimiser's weight-normed convolutions; the per-layer launches were 5-7 us each,
// ~300 per GAN step).  table [6][L + 1] int64: row_start (prefix sums of the layers' row counts), n_per_row, and the offsets of
// g / v in the flat parameter buffer, of w in the effective-weight buffer (same for its gradient sink) and of the norms.
// One workgroup per row; rows [row_lo, row_hi) of the concatenated row list.
__device__ __forceinline__ int wn_find_layer(const long long* __restrict__ row_start, int L, long long r) {
  int lo = 0, hi = L - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (row_start[mid] <= r) lo = mid; else hi = mid - 1;
  }
  return lo;
}
__global__ __launch_bounds__(256) void weight_norm_fwd_batched_kernel(const float* __restrict__ flat, float* __restrict__ eff,
                                                                      float* __restrict__ norms, const long long* __restrict__ tab, int L,
                                                                      long long row_lo) {
  __shared__ float part[4];
  __shared__ float nrm;
  const long long r = row_lo + blockIdx.x;
  const int l = wn_find_layer(tab, L, r);
  const long long lr = r - tab[l];
  const int N = (int)tab[(L + 1) + l];
  const float* v = flat + tab[3 * (L + 1) + l] + lr * N;
  float* w = eff + tab[4 * (L + 1) + l] + lr * N;
  float acc = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) { const float t = v[i]; acc += t * t; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) { nrm = sqrtf(part[0] + part[1] + part[2] + part[3]); norms[tab[5 * (L + 1) + l] + lr] = nrm; }
  __syncthreads();
  const float sc = flat[tab[2 * (L + 1) + l] + lr] / nrm;
  for (int i = threadIdx.x; i < N; i += 256) w[i] = v[i] * sc;
}
// dg, dv from the effective-weight gradient sink (then zeroed for the next step), same arithmetic as weight_norm_bwd_kernel
__global__ __launch_bounds__(256) void weight_norm_bwd_batched_kernel(const float* __restrict__ flat, float* __restrict__ grad,
                                                                      const float* __restrict__ norms, float* __restrict__ dw_eff,
                                                                      const long long* __restrict__ tab, int L, long long row_lo) {
  __shared__ float part[4];
  __shared__ float dot;
  const long long r = row_lo + blockIdx.x;
  const int l = wn_find_layer(tab, L, r);
  const long long lr = r - tab[l];
  const int N = (int)tab[(L + 1) + l];
  const long long voff = tab[3 * (L + 1) + l] + lr * N, goff = tab[2 * (L + 1) + l] + lr;
  const float* v = flat + voff;
  float* dw = dw_eff + tab[4 * (L + 1) + l] + lr * N;
  float acc = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) acc += dw[i] * v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  const float nr = norms[tab[5 * (L + 1) + l] + lr];
  if (threadIdx.x == 0) { dot = part[0] + part[1] + part[2] + part[3]; grad[goff] = dot / nr; }
  __syncthreads();
  const float sc = flat[goff] / nr, k = dot / (nr * nr);
  float* dv = grad + voff;
  for (int i = threadIdx.x; i < N; i += 256) {
    dv[i] = sc * (dw[i] - v[i] * k);
    dw[i] = 0.f;
  }
}

// iSTFTNet head, between conv_post and the inverse STFT (the generator of `istft_layer: true`): a [2H][n] = H log-magnitude rows
// then H phase rows  ->  s [2H][n] = H real rows then H imaginary rows of exp(a) * exp(i * sin(b)).
__global__ __launch_bounds__(256) void istft_polar_kernel(const float* __restrict__ a, float* __restrict__ s, int H, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)H * n) return;
  const float mag = expf(a[i]), ph = sinf(a[(long long)H * n + i]);
  s[i] = mag * cosf(ph);
  s[(long long)H * n + i] = mag * sinf(ph);
}
// da from ds (same layouts): d log-mag = (dre cos + dim sin) mag;  d phase-pre = mag (-dre sin + dim cos) cos(b)
__global__ __launch_bounds__(256) void istft_polar_bwd_kernel(const float* __restrict__ a, const float* __restrict__ ds, float* __restrict__ da,
                                                              int H, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)H * n) return;
  const long long j = (long long)H * n + i;
  const float b = a[j], mag = expf(a[i]), ph = sinf(b), c = cosf(ph), sn = sinf(ph);
  const float dre = ds[i], dim = ds[j];
  da[i] = (dre * c + dim * sn) * mag;
  da[j] = mag * (dim * c - dre * sn) * cosf(b);
}
// ReflectionPad1d((1, 0)) on rows [rows][T] -> [rows][T + 1]: y[0] = x[1], y[1 + t] = x[t]; bwd = 1: the adjoint (x, y swap roles:
// `x` receives dx[t] = dy[t + 1] + (t == 1 ? dy[0] : 0) from `y` = dy)
__global__ __launch_bounds__(256) void reflect_pad_left1_kernel(const float* __restrict__ src, float* __restrict__ dst, long long rows, int T,
                                                                int bwd) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (!bwd) {
    if (i >= rows * (T + 1)) return;
    const long long r = i / (T + 1);
    const int t = (int)(i - r * (T + 1));
    dst[i] = src[r * T + (t == 0 ? 1 : t - 1)];
  } else {
    if (i >= rows * T) return;
    const long long r = i / T;
    const int t = (int)(i - r * T);
    const float* dy = src + r * (T + 1);
    dst[i] = dy[t + 1] + (t == 1 ? dy[0] : 0.f);
  }
}

// y = x / max(||x||, eps)  (one workgroup; spectral norm's power iteration vectors are <= a few thousand long)
__global__ __launch_bounds__(1024) void normalize_vec_kernel(const float* __restrict__ x, float* __restrict__ y, int n, float eps) {
  __shared__ float part[16];
  __shared__ float nrm;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) acc += x[i] * x[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += part[i];
    nrm = fmaxf(sqrtf(t), eps);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += 1024) y[i] = x[i] / nrm;
}

// AdamW (torch.optim.AdamW, amsgrad off): decoupled weight decay, bias-corrected moments
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float pv = p[i];
  const float gv = g[i];
  pv *= 1.f - lr * wd;
  const float mv = beta1 * m[i] + (1.f - beta1) * gv;
  const float vv = beta2 * v[i] + (1.f - beta2) * gv * gv;
  m[i] = mv;
  v[i] = vv;
  const float denom = sqrtf(vv) / sqrtf(bc2) + eps;
  p[i] = pv - (lr / bc1) * mv / denom;
}

// One step of Adam / AdamW / RMSprop on a flat buffer (torch.optim semantics, amsgrad / momentum / centered off), the optimiser
// union the reference's training config allows (everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:434-622).
//   kind 0 AdamW: decoupled decay p *= 1 - lr wd;  kind 1 Adam: g += wd p (L2);  kind 2 RMSprop: v = a v + (1-a) g^2, p -= lr g / (sqrt(v) + eps)
// The 1-based step number comes from a device counter when `step_dev` is given (so a captured HIP graph replays correctly);
// clip > 0 clamps the updated parameters to [-clip, clip] (WGAN weight clipping, wgan_clip_value).
__device__ __forceinline__ float optimizer_update(int kind, float pv, float gv, float& mv, float& vv, float lr, float beta1, float beta2,
                                                  float eps, float wd, float bc1, float rsq_bc2, float clip) {
  if (kind == 2) {
    gv = fmaf(wd, pv, gv);
    vv = beta1 * vv + (1.f - beta1) * gv * gv;  // beta1 carries RMSprop's alpha
    pv -= lr * gv / (sqrtf(vv) + eps);
  } else {
    if (kind == 0) pv *= 1.f - lr * wd;
    else gv = fmaf(wd, pv, gv);
    mv = beta1 * mv + (1.f - beta1) * gv;
    vv = beta2 * vv + (1.f - beta2) * gv * gv;
    pv -= (lr / bc1) * mv / (sqrtf(vv) * rsq_bc2 + eps);
  }
  if (clip > 0.f) pv = fminf(fmaxf(pv, -clip), clip);
  return pv;
}
// four parameters per thread (16-byte accesses: the update is pure HBM traffic, 7 streams); n4 = n / 4, tail elements scalar
__global__ void optimizer_step_kernel(int kind, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                      float* __restrict__ v, long long n, float lr, float beta1, float beta2, float eps, float wd,
                                      int step, const int* __restrict__ step_dev, float clip, const float* __restrict__ lr_dev) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long n4 = n >> 2;
  if (lr_dev) lr = *lr_dev;  // a scheduled learning rate the host stored on the device (a captured step replays with the new value)
  const float st = (float)(step_dev ? *step_dev : step);
  const float bc1 = kind == 2 ? 1.f : 1.f - powf(beta1, st), rsq_bc2 = kind == 2 ? 1.f : 1.f / sqrtf(1.f - powf(beta2, st));
  if (i < n4) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = kind == 2 ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    pv.x = optimizer_update(kind, pv.x, gv.x, mv.x, vv.x, lr, beta1, beta2, eps, wd, bc1, rsq_bc2, clip);
    pv.y = optimizer_update(kind, pv.y, gv.y, mv.y, vv.y, lr, beta1, beta2, eps, wd, bc1, rsq_bc2, clip);
    pv.z = optimizer_update(kind, pv.z, gv.z, mv.z, vv.z, lr, beta1, beta2, eps, wd, bc1, rsq_bc2, clip);
    pv.w = optimizer_update(kind, pv.w, gv.w, mv.w, vv.w, lr, beta1, beta2, eps, wd, bc1, rsq_bc2, clip);
    reinterpret_cast<float4*>(p)[i] = pv;
    if (kind != 2) reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  } else if (i < n4 + (n & 3)) {
    const long long e = n4 * 4 + (i - n4);
    float mv = kind == 2 ? 0.f : m[e], vv = v[e];
    p[e] = optimizer_update(kind, p[e], g[e], mv, vv, lr, beta1, beta2, eps, wd, bc1, rsq_bc2, clip);
    if (kind != 2) m[e] = mv;
    v[e] = vv;
  }
}
__global__ void counter_add_kernel(int* c, int delta) { *c += delta; }
// out[c][b][t] = in[b][c][t]  (torch's [B, C, T] batch to the channel-major training layout; rows of T stay contiguous)
__global__ void transpose_bct_cbt_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int T, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int t = (int)(i % T);
  const long long r = i / T;
  const int b = (int)(r % B), c = (int)(r / B);
  out[i] = in[((long long)b * C + c) * T + t];
}

// Spectral-norm pieces with the scale sigma left on the device (no host round trip in the step):
//   sn_grad: gW[r][c] += dw[r][c] / sigma - (dot / sigma^2) * u[r] * v[c]      (dot = <dw, W>, sigma = u^T W v)
__global__ void sn_grad_kernel(float* __restrict__ gW, const float* __restrict__ dw, const float* __restrict__ u,
                               const float* __restrict__ v, const float* __restrict__ sigma, const float* __restrict__ dot, int cols,
                               long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float sg = *sigma;
  const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
  gW[i] += dw[i] / sg - (*dot / (sg * sg)) * u[r] * v[c];
}
// out[0] += w * sqrt(c[0] / c[1])   (the spectral-convergence term of the multi-resolution STFT loss from its two squared norms)
__global__ void ratio_acc_kernel(float* out, const float* c, float w) { out[0] += c[1] > 0.f ? w * sqrtf(c[0] / c[1]) : 0.f; }

static inline dim3 grid1d(long long n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace evmi

using namespace evmi;

#define EVMI_NONNULL(p, what) \
  if (!(p)) return fail(EVMI_ERR_INVALID_ARG, what ": null pointer")

extern "C" {

int evmi_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* a_dev, int lda,
                  const float* b_dev, int ldb, float beta, float* c_dev, int ldc, void* stream) {
  EVMI_NONNULL(a_dev && b_dev && c_dev, "gemm_f32");
  if (M <= 0 || N <= 0 || K <= 0) return fail(EVMI_ERR_INVALID_ARG, "gemm_f32: empty problem");
  return gemm_rm(trans_a != 0, trans_b != 0, M, N, K, alpha, a_dev, lda, b_dev, ldb, beta, c_dev, ldc, (hipStream_t)stream);
}

int evmi_gemm_batched_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* a_dev, int lda,
                          long long stride_a, const float* b_dev, int ldb, long long stride_b, float beta, float* c_dev,
                          int ldc, long long stride_c, int batch, void* stream) {
  EVMI_NONNULL(a_dev && b_dev && c_dev, "gemm_batched_f32");
  if (M <= 0 || N <= 0 || K <= 0 || batch <= 0) return fail(EVMI_ERR_INVALID_ARG, "gemm_batched_f32: empty problem");
  return gemm_rm_batched(trans_a != 0, trans_b != 0, M, N, K, alpha, a_dev, lda, stride_a, b_dev, ldb, stride_b, beta, c_dev, ldc,
                         stride_c, batch, (hipStream_t)stream);
}

int evmi_dgrad_weights_f32(const float* w_dev, float* wt_dev, int c_in, int c_out, int k, int groups, int stride, int phi,
                           void* stream) {
  EVMI_NONNULL(w_dev && wt_dev, "dgrad_weights");
  if (phi < 0 || phi >= stride || phi >= k) return fail(EVMI_ERR_INVALID_ARG, "dgrad_weights: phase");
  const int M = (k - phi + stride - 1) / stride;
  const long long n = (long long)c_in * (c_out / groups) * M;
  hipLaunchKernelGGL(dgrad_weights_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, w_dev, wt_dev, c_in / groups, c_out / groups, k,
                     stride, phi, M, n);
  EVMI_LAUNCH_CHECK("dgrad_weights");
  return EVMI_OK;
}

int evmi_unfold_cbt_f32(const float* x_dev, float* col_dev, int C, int B, int t_in, int t_out, int k, int stride, int pad,
                        int dil, void* stream) {
  EVMI_NONNULL(x_dev && col_dev, "unfold_cbt");
  if (t_out >= 512 && (long long)C * k <= 65535 && B <= 65535) {
    hipLaunchKernelGGL(unfold_cbt_kernel, dim3((t_out + 1023) / 1024, B, C * k), dim3(256), 0, (hipStream_t)stream, x_dev, col_dev,
                       B, t_in, t_out, k, stride, pad, dil);
  } else {
    const long long n = (long long)C * k * B * t_out;
    hipLaunchKernelGGL(unfold_cbt_flat_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x_dev, col_dev, B, t_in, t_out, k,
                       stride, pad, dil, n);
  }
  EVMI_LAUNCH_CHECK("unfold_cbt");
  return EVMI_OK;
}

int evmi_fold_cbt_f32(const float* dcol_dev, float* dx_dev, int C, int B, int t_in, int t_out, int k, int stride, int pad,
                      int dil, int accumulate, void* stream) {
  EVMI_NONNULL(dcol_dev && dx_dev, "fold_cbt");
  if (t_in >= 512 && C <= 65535 && B <= 65535) {
    hipLaunchKernelGGL(fold_cbt_kernel, dim3((t_in + 1023) / 1024, B, C), dim3(256), 0, (hipStream_t)stream, dcol_dev, dx_dev, B,
                       t_in, t_out, k, stride, pad, dil, accumulate);
  } else {
    const long long n = (long long)C * B * t_in;
    hipLaunchKernelGGL(fold_cbt_flat_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, dcol_dev, dx_dev, B, t_in, t_out, k,
                       stride, pad, dil, n, accumulate);
  }
  EVMI_LAUNCH_CHECK("fold_cbt");
  return EVMI_OK;
}

int evmi_bias_add_rows_f32(float* y_dev, const float* bias_dev, int rows, long long n_per_row, void* stream) {
  EVMI_NONNULL(y_dev && bias_dev, "bias_add_rows");
  const long long n = (long long)rows * n_per_row;
  hipLaunchKernelGGL(bias_add_rows_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, y_dev, bias_dev, n_per_row, n);
  EVMI_LAUNCH_CHECK("bias_add_rows");
  return EVMI_OK;
}

/* out[r] (+)= scale * sum_n f;  mode 0: a, 1: a*b, 2: a*a */
int evmi_row_reduce_f32(int mode, const float* a_dev, const float* b_dev, float* out_dev, int rows, long long n_per_row,
                        float scale, int accumulate, void* stream) {
  EVMI_NONNULL(a_dev && out_dev, "row_reduce");
  hipStream_t s = (hipStream_t)stream;
  // few long rows: split every row into segments so the chip is busy (two passes, fixed summation order)
  constexpr int MAXSEG = 64;
  float* part = nullptr;
  int nseg = 1;
  if (rows < 512 && n_per_row > 16384) nseg = (int)std::min<long long>(MAXSEG, std::min<long long>((1024 + rows - 1) / rows, (n_per_row + 8191) / 8192));
  const long long seg = (n_per_row + nseg - 1) / nseg;
  if (nseg > 1)
    if (int rc = splitk_ws(s, (size_t)rows * MAXSEG, &part)) return rc;
  const dim3 grid(rows, nseg);
  if (mode == 0) hipLaunchKernelGGL(row_reduce_kernel<0>, grid, dim3(256), 0, s, a_dev, b_dev, out_dev, part, n_per_row, seg, scale, accumulate);
  else if (mode == 1) hipLaunchKernelGGL(row_reduce_kernel<1>, grid, dim3(256), 0, s, a_dev, b_dev, out_dev, part, n_per_row, seg, scale, accumulate);
  else if (mode == 2) hipLaunchKernelGGL(row_reduce_kernel<2>, grid, dim3(256), 0, s, a_dev, b_dev, out_dev, part, n_per_row, seg, scale, accumulate);
  else return fail(EVMI_ERR_INVALID_ARG, "row_reduce: mode");
  if (nseg > 1) hipLaunchKernelGGL(row_reduce_final_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, out_dev, rows, nseg, scale, accumulate);
  EVMI_LAUNCH_CHECK("row_reduce");
  return EVMI_OK;
}

int evmi_lrelu_bwd_rowsum_f32(const float* dy_dev, const float* y_dev, float* dpre_dev, float* db_dev, int rows, long long n_per_row,
                              float slope, int accumulate, void* stream) {
  EVMI_NONNULL(dy_dev && y_dev && dpre_dev && db_dev, "lrelu_bwd_rowsum");
  hipStream_t s = (hipStream_t)stream;
  constexpr int MAXSEG = 64;
  float* part = nullptr;
  int nseg = 1;
  if (rows < 512 && n_per_row > 16384) nseg = (int)std::min<long long>(MAXSEG, std::min<long long>((1024 + rows - 1) / rows, (n_per_row + 8191) / 8192));
  const long long seg = (n_per_row + nseg - 1) / nseg;
  if (nseg > 1)
    if (int rc = splitk_ws(s, (size_t)rows * MAXSEG, &part)) return rc;
  hipLaunchKernelGGL(lrelu_bwd_rowsum_kernel, dim3(rows, nseg), dim3(256), 0, s, dy_dev, y_dev, dpre_dev, db_dev, part, n_per_row, seg, slope,
                     accumulate);
  if (nseg > 1) hipLaunchKernelGGL(row_reduce_final_kernel, dim3((rows + 255) / 256), dim3(256), 0, s, part, db_dev, rows, nseg, 1.f, accumulate);
  EVMI_LAUNCH_CHECK("lrelu_bwd_rowsum");
  return EVMI_OK;
}

int evmi_elementwise_f32(int op, const float* a_dev, const float* b_dev, const float* c_dev, float* y_dev, long long n,
                         float p0, float p1, void* stream) {
  EVMI_NONNULL(a_dev && y_dev, "elementwise");
  hipStream_t s = (hipStream_t)stream;
#define EW(OPN) case OPN: hipLaunchKernelGGL(ew_kernel<OPN>, grid1d(n), dim3(256), 0, s, a_dev, b_dev, c_dev, y_dev, n, p0, p1); break;
  switch (op) {
    EW(0) EW(1) EW(2) EW(3) EW(4) EW(5) EW(6) EW(7) EW(8) EW(9) EW(10) EW(11) EW(12) EW(13) EW(14) EW(15) EW(16) EW(17) EW(18) EW(19) EW(20) EW(21) EW(22) EW(23) EW(24)
    default: return fail(EVMI_ERR_INVALID_ARG, "elementwise: unknown op");
  }
#undef EW
  EVMI_LAUNCH_CHECK("elementwise");
  return EVMI_OK;
}

/* out[0] (+)= scale * sum f;  mode 0: |a-b|, 1: (a-p)^2, 2: a   (fixed-shape two-pass reduction: reproducible) */
int evmi_scalar_reduce_f32(int mode, const float* a_dev, const float* b_dev, float* out_dev, long long n, float scale,
                           float p, int accumulate, void* stream) {
  EVMI_NONNULL(a_dev && out_dev, "scalar_reduce");
  hipStream_t s = (hipStream_t)stream;
  constexpr int MAXB = 1024;
  float* part_f = nullptr;
  if (int rc = splitk_ws(s, 2 * MAXB, &part_f)) return rc;
  double* part = reinterpret_cast<double*>(part_f);
  int nb = (int)((n + 256 * 8 - 1) / (256 * 8));
  if (nb < 1) nb = 1;
  if (nb > MAXB) nb = MAXB;
  if (mode == 0) hipLaunchKernelGGL(scalar_partial_kernel<0>, dim3(nb), dim3(256), 0, s, a_dev, b_dev, part, n, p);
  else if (mode == 1) hipLaunchKernelGGL(scalar_partial_kernel<1>, dim3(nb), dim3(256), 0, s, a_dev, b_dev, part, n, p);
  else if (mode == 2) hipLaunchKernelGGL(scalar_partial_kernel<2>, dim3(nb), dim3(256), 0, s, a_dev, b_dev, part, n, p);
  else return fail(EVMI_ERR_INVALID_ARG, "scalar_reduce: mode");
  EVMI_LAUNCH_CHECK("scalar_partial");
  hipLaunchKernelGGL(scalar_final_kernel, dim3(1), dim3(256), 0, s, part, nb, out_dev, scale, accumulate);
  EVMI_LAUNCH_CHECK("scalar_final");
  return EVMI_OK;
}

int evmi_avgpool4s2_f32(const float* x_dev, float* y_dev, long long rows, int t_in, int backward, void* stream) {
  EVMI_NONNULL(x_dev && y_dev, "avgpool4s2");
  const int t_out = t_in / 2 + 1;  // (t_in + 2*2 - 4) / 2 + 1
  const long long n = rows * (backward ? t_in : t_out);
  hipLaunchKernelGGL(avgpool4s2_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x_dev, y_dev, t_in, t_out, n, backward);
  EVMI_LAUNCH_CHECK("avgpool4s2");
  return EVMI_OK;
}

int evmi_period_view_f32(const float* x_dev, float* x2_dev, int B, int T, int period, int backward, void* stream) {
  EVMI_NONNULL(x_dev && x2_dev, "period_view");
  const int H = (T + period - 1) / period;
  if (!backward) {
    const long long n = (long long)B * period * H;
    hipLaunchKernelGGL(period_view_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x_dev, x2_dev, T, H, period, n);
  } else {  // x_dev = d(x2) [B*p][H], x2_dev = d(x) [B][T]
    const long long n = (long long)B * T;
    hipLaunchKernelGGL(period_view_bwd_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x_dev, x2_dev, T, H, period, n);
  }
  EVMI_LAUNCH_CHECK("period_view");
  return EVMI_OK;
}

int evmi_stft_frames_f32(const float* x_dev, float* frames_dev, int B, int T, int n_fft, int hop, int backward, void* stream) {
  EVMI_NONNULL(x_dev && frames_dev, "stft_frames");
  const int F = 1 + T / hop;
  if (!backward) {
    const long long n = (long long)n_fft * B * F;
    hipLaunchKernelGGL(frame_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x_dev, frames_dev, T, F, n_fft, hop, n);
  } else {  // x_dev = d(frames) [n_fft][B*F], frames_dev = d(x) [B][T]
    const long long n = (long long)B * T;
    hipLaunchKernelGGL(frame_bwd_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x_dev, frames_dev, B, T, F, n_fft, hop, n);
  }
  EVMI_LAUNCH_CHECK("stft_frames");
  return EVMI_OK;
}

int evmi_weight_norm_fwd_f32(const float* g_dev, const float* v_dev, float* w_dev, float* norm_dev, int rows, int n_per_row,
                             void* stream) {
  EVMI_NONNULL(g_dev && v_dev && w_dev && norm_dev, "weight_norm_fwd");
  hipLaunchKernelGGL(weight_norm_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g_dev, v_dev, w_dev, norm_dev, n_per_row);
  EVMI_LAUNCH_CHECK("weight_norm_fwd");
  return EVMI_OK;
}

int evmi_weight_norm_bwd_f32(const float* g_dev, const float* v_dev, const float* norm_dev, const float* dw_dev, float* dg_dev,
                             float* dv_dev, int rows, int n_per_row, void* stream) {
  EVMI_NONNULL(g_dev && v_dev && norm_dev && dw_dev && dg_dev && dv_dev, "weight_norm_bwd");
  hipLaunchKernelGGL(weight_norm_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, g_dev, v_dev, norm_dev, dw_dev, dg_dev, dv_dev, n_per_row);
  EVMI_LAUNCH_CHECK("weight_norm_bwd");
  return EVMI_OK;
}

int evmi_weight_norm_fwd_batched_f32(const float* flat_dev, float* eff_dev, float* norms_dev, const long long* table_dev, int n_layers,
                                     long long row_lo, long long row_hi, void* stream) {
  EVMI_NONNULL(flat_dev && eff_dev && norms_dev && table_dev, "weight_norm_fwd_batched");
  if (n_layers <= 0 || row_hi <= row_lo || row_hi - row_lo > 0x7fffffffLL) return fail(EVMI_ERR_INVALID_ARG, "weight_norm_fwd_batched: rows");
  hipLaunchKernelGGL(weight_norm_fwd_batched_kernel, dim3((unsigned)(row_hi - row_lo)), dim3(256), 0, (hipStream_t)stream, flat_dev, eff_dev,
                     norms_dev, table_dev, n_layers, row_lo);
  EVMI_LAUNCH_CHECK("weight_norm_fwd_batched");
  return EVMI_OK;
}

int evmi_weight_norm_bwd_batched_f32(const float* flat_dev, float* grad_dev, const float* norms_dev, float* dw_eff_dev,
                                     const long long* table_dev, int n_layers, long long row_lo, long long row_hi, void* stream) {
  EVMI_NONNULL(flat_dev && grad_dev && norms_dev && dw_eff_dev && table_dev, "weight_norm_bwd_batched");
  if (n_layers <= 0 || row_hi <= row_lo || row_hi - row_lo > 0x7fffffffLL) return fail(EVMI_ERR_INVALID_ARG, "weight_norm_bwd_batched: rows");
  hipLaunchKernelGGL(weight_norm_bwd_batched_kernel, dim3((unsigned)(row_hi - row_lo)), dim3(256), 0, (hipStream_t)stream, flat_dev, grad_dev,
                     norms_dev, dw_eff_dev, table_dev, n_layers, row_lo);
  EVMI_LAUNCH_CHECK("weight_norm_bwd_batched");
  return EVMI_OK;
}

int evmi_istft_polar_f32(const float* a_dev, float* s_dev, int H, long long n, void* stream) {
  EVMI_NONNULL(a_dev && s_dev, "istft_polar");
  if (H <= 0 || n <= 0) return fail(EVMI_ERR_INVALID_ARG, "istft_polar: shape");
  hipLaunchKernelGGL(istft_polar_kernel, grid1d((long long)H * n), dim3(256), 0, (hipStream_t)stream, a_dev, s_dev, H, n);
  EVMI_LAUNCH_CHECK("istft_polar");
  return EVMI_OK;
}

int evmi_istft_polar_bwd_f32(const float* a_dev, const float* ds_dev, float* da_dev, int H, long long n, void* stream) {
  EVMI_NONNULL(a_dev && ds_dev && da_dev, "istft_polar_bwd");
  if (H <= 0 || n <= 0) return fail(EVMI_ERR_INVALID_ARG, "istft_polar_bwd: shape");
  hipLaunchKernelGGL(istft_polar_bwd_kernel, grid1d((long long)H * n), dim3(256), 0, (hipStream_t)stream, a_dev, ds_dev, da_dev, H, n);
  EVMI_LAUNCH_CHECK("istft_polar_bwd");
  return EVMI_OK;
}

int evmi_reflect_pad_left1_f32(const float* src_dev, float* dst_dev, long long rows, int T, int backward, void* stream) {
  EVMI_NONNULL(src_dev && dst_dev, "reflect_pad_left1");
  if (rows <= 0 || T < 2) return fail(EVMI_ERR_INVALID_ARG, "reflect_pad_left1: needs at least two positions per row");
  hipLaunchKernelGGL(reflect_pad_left1_kernel, grid1d(rows * (backward ? T : T + 1)), dim3(256), 0, (hipStream_t)stream, src_dev, dst_dev,
                     rows, T, backward);
  EVMI_LAUNCH_CHECK("reflect_pad_left1");
  return EVMI_OK;
}

int evmi_normalize_vec_f32(const float* x_dev, float* y_dev, int n, float eps, void* stream) {
  EVMI_NONNULL(x_dev && y_dev, "normalize_vec");
  hipLaunchKernelGGL(normalize_vec_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x_dev, y_dev, n, eps);
  EVMI_LAUNCH_CHECK("normalize_vec");
  return EVMI_OK;
}

static int optimizer_step_impl(int kind, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, const int* step_dev, float clip, const float* lr_dev,
                               void* stream);

int evmi_optimizer_step_f32(int kind, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n, float lr, float beta1,
                            float beta2, float eps, float weight_decay, int step, const int* step_dev, float clip, void* stream) {
  return optimizer_step_impl(kind, p_dev, g_dev, m_dev, v_dev, n, lr, beta1, beta2, eps, weight_decay, step, step_dev, clip, nullptr, stream);
}

int evmi_optimizer_step_lrdev_f32(int kind, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n, const float* lr_dev,
                                  float beta1, float beta2, float eps, float weight_decay, const int* step_dev, float clip, void* stream) {
  EVMI_NONNULL(lr_dev && step_dev, "optimizer_step_lrdev");
  return optimizer_step_impl(kind, p_dev, g_dev, m_dev, v_dev, n, 0.f, beta1, beta2, eps, weight_decay, 0, step_dev, clip, lr_dev, stream);
}

__global__ void store_f32_kernel(float* dst, int n, float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7) {
  const float v[8] = {v0, v1, v2, v3, v4, v5, v6, v7};
  for (int i = 0; i < n; ++i) dst[i] = v[i];
}
__global__ void store_u64_kernel(unsigned long long* dst, unsigned long long v) { *dst = v; }

int evmi_store_f32(float* dst_dev, int n, const float* values_host, void* stream) {
  EVMI_NONNULL(dst_dev && values_host, "store_f32");
  if (n < 1 || n > 8) return fail(EVMI_ERR_INVALID_ARG, "store_f32: 1..8 values");
  float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) v[i] = values_host[i];  // copied NOW: they travel in the kernel's argument block, not through host memory later
  hipLaunchKernelGGL(store_f32_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst_dev, n, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
  EVMI_LAUNCH_CHECK("store_f32");
  return EVMI_OK;
}

int evmi_store_u64(unsigned long long* dst_dev, unsigned long long value, void* stream) {
  EVMI_NONNULL(dst_dev, "store_u64");
  hipLaunchKernelGGL(store_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, dst_dev, value);
  EVMI_LAUNCH_CHECK("store_u64");
  return EVMI_OK;
}

static int optimizer_step_impl(int kind, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n, float lr, float beta1,
                               float beta2, float eps, float weight_decay, int step, const int* step_dev, float clip, const float* lr_dev,
                               void* stream) {
  EVMI_NONNULL(p_dev && g_dev && v_dev && (kind == 2 || m_dev), "optimizer_step");
  if (kind < 0 || kind > 2) return fail(EVMI_ERR_INVALID_ARG, "optimizer_step: kind (0 AdamW, 1 Adam, 2 RMSprop)");
  if ((reinterpret_cast<uintptr_t>(p_dev) | reinterpret_cast<uintptr_t>(g_dev) | reinterpret_cast<uintptr_t>(m_dev) | reinterpret_cast<uintptr_t>(v_dev)) & 15)
    return fail(EVMI_ERR_INVALID_ARG, "optimizer_step: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(optimizer_step_kernel, grid1d((n >> 2) + 3), dim3(256), 0, (hipStream_t)stream, kind, p_dev, g_dev, m_dev, v_dev, n, lr, beta1,
                     beta2, eps, weight_decay, step, step_dev, clip, lr_dev);
  EVMI_LAUNCH_CHECK("optimizer_step");
  return EVMI_OK;
}

int evmi_transpose_bct_cbt_f32(const float* in_dev, float* out_dev, int B, int C, int T, void* stream) {
  EVMI_NONNULL(in_dev && out_dev, "transpose_bct_cbt");
  const long long n = (long long)B * C * T;
  hipLaunchKernelGGL(transpose_bct_cbt_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, B, C, T, n);
  EVMI_LAUNCH_CHECK("transpose_bct_cbt");
  return EVMI_OK;
}

int evmi_counter_add_i32(int* counter_dev, int delta, void* stream) {
  EVMI_NONNULL(counter_dev, "counter_add");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, counter_dev, delta);
  EVMI_LAUNCH_CHECK("counter_add");
  return EVMI_OK;
}

int evmi_spectral_norm_grad_f32(float* gw_dev, const float* dw_dev, const float* u_dev, const float* v_dev, const float* sigma_dev,
                                const float* dot_dev, int rows, int cols, void* stream) {
  EVMI_NONNULL(gw_dev && dw_dev && u_dev && v_dev && sigma_dev && dot_dev, "spectral_norm_grad");
  const long long n = (long long)rows * cols;
  hipLaunchKernelGGL(sn_grad_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, gw_dev, dw_dev, u_dev, v_dev, sigma_dev, dot_dev, cols, n);
  EVMI_LAUNCH_CHECK("spectral_norm_grad");
  return EVMI_OK;
}

int evmi_ratio_accumulate_f32(float* out_dev, const float* sq_dev, float weight, void* stream) {
  EVMI_NONNULL(out_dev && sq_dev, "ratio_accumulate");
  hipLaunchKernelGGL(ratio_acc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, out_dev, sq_dev, weight);
  EVMI_LAUNCH_CHECK("ratio_accumulate");
  return EVMI_OK;
}

int evmi_adamw_f32(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, void* stream) {
  EVMI_NONNULL(p_dev && g_dev && m_dev && v_dev, "adamw");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, p_dev, g_dev, m_dev, v_dev, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2);
  EVMI_LAUNCH_CHECK("adamw");
  return EVMI_OK;
}

}  // extern "C"
