// Alignment learning, backward side (Badlani et al. 2021 as implemented in FastPitch's ConvAttention /
// AttentionCTCLoss / AttentionBinarizationLoss; the forward kernels are in fs2_ops.hip).
//
//   forward_sum_grad_kernel   CTC forward-sum loss AND its gradient w.r.t. the alignment log-probabilities: alpha pass
//                             (stored), beta pass, occupancies -> d logprob.  One workgroup per item, the 2 L_b + 1 states of
//                             the extended target in parallel, frames in sequence (both directions).
//   align_attention_bwd_kernel  one workgroup per (item, frame) row: binarisation-loss gradient on the hard path, softmax and
//                             log-softmax backward -> d score row + its row sum
//   align_colsum_kernel / align_qk_grad_kernel   the distance scores' gradient w.r.t. the projected mel and text:
//                             dq = -2 temp (q * rowsum - K . da^T), dk = 2 temp (Q . da - k * colsum); the two products are
//                             batched GEMMs issued by the host between these kernels.
#include <cmath>

#include "common.h"
#include "evmi.h"

namespace evmi {

__device__ __forceinline__ float lae(float a, float b) {
  if (a == -INFINITY) return b;
  if (b == -INFINITY) return a;
  const float m = fmaxf(a, b);
  return m + log1pf(expf(-fabsf(a - b)));
}

// lse[b][t] = logsumexp over [blank, tokens < L_b] of frame t: one wave per frame
__global__ __launch_bounds__(256) void forward_sum_lse_kernel(const float* __restrict__ logprob, const int* __restrict__ text_lens,
                                                             const int* __restrict__ mel_lens, float* __restrict__ lse, int B, int T, int L,
                                                             float blank_logprob) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= (long long)B * T) return;
  const int b = (int)(row / T), t = (int)(row - (long long)b * T);
  if (t >= mel_lens[b]) return;
  const int Lb = min(text_lens[b], L);
  const float* r = logprob + row * L;
  float m = blank_logprob;
  for (int l = lane; l < Lb; l += 64) m = fmaxf(m, r[l]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  float s = lane == 0 ? expf(blank_logprob - m) : 0.f;
  for (int l = lane; l < Lb; l += 64) s += expf(r[l] - m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
  if (lane == 0) lse[row] = m + logf(s);
}

// loss[b] = -log p(target) / L_b ; grad[b][t][l] = scale[b] * (softmax row - state occupancy), scale = weight / (B * L_b).
// One workgroup of 512 threads per item, NS states of the extended target per thread; every thread fetches the emission of
// its own states one frame ahead, so a frame costs one barrier (double-buffered lattice column in LDS).
// alpha_ws [B][T][2L+1], lse [B][T] from the kernel above.
constexpr int CTC_THREADS = 512;
template <int NS>
__global__ __launch_bounds__(CTC_THREADS) void forward_sum_grad_kernel(const float* __restrict__ logprob, const int* __restrict__ text_lens,
                                                                      const int* __restrict__ mel_lens, float* __restrict__ loss,
                                                                      float* __restrict__ grad, float* __restrict__ alpha_ws,
                                                                      const float* __restrict__ lse_ws, int B, int T, int L,
                                                                      float blank_logprob, float weight) {
  extern __shared__ float sm[];  // 2 x [2L + 1] lattice columns
  const int b = blockIdx.x, tid = threadIdx.x;
  const int Lb = min(text_lens[b], L), Tb = min(mel_lens[b], T), S = 2 * Lb + 1;
  const int SW = 2 * L + 1;
  float* buf0 = sm;
  float* buf1 = sm + SW;
  float* gb = grad + (long long)b * T * L;
  for (long long i = tid; i < (long long)T * L; i += CTC_THREADS) gb[i] = 0.f;
  if (Lb <= 0 || Tb <= 0) { if (tid == 0) loss[b] = 0.f; return; }
  float* aw = alpha_ws + (long long)b * T * SW;
  const float* lw = lse_ws + (long long)b * T;
  const float* lpb = logprob + (long long)b * T * L;
  auto emission = [&](int t, int st) {  // normalised log-probability of state st's label at frame t
    const float raw = (st & 1) ? lpb[(long long)t * L + ((st - 1) >> 1)] : blank_logprob;
    return raw - lw[t];
  };
  float y[NS], yn[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int st = tid + j * CTC_THREADS;
    y[j] = st < S ? emission(0, st) : 0.f;
    yn[j] = (st < S && Tb > 1) ? emission(1, st) : 0.f;
  }
  for (int t = 0; t < Tb; ++t) {  // alpha
    float* cur = (t & 1) ? buf1 : buf0;
    const float* prev = (t & 1) ? buf0 : buf1;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int st = tid + j * CTC_THREADS;
      if (st < S) {
        float a;
        if (t == 0) a = st < 2 ? 0.f : -INFINITY;
        else {
          a = prev[st];
          if (st >= 1) a = lae(a, prev[st - 1]);
          if ((st & 1) && st >= 3) a = lae(a, prev[st - 2]);
        }
        a += y[j];
        cur[st] = a;
        aw[(long long)t * SW + st] = a;
        y[j] = yn[j];
        if (t + 2 < Tb) yn[j] = emission(t + 2, st);
      }
    }
    __syncthreads();
  }
  const float* fin = ((Tb - 1) & 1) ? buf1 : buf0;
  const float ll = S >= 2 ? lae(fin[S - 1], fin[S - 2]) : fin[S - 1];
  __syncthreads();
  if (tid == 0) loss[b] = ll == -INFINITY ? 0.f : -ll / (float)Lb;
  if (ll == -INFINITY) return;  // zero_infinity: no gradient either
  const float scale = weight / ((float)B * (float)Lb);

  float al[NS], aln[NS];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const int st = tid + j * CTC_THREADS;
    y[j] = st < S ? emission(Tb - 1, st) : 0.f;
    yn[j] = (st < S && Tb > 1) ? emission(Tb - 2, st) : 0.f;
    al[j] = st < S ? aw[(long long)(Tb - 1) * SW + st] : 0.f;
    aln[j] = (st < S && Tb > 1) ? aw[(long long)(Tb - 2) * SW + st] : 0.f;
  }
  for (int t = Tb - 1; t >= 0; --t) {  // beta (emission at t included, like alpha) + gradient of frame t
    float* cur = (t & 1) ? buf1 : buf0;
    const float* nxt = (t & 1) ? buf0 : buf1;
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int st = tid + j * CTC_THREADS;
      if (st < S) {
        float v;
        if (t == Tb - 1) v = st >= S - 2 ? 0.f : -INFINITY;
        else {
          v = nxt[st];
          if (st + 1 < S) v = lae(v, nxt[st + 1]);
          if ((st & 1) && st + 2 < S) v = lae(v, nxt[st + 2]);
        }
        v += y[j];
        cur[st] = v;
        if (st & 1) {  // token l = (st - 1) / 2: one state per token
          const float occ = expf(al[j] + v - y[j] - ll);
          gb[(long long)t * L + ((st - 1) >> 1)] = scale * (expf(y[j]) - occ);
        }
        y[j] = yn[j];
        al[j] = aln[j];
        if (t >= 2) {
          yn[j] = emission(t - 2, st);
          aln[j] = aw[(long long)(t - 2) * SW + st];
        }
      }
    }
    __syncthreads();
  }
}

// One workgroup per (item, frame) row.  z = log_softmax_all(a) + log(prior + 1e-8) (or z = a without a prior) is what the
// forward stored as logprob; soft = softmax over the unpadded tokens of z.
//   d soft  = -bin_scale / soft on the hard path (soft > 1e-12)         (binarisation loss)
//   d z     = soft * (d soft - sum soft * d soft)  +  dlogprob            (+ the CTC gradient)
//   d a     = d z - softmax_all(a) * sum d z   with softmax_all(a) = exp(logprob) / (prior + 1e-8)
__global__ __launch_bounds__(256) void align_attention_bwd_kernel(const float* __restrict__ soft, const float* __restrict__ logprob,
                                                                 const double* __restrict__ prior, const int* __restrict__ hard,
                                                                 const float* __restrict__ dlogprob, const int* __restrict__ text_lens,
                                                                 float* __restrict__ da, float* __restrict__ rowsum, int T, int L,
                                                                 float bin_scale, const float* __restrict__ bin_count) {
  __shared__ float red[4];
  if (bin_count) bin_scale /= *bin_count;  // weight / (number of hard cells), the count left on the device
  const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
  const long long row = ((long long)b * T + t) * L;
  const int len = min(text_lens[b], L);
  auto block_sum = [&](float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
  };
  float acc = 0.f;  // sum soft * dsoft = -bin_scale * (number of hard cells with soft > 1e-12)
  if (hard && bin_scale != 0.f)
    for (int l = tid; l < len; l += 256)
      if (hard[row + l] == 1 && soft[row + l] > 1e-12f) acc -= bin_scale;
  const float sds = block_sum(acc);
  float part = 0.f;
  for (int l = tid; l < L; l += 256) {
    float dz = dlogprob ? dlogprob[row + l] : 0.f;
    if (l < len && hard && bin_scale != 0.f) {
      const float s = soft[row + l];
      const float ds = (hard[row + l] == 1 && s > 1e-12f) ? -bin_scale / s : 0.f;
      dz += s * (ds - sds);
    }
    da[row + l] = dz;
    part += dz;
  }
  const float sdz = block_sum(part);
  float rs = sdz;
  if (prior) {
    float p2 = 0.f;
    for (int l = tid; l < L; l += 256) {
      const float pa = expf(logprob[row + l] - logf((float)prior[row + l] + 1e-8f));
      const float v = da[row + l] - pa * sdz;
      da[row + l] = v;
      p2 += v;
    }
    rs = block_sum(p2);
  }
  if (tid == 0) rowsum[(long long)b * T + t] = rs;
}

// colsum[b][l] = sum_t da[b][t][l]
__global__ void align_colsum_kernel(const float* __restrict__ da, float* __restrict__ colsum, int B, int T, int L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, l = i - b * L;
  const float* p = da + (long long)b * T * L + l;
  colsum[i] = ordered_sum_strided(p, L, T);  // (same order of additions, sixteen loads in flight: 238 -> ~30 us at 947 frames)
}

// in place: m[c][b][n] = coef * (x[c][b][n] * sums[b][n] - m[c][b][n])
__global__ void align_qk_grad_kernel(const float* __restrict__ x, const float* __restrict__ sums, float* __restrict__ m, int A, long long BN,
                                     float coef) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)A * BN) return;
  m[i] = coef * (x[i] * sums[i % BN] - m[i]);
}

}  // namespace evmi

using namespace evmi;

extern "C" {

long long evmi_forward_sum_grad_f32_ws_elems(int B, int T, int L) { return (long long)B * T * (2 * L + 1) + (long long)B * T; }

int evmi_forward_sum_grad_f32(const float* logprob, const int* text_lens, const int* mel_lens, float* loss_per_item, float* grad, float* ws,
                              long long ws_elems, int B, int T, int L, float blank_logprob, float weight, void* stream) {
  if (!logprob || !text_lens || !mel_lens || !loss_per_item || !grad || !ws) return fail(EVMI_ERR_INVALID_ARG, "forward_sum_grad: null pointer");
  if (B <= 0 || T <= 0 || L <= 0) return fail(EVMI_ERR_INVALID_ARG, "forward_sum_grad: shape");
  if (ws_elems < evmi_forward_sum_grad_f32_ws_elems(B, T, L)) return fail(EVMI_ERR_INVALID_ARG, "forward_sum_grad: workspace too small");
  const size_t lds = (size_t)(2 * (2 * L + 1)) * sizeof(float);
  if (2 * L + 1 > 4 * CTC_THREADS) return fail(EVMI_ERR_UNSUPPORTED, "forward_sum_grad: more than 1023 tokens");
  hipStream_t s = (hipStream_t)stream;
  float* lse = ws + (long long)B * T * (2 * L + 1);
  hipLaunchKernelGGL(forward_sum_lse_kernel, dim3((unsigned)(((long long)B * T + 3) / 4)), dim3(256), 0, s, logprob, text_lens, mel_lens, lse, B, T, L,
                     blank_logprob);
  EVMI_LAUNCH_CHECK("forward_sum_lse");
  const int ns = (2 * L + 1 + CTC_THREADS - 1) / CTC_THREADS;
#define EVMI_CTC(NSV)                                                                                                                    \
  hipLaunchKernelGGL(forward_sum_grad_kernel<NSV>, dim3(B), dim3(CTC_THREADS), lds, s, logprob, text_lens, mel_lens, loss_per_item, grad, ws, \
                     lse, B, T, L, blank_logprob, weight)
  if (ns == 1) EVMI_CTC(1);
  else if (ns == 2) EVMI_CTC(2);
  else EVMI_CTC(4);
#undef EVMI_CTC
  EVMI_LAUNCH_CHECK("forward_sum_grad");
  return EVMI_OK;
}

int evmi_align_attention_bwd_f32(const float* soft, const float* logprob, const double* prior, const int* hard, const float* dlogprob,
                                 const int* text_lens, float* da, float* rowsum, float* colsum, int B, int T, int L, float bin_scale,
                                 const float* bin_count_dev, void* stream) {
  if (!soft || !logprob || !text_lens || !da || !rowsum || !colsum) return fail(EVMI_ERR_INVALID_ARG, "align_attention_bwd: null pointer");
  if (B <= 0 || T <= 0 || L <= 0 || B > 65535) return fail(EVMI_ERR_INVALID_ARG, "align_attention_bwd: shape");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(align_attention_bwd_kernel, dim3(T, B), dim3(256), 0, s, soft, logprob, prior, hard, dlogprob, text_lens, da, rowsum, T, L,
                     bin_scale, bin_count_dev);
  EVMI_LAUNCH_CHECK("align_attention_bwd");
  hipLaunchKernelGGL(align_colsum_kernel, dim3((B * L + 255) / 256), dim3(256), 0, s, da, colsum, B, T, L);
  EVMI_LAUNCH_CHECK("align_colsum");
  return EVMI_OK;
}

int evmi_align_qk_grad_f32(const float* x, const float* sums, float* m, int A, long long BN, float coef, void* stream) {
  if (!x || !sums || !m || A <= 0 || BN <= 0) return fail(EVMI_ERR_INVALID_ARG, "align_qk_grad: bad arguments");
  hipLaunchKernelGGL(align_qk_grad_kernel, dim3((unsigned)(((long long)A * BN + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, sums, m, A, BN,
                     coef);
  EVMI_LAUNCH_CHECK("align_qk_grad");
  return EVMI_OK;
}

}  // extern "C"
