// Layout change at the generator's entrance and the 1-channel output convolution at its exit.
// Both are HBM-bound byte movers: full-row coalesced accesses, LDS for the transposition / halo.
#include "common.h"

namespace evmi {

// mel [B][C][T] fp32 (torch layout)  ->  x [B][T][C] bf16 (time-major, channel-last)
__global__ __launch_bounds__(256) void nct_f32_to_tc_bf16_kernel(const float* __restrict__ in,
                                                                 bf16_t* __restrict__ out, int C, int T) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tile = reinterpret_cast<float*>(smem);  // [64][C + 1]
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int c = ty; c < C; c += 4) {
    const int t = t0 + tx;
    tile[tx * (C + 1) + c] = t < T ? in[((long long)b * C + c) * T + t] : 0.f;
  }
  __syncthreads();
  const int n = 64 * C;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int t = i / C, c = i % C;
    if (t0 + t < T) out[((long long)b * T + t0 + t) * C + c] = (bf16_t)tile[t * (C + 1) + c];
  }
}

// wav[b, t] = tanh( bias + sum_{j<KS} sum_{c<CIN} w[j][c] * pre(x[b, t + j - KS/2, c]) ),  x bf16 [B][T][CIN]
template <int CIN, int KS>
__global__ __launch_bounds__(256) void conv_post_tanh_kernel(const bf16_t* __restrict__ x,
                                                             const float* __restrict__ w,  // [KS][CIN]
                                                             float bias, float* __restrict__ wav,
                                                             int T, float pre_slope) {
  constexpr int BN = 256, HALO = KS / 2, R = BN + KS - 1, XS = CIN + 8;
  __shared__ __attribute__((aligned(16))) bf16_t Xs[R * XS];
  __shared__ float Ws[KS * CIN];
  const int b = blockIdx.y, t0 = blockIdx.x * BN, tid = threadIdx.x;
  const bf16_t* xb = x + (long long)b * T * CIN;
  for (int v = tid; v < R * (CIN / 8); v += 256) {
    const int i = v / (CIN / 8), c8 = v % (CIN / 8);
    const int rr = t0 - HALO + i;
    bf16x8 val;
    if (rr >= 0 && rr < T) {
      val = *reinterpret_cast<const bf16x8*>(xb + (long long)rr * CIN + c8 * 8);
      if (pre_slope != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) val[e] = (bf16_t)lrelu((float)val[e], pre_slope);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] = (bf16_t)0.f;
    }
    *reinterpret_cast<bf16x8*>(Xs + i * XS + c8 * 8) = val;
  }
  for (int i = tid; i < KS * CIN; i += 256) Ws[i] = w[i];
  __syncthreads();
  const int t = t0 + tid;
  if (t >= T) return;
  float acc = bias;
#pragma unroll
  for (int j = 0; j < KS; ++j) {
#pragma unroll
    for (int c8 = 0; c8 < CIN / 8; ++c8) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(Xs + (tid + j) * XS + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) acc = fmaf(Ws[j * CIN + c8 * 8 + e], (float)v[e], acc);
    }
  }
  wav[(long long)b * T + t] = tanhf(acc);
}

int launch_nct_f32_to_tc_bf16(const float* in, bf16_t* out, int B, int C, int T, hipStream_t s) {
  const size_t lds = (size_t)64 * (C + 1) * sizeof(float);
  hipLaunchKernelGGL(nct_f32_to_tc_bf16_kernel, dim3((T + 63) / 64, B), dim3(256), lds, s, in, out, C, T);
  EVMI_LAUNCH_CHECK("nct_f32_to_tc_bf16");
  return EVMI_OK;
}

int launch_conv_post_tanh(const bf16_t* x, const float* w_kc, float bias, float* wav, int B, int T,
                          int c_in, int ks, float pre_slope, hipStream_t s) {
  dim3 grid((T + 255) / 256, B);
  if (ks == 7 && c_in == 32) {
    hipLaunchKernelGGL((conv_post_tanh_kernel<32, 7>), grid, dim3(256), 0, s, x, w_kc, bias, wav, T, pre_slope);
  } else if (ks == 7 && c_in == 64) {
    hipLaunchKernelGGL((conv_post_tanh_kernel<64, 7>), grid, dim3(256), 0, s, x, w_kc, bias, wav, T, pre_slope);
  } else if (ks == 7 && c_in == 16) {
    hipLaunchKernelGGL((conv_post_tanh_kernel<16, 7>), grid, dim3(256), 0, s, x, w_kc, bias, wav, T, pre_slope);
  } else if (ks == 7 && c_in == 128) {
    hipLaunchKernelGGL((conv_post_tanh_kernel<128, 7>), grid, dim3(256), 0, s, x, w_kc, bias, wav, T, pre_slope);
  } else {
    return fail(EVMI_ERR_UNSUPPORTED, "conv_post_tanh: unsupported (c_in, k)");
  }
  EVMI_LAUNCH_CHECK("conv_post_tanh");
  return EVMI_OK;
}

}  // namespace evmi
