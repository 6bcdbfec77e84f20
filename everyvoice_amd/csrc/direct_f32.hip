// Direct fp32 convolutions on torch-native [B, C, T] layouts: the exact-arithmetic path
// (EVMI_PREC_F32) and the generic fallback shape coverage (stride / dilation / groups).
// One thread per output element, fmaf chain over (c_in, k) in that order; reads of x are
// coalesced along t, weights are wave-uniform (scalar loads).
#include "common.h"

namespace evmi {

__global__ __launch_bounds__(256) void conv1d_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ res, float* __restrict__ y, int c_in, int t_in, int c_out, int t_out,
    int k, int stride, int pad, int dil, int groups, float pre_slope, float out_scale, int accumulate) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int co = blockIdx.y;
  const int b = blockIdx.z;
  if (t >= t_out) return;
  const int cin_g = c_in / groups;
  const int g = co / (c_out / groups);
  const float* xb = x + ((long long)b * c_in + (long long)g * cin_g) * t_in;
  const float* wc = w + (long long)co * cin_g * k;
  float acc = 0.f;
  const int base = t * stride - pad;
  for (int ci = 0; ci < cin_g; ++ci) {
    const float* xr = xb + (long long)ci * t_in;
    for (int j = 0; j < k; ++j) {
      const int ti = base + j * dil;
      if (ti >= 0 && ti < t_in) {
        float v = xr[ti];
        v = v > 0.f ? v : v * pre_slope;
        acc = fmaf(wc[ci * k + j], v, acc);
      }
    }
  }
  const long long o = ((long long)b * c_out + co) * t_out + t;
  if (bias) acc += bias[co];
  if (res) acc += res[o];
  acc *= out_scale;
  if (accumulate) acc += y[o];
  y[o] = acc;
}

__global__ __launch_bounds__(256) void conv_transpose1d_f32_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
    float* __restrict__ y, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad,
    float pre_slope) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int co = blockIdx.y;
  const int b = blockIdx.z;
  if (t >= t_out) return;
  const float* xb = x + (long long)b * c_in * t_in;
  float acc = 0.f;
  // y[t] = sum_{ci} sum_{j : (t + pad - j) % stride == 0} w[ci][co][j] * x[ci][(t + pad - j)/stride]
  const int tp = t + pad;
  const int j0 = tp % stride;
  for (int ci = 0; ci < c_in; ++ci) {
    const float* xr = xb + (long long)ci * t_in;
    const float* wr = w + ((long long)ci * c_out + co) * k;
    for (int j = j0; j < k; j += stride) {
      const int ti = (tp - j) / stride;
      if (ti >= 0 && ti < t_in) {
        float v = xr[ti];
        v = v > 0.f ? v : v * pre_slope;
        acc = fmaf(wr[j], v, acc);
      }
    }
  }
  if (bias) acc += bias[co];
  y[((long long)b * c_out + co) * t_out + t] = acc;
}

__global__ __launch_bounds__(256) void tanh_f32_kernel(float* __restrict__ y, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = tanhf(y[i]);
}

int launch_conv1d_f32(const float* x, const float* w, const float* bias, const float* res, float* y,
                      int B, int c_in, int t_in, int c_out, int k, int stride, int pad, int dil,
                      int groups, float pre_slope, float out_scale, int accumulate, hipStream_t s) {
  if (B <= 0 || c_in <= 0 || c_out <= 0 || k <= 0 || stride <= 0 || dil <= 0 || groups <= 0 ||
      c_in % groups || c_out % groups)
    return fail(EVMI_ERR_INVALID_ARG, "conv1d_f32: bad shape");
  const int t_out = (t_in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
  if (t_out <= 0) return fail(EVMI_ERR_INVALID_ARG, "conv1d_f32: empty output");
  dim3 grid((t_out + 255) / 256, c_out, B);
  hipLaunchKernelGGL(conv1d_f32_kernel, grid, dim3(256), 0, s, x, w, bias, res, y, c_in, t_in, c_out,
                     t_out, k, stride, pad, dil, groups, pre_slope, out_scale, accumulate);
  EVMI_LAUNCH_CHECK("conv1d_f32");
  return EVMI_OK;
}

int launch_conv_transpose1d_f32(const float* x, const float* w, const float* bias, float* y, int B,
                                int c_in, int t_in, int c_out, int k, int stride, int pad,
                                float pre_slope, hipStream_t s) {
  if (B <= 0 || c_in <= 0 || c_out <= 0 || k <= 0 || stride <= 0)
    return fail(EVMI_ERR_INVALID_ARG, "conv_transpose1d_f32: bad shape");
  const int t_out = (t_in - 1) * stride - 2 * pad + k;
  if (t_out <= 0) return fail(EVMI_ERR_INVALID_ARG, "conv_transpose1d_f32: empty output");
  dim3 grid((t_out + 255) / 256, c_out, B);
  hipLaunchKernelGGL(conv_transpose1d_f32_kernel, grid, dim3(256), 0, s, x, w, bias, y, c_in, t_in,
                     c_out, t_out, k, stride, pad, pre_slope);
  EVMI_LAUNCH_CHECK("conv_transpose1d_f32");
  return EVMI_OK;
}

int launch_tanh_f32(float* y, long long n, hipStream_t s) {
  hipLaunchKernelGGL(tanh_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, y, n);
  EVMI_LAUNCH_CHECK("tanh_f32");
  return EVMI_OK;
}

}  // namespace evmi
