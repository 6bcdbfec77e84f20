// Shared by the packed bf16 convolution kernels (conv_cbt_bf16_pk.hip, conv_wgrad_bf16_pk.hip): the packed operand layout
// (16-byte units of 8 channels at one position), its pack kernel, and the LDS-direct 16-byte load.
#pragma once

#include "common.h"

namespace evmi {

typedef __attribute__((address_space(3))) uint4 lds_u4_t;
typedef __attribute__((address_space(1))) const uint4 glb_u4_t;
__device__ __forceinline__ void pk_lds_direct(const uint4* g, uint4* l) {
  __builtin_amdgcn_global_load_lds((glb_u4_t*)g, (lds_u4_t*)l, 16, 0, 0);
}

__device__ __forceinline__ unsigned pk_bf16x2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 v;
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}

// x [groups*cin_g][B][t_in] fp32 -> xp [groups][octs][B][Tp] units; unit (g, o, b, u) = channels g*cin_g + 8o..8o+7 at t = u - PL.
// grid (ceil(Tp / 256), B, groups*octs): one thread per unit, consecutive threads consecutive u (coalesced reads of 8 channel
// rows, 16-byte writes); no index divisions.  The workgroups of the last row also zero `slack_units` units behind the tensor.
static __global__ __launch_bounds__(256) void pack_x_kernel(const float* __restrict__ x, uint4* __restrict__ xp, int cin_g, int octs, int B,
                                                     int t_in, int Tp, int PL, int slack_units) {
  const int u = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y, go = blockIdx.z;
  if (slack_units > 0 && b == B - 1 && go == (int)gridDim.z - 1) {
    // the units behind the last row are read by the last tiles' window loads (never used): keep them finite
    uint4* tail = xp + (long long)gridDim.z * B * Tp;
    for (int i = u; i < slack_units; i += gridDim.x * 256) tail[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (u >= Tp) return;
  const int g = go / octs, o = go - g * octs;
  const int t = u - PL;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (t >= 0 && t < t_in) {
    const float* src = x + ((long long)(g * cin_g + o * 8) * B + b) * t_in + t;
    const long long cs = (long long)B * t_in;
    const int nch = min(8, cin_g - o * 8);
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i < nch) v[i] = src[i * cs];
  }
  uint4 out;
  out.x = pk_bf16x2(v[0], v[1]);
  out.y = pk_bf16x2(v[2], v[3]);
  out.z = pk_bf16x2(v[4], v[5]);
  out.w = pk_bf16x2(v[6], v[7]);
  xp[((long long)go * B + b) * Tp + u] = out;
}


}  // namespace evmi
