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

// Tail of a flat packed output for four channels of one unit: v = (v + fm_scale * sign(xf - xr)) * lrelu'(xf), xf / xr the four
// bf16 values of the mask tensor (the layer's own activated output) and of the feature-matching reference (xr = xf: no term).
__device__ __forceinline__ void pk_flat_tail4(float* v, uint2 mk, uint2 fr, float fm_scale, float mask_slope) {
#pragma clang fp contract(off)
  const float xf[4] = {bf16_lo(mk.x), bf16_hi(mk.x), bf16_lo(mk.y), bf16_hi(mk.y)};
  const float xr[4] = {bf16_lo(fr.x), bf16_hi(fr.x), bf16_lo(fr.y), bf16_hi(fr.y)};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float dlt = xf[e] - xr[e];
    const float sg = dlt > 0.f ? fm_scale : (dlt < 0.f ? -fm_scale : 0.f);
    const float t = v[e] + sg;
    v[e] = t * (xf[e] > 0.f ? 1.f : mask_slope);
  }
}

// x [groups*cin_g][B][t_in] fp32 -> xp [groups][octs][B][Tp] units; unit (g, o, b, u) = channels g*cin_g + 8o..8o+7 at t = u - PL.
// Logical grid (ceil(Tp / 256), B, groups*octs): one thread per unit, consecutive threads consecutive u (coalesced reads of 8
// channel rows, 16-byte writes).  The workgroups of the last row also zero `slack_units` units behind the tensor (read by the
// last tiles' window loads, never used: kept finite).
// Fused on the way in: pre_slope != 1 packs leaky_relu(x, pre_slope) (the activation in front of a convolution: no separate
// pass, no activated copy in HBM); mask != nullptr packs x * (mask > 0 ? 1 : mask_slope) (a gradient through the leaky ReLU
// whose OUTPUT is `mask`: the activation backward folded into the pack of dy).
struct PackArgs {
  const float* x;
  uint4* xp;
  int cin_g, octs, B, t_in, Tp, PL, slack_units;
  int gx, gy, gz;  // logical grid
  float pre_slope;
  const float* mask;
  float mask_slope;
  // the activation + dropout between two dense layers, applied while the tensor is packed (the fp32 result is never stored; the
  // arithmetic and the mask stream of evmi_dropout_fused_f32: element index = the element's index in x):
  //   fuse 1: x -> dropout(silu(x), p_drop)            (the forward of the second layer: its input)
  //   fuse 2: x -> dropout(x, p_drop) * silu'(aux)     (the backward of the first layer: its output gradient; aux = its pre-activation output)
  //   fuse 3: x -> fuse_scale * dropout(x, p_drop)     (the backward of a + s * dropout(dense(h)): the gradient the dense layer sees)
  int fuse;
  float fuse_scale;
  const float* aux;
  float p_drop;
  SeedArg seed;
};
__device__ __forceinline__ void pack_x_block(const PackArgs& p, int bx, int b, int go) {
  const int u = bx * 256 + threadIdx.x;
  if (p.slack_units > 0 && b == p.B - 1 && go == p.gz - 1) {
    uint4* tail = p.xp + (long long)p.gz * p.B * p.Tp;
    for (int i = u; i < p.slack_units; i += p.gx * 256) tail[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  if (u >= p.Tp) return;
  const int g = go / p.octs, o = go - g * p.octs;
  const int t = u - p.PL;
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (t >= 0 && t < p.t_in) {
    const float* src = p.x + ((long long)(g * p.cin_g + o * 8) * p.B + b) * p.t_in + t;
    const long long cs = (long long)p.B * p.t_in;
    const int nch = min(8, p.cin_g - o * 8);
    // the 8 channel rows (and the 8 mask rows) are requested together from clamped rows and masked by a select: a read under
    // `i < nch` is compiled as a branch with a vmcnt(0) drain behind it, 8-16 serial round trips per unit
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = src[min(i, nch - 1) * cs];
    float m[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = 1.f;
    if (p.mask) {
      const float* ms = p.mask + (src - p.x);
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = ms[min(i, nch - 1) * cs];
#pragma unroll
      for (int i = 0; i < 8; ++i) m[i] = m[i] > 0.f ? 1.f : p.mask_slope;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = i < nch ? v[i] : 0.f;
    if (p.pre_slope != 1.f) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * p.pre_slope;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] *= m[i];
    if (p.fuse) {
      const long long i0 = src - p.x;
      float z[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) z[i] = 0.f;
      if (p.fuse == 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) z[i] = p.aux[i0 + min(i, nch - 1) * cs];
      }
      const unsigned long long seed = p.seed.get();
      const float inv_keep = 1.f - p.p_drop;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const bool keep = uniform01(seed, (unsigned long long)(i0 + min(i, nch - 1) * cs)) >= p.p_drop;
        float val = v[i];
        if (p.fuse == 1) val = silu_value(val);
        float d = keep ? val / inv_keep : 0.f;
        if (p.fuse == 2) d = d * silu_grad(z[i]);
        if (p.fuse == 3) d = p.fuse_scale * d;
        v[i] = i < nch ? d : 0.f;
      }
    }
  }
  uint4 out;
  out.x = pk_bf16x2(v[0], v[1]);
  out.y = pk_bf16x2(v[2], v[3]);
  out.z = pk_bf16x2(v[4], v[5]);
  out.w = pk_bf16x2(v[6], v[7]);
  p.xp[((long long)go * p.B + b) * p.Tp + u] = out;
}
// A packed tensor that BOTH the convolution kernels and the weight-gradient kernel read.  For pointwise stride-1 layers (k = 1, no
// padding, one group) nothing reads across an item boundary, so the weight gradient needs no padding between the items -- only rows
// (B * Tp units per channel octet) that end on one of its K steps (64 positions), or its last step would run into the next octet's
// row.  The convolution kernels pack such a layer TIGHT (Tp = t_in: a padded item would cost their column tiles a third window piece:
// measured 76 -> 90 us on the FastSpeech2 feed-forward layers).  So when B * t_in is a multiple of 64 the forward convolution's packed
// input and the input-gradient convolution's packed dy ARE the weight gradient's operands (Tq = t_in): no second pack (`pack2_kernel`:
// 90 launches, 1.6 ms of a FastSpeech2 step).  PK_SHARED_SLACK: units behind the tensor that are kept zero (the weight gradient's
// staging reads a window past the end).
static inline bool pk_shared_shape(int k, int stride, int pad, int dil, int groups) { return k == 1 && stride == 1 && pad == 0 && dil == 1 && groups == 1; }
// A layer called with ONE item (B == 1: a caller whose layer is position-wise hands over all its columns as one row -- the FastSpeech2
// dense layers do) shares for EVERY length: the row pitch of a single item is free, so it is rounded up to the K step and the units
// behind the row's last column are zero (pk_shared_pitch).
static inline bool pk_shared_items(int B, int n) { return B == 1 || ((long long)B * n) % 64 == 0; }
static inline long long pk_shared_pitch(int B, int n) { return B == 1 ? ((long long)n + 63) / 64 * 64 : n; }  // units per item
constexpr int PK_SHARED_SLACK = 384;

static inline PackArgs make_pack_args(const float* x, uint4* xp, int cin_g, int octs, int B, int t_in, int Tp, int PL, int slack_units,
                                      int groups) {
  PackArgs p;
  p.x = x; p.xp = xp; p.cin_g = cin_g; p.octs = octs; p.B = B; p.t_in = t_in; p.Tp = Tp; p.PL = PL; p.slack_units = slack_units;
  p.gx = (Tp + 255) / 256; p.gy = B; p.gz = groups * octs;
  p.pre_slope = 1.f; p.mask = nullptr; p.mask_slope = 1.f;
  p.fuse = 0; p.fuse_scale = 1.f; p.aux = nullptr; p.p_drop = 0.f; p.seed = SeedArg{0ull, nullptr};
  return p;
}
// decode a flat block index into the logical 3-D grid of a pack (x fastest)
__device__ __forceinline__ void pack_x_flat(const PackArgs& p, unsigned bid) {
  const unsigned bx = bid % p.gx, r = bid / p.gx;
  pack_x_block(p, (int)bx, (int)(r % p.gy), (int)(r / p.gy));
}
// two packs in one launch (the weight gradient's dy and x): blocks [0, n0) take the first
static __global__ __launch_bounds__(256) void pack2_kernel(PackArgs p0, PackArgs p1) {
  const unsigned n0 = (unsigned)p0.gx * p0.gy * p0.gz;
  if (blockIdx.x < n0) pack_x_flat(p0, blockIdx.x);
  else pack_x_flat(p1, blockIdx.x - n0);
}

}  // namespace evmi
