// Shared device/host helpers for libevmi_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <string>

#include "evmi.h"

namespace evmi {

// Launch-attribute caches and library-owned scratch are kept PER DEVICE (and per host thread): a process that drives several GPUs
// -- a module moved with .to(), tests over several devices -- must set hipFuncAttributeMaxDynamicSharedMemorySize on each of them
// and must not hand one device's scratch to another.
constexpr int kMaxDevices = 16;
inline int device_slot() {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return 0;
  return d;
}


// ---- error plumbing -----------------------------------------------------------------------
void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define EVMI_HIP_CHECK(expr)                                                                   \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      return ::evmi::fail(EVMI_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));    \
    }                                                                                          \
  } while (0)

#define EVMI_LAUNCH_CHECK(what)                                                                \
  do {                                                                                         \
    hipError_t _e = hipGetLastError();                                                         \
    if (_e != hipSuccess) {                                                                    \
      return ::evmi::fail(EVMI_ERR_HIP, std::string(what) + " launch: " + hipGetErrorString(_e)); \
    }                                                                                          \
  } while (0)

// ---- bf16 vectors ---------------------------------------------------------------------------
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// Accumulator layout <-> 16 contiguous bytes per lane.  After a 32 x 32 MFMA chain lane (n, h) holds, for every register quad q,
// channels 8q + 4h .. + 3 of row n.  With d = (two packed-bf16 dwords of quad 2p | two of quad 2p + 1), one v_permlane32_swap per
// dword pair leaves lane (n, 0) with channels 16p .. 16p + 7 and lane (n, 1) with 16p + 8 .. 16p + 15 of the row.  The exchange is
// its own inverse (also turns a lane's 16 loaded bytes into the two quads' dwords).
__device__ __forceinline__ u32x4 swap_quads_bf16(u32x4 d) {
  const auto r0 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);  // vdst = dword of quad 2p, src = of quad 2p + 1
  const auto r1 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
  u32x4 o;
  o[0] = r0[0];
  o[2] = r0[1];
  o[1] = r1[0];
  o[3] = r1[1];
  return o;
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  bf16x2_t pk;
  pk[0] = (bf16_t)lo;
  pk[1] = (bf16_t)hi;
  return __builtin_bit_cast(unsigned, pk);
}
__device__ __forceinline__ float bf16_lo(unsigned d) { return __builtin_bit_cast(float, d << 16); }
__device__ __forceinline__ float bf16_hi(unsigned d) { return __builtin_bit_cast(float, d & 0xffff0000u); }

// Workgroup barrier for LDS hand-offs only.  __syncthreads() carries a workgroup-scope release fence,
// which on gfx950 waits vmcnt(0) while stores may be outstanding: in a persistent kernel it would drain every global load that is
// deliberately kept in flight across the barrier (next tile's activation rows, next tap group's
// weights) and every epilogue store.  Here only this wave's LDS operations are completed before the
// barrier; global loads are waited for by the compiler exactly where their registers are consumed.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Barrier behind which the tiles requested with LDS-direct loads (global_load_lds: counted by vmcnt) have landed for EVERY wave.
// A plain __syncthreads() does not promise that: its release fence waits for outstanding STORES, and where the compiler sees none
// it emits s_waitcnt lgkmcnt(0) + s_barrier only, nor does it order an LDS-direct load against a later ds_read of the same bytes.
// The bf16 dK/dV attention kernel read its dO tile with no vmcnt wait at all -- a stale tile about once per 60 FastSpeech2 training
// steps, seen as run-to-run differences of otherwise bit-reproducible steps (tests/test_gpu_fs2_train.py: lockstep trainers).
__device__ __forceinline__ void lds_dma_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// One tap group of the implicit GEMM for one wave: NTAPS taps x KSTEPS 16-deep k-steps, fragments
// read from LDS one k-step ahead of the MFMAs that consume them (explicit software pipelining: the
// compiler otherwise issues the ds_reads of step k+1 only after the last MFMA of step k).
//   Arow: this lane's A row (tap 0, m-tile 0) + (lane>>5)*8;  A tap stride / m-tile stride in elements
//   Brow: this lane's B row (tap 0, n-tile 0) + (lane>>5)*8;  b_tap_stride = dilation * row stride
template <int MT, int NT, int KSTEPS, int NTAPS, int A_TAP_STRIDE, int A_MT_STRIDE, int B_NT_STRIDE, int PRIO = 0>
__device__ __forceinline__ void mma_tap_group(const bf16_t* __restrict__ Arow, const bf16_t* __restrict__ Brow,
                                              int b_tap_stride, f32x16 (&acc)[MT][NT]) {
  constexpr int N = NTAPS * KSTEPS;
  bf16x8 af[2][MT], bfr[2][NT];
  auto load = [&](int kk, int buf) {
    const int jj = kk / KSTEPS, ks = kk % KSTEPS;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
      af[buf][mt] = *reinterpret_cast<const bf16x8*>(Arow + jj * A_TAP_STRIDE + mt * A_MT_STRIDE + ks * 16);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      bfr[buf][nt] = *reinterpret_cast<const bf16x8*>(Brow + jj * b_tap_stride + nt * B_NT_STRIDE + ks * 16);
  };
  load(0, 0);
#pragma unroll
  for (int kk = 0; kk < N; ++kk) {
    const int cur = kk & 1;
    if (kk + 1 < N) load(kk + 1, cur ^ 1);
    if (PRIO) __builtin_amdgcn_s_setprio(1);  // favour the wave that has its operands over the one issuing loads
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mt], bfr[cur][nt], acc[mt][nt], 0, 0, 0);
    if (PRIO) __builtin_amdgcn_s_setprio(0);
  }
}

// leaky ReLU with packed fp32 multiplies (v_pk_mul_f32: two values per op) -- the activation passes of the 32 / 64-channel inference
// kernels are VALU-bound.  Same arithmetic as fmaxf(f, f * slope) per element.
__device__ __forceinline__ bf16x4 lrelu4_bf16(f32x4 f, float slope) {
  const f32x4 m = f * slope;
  bf16x4 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) r[e] = (bf16_t)fmaxf(f[e], m[e]);
  return r;
}
__device__ __forceinline__ bf16x8 lrelu8_bf16(bf16x8 x, float slope) {
  f32x4 lo, hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) { lo[e] = (float)x[e]; hi[e] = (float)x[4 + e]; }
  const f32x4 ml = lo * slope, mh = hi * slope;
  bf16x8 r;
#pragma unroll
  for (int e = 0; e < 4; ++e) { r[e] = (bf16_t)fmaxf(lo[e], ml[e]); r[4 + e] = (bf16_t)fmaxf(hi[e], mh[e]); }
  return r;
}

// host-side round-to-nearest-even fp32 -> bf16 bits (NaN kept quiet)
inline uint16_t f32_to_bf16_bits(float f) {
  uint32_t u;
  __builtin_memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// acc + p[0] + p[stride] + ... + p[(n - 1) stride], added in index order (the order is part of the bitwise-reproducibility
// contract of the split reductions) with eight loads in flight: a plain `for (s) acc += p[s * stride]` is compiled as one
// load -> vmcnt(0) -> add per trip, i.e. one memory round trip per term.
__device__ __forceinline__ float ordered_sum_strided(const float* __restrict__ p, long long stride, int n, float acc = 0.f) {
  int s = 0;
  for (; s + 16 <= n; s += 16) {  // (sixteen in flight while there are that many: a 64-way split is four round trips, not eight)
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = p[(long long)(s + u) * stride];
#pragma unroll
    for (int u = 0; u < 16; ++u) acc += v[u];
  }
  for (; s < n; s += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(long long)min(s + u, n - 1) * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = s + u < n ? acc + v[u] : acc;
  }
  return acc;
}


// A dropout seed as the kernels receive it: `value` by value plus an optional device-resident base.  With a base the effective
// seed is (value + *base) & (2^63 - 1): a step captured into a HIP graph keeps `value` (the draw's index inside the step) and gets
// a new mask every replay because the host rewrites *base (seed, step, rank) before it -- and an eager step that stores the same
// base draws bit for bit the same masks.  base == NULL: the seed is `value`, as before.
// silu'(z) = s (1 + z (1 - s)), s = sigmoid(z): ONE statement of the arithmetic, without contraction, for every kernel that applies it
// (the separate dropout / activation passes and the packs that fuse them round the same values to bf16)
__device__ __forceinline__ float silu_grad(float z) {
#pragma clang fp contract(off)
  const float sg = 1.f / (1.f + expf(-z));
  return sg * (1.f + z * (1.f - sg));
}
__device__ __forceinline__ float silu_value(float v) {
#pragma clang fp contract(off)
  return v / (1.f + expf(-v));
}

struct SeedArg {
  unsigned long long value;
  const unsigned long long* base;
  __device__ __forceinline__ unsigned long long get() const { return base ? ((value + *base) & 0x7FFFFFFFFFFFFFFFull) : value; }
};

// Counter-based uniform in [0, 1) for dropout masks: element i of the stream `seed`; the backward (and the attention kernels,
// tile by tile) regenerate the same mask from the same (seed, i).  A keyed two-round 32-bit mix (the multiply / xor-shift rounds of
// the `lowbias32` integer hash, the second key injected between the rounds so that streams are not shifted or permuted copies of
// each other): 3 quarter-rate 32-bit multiplies per hash against the 12 of the splitmix64 it replaces.  The seed-only part is
// wave-uniform (scalar unit).
// ONE hash serves TWO elements: elements 2 j and 2 j + 1 take the low and the high 16 bits of hash(seed, j) (a draw has 16 bits: the
// keep probability is 1 - ceil(65536 p) / 65536, within 1.5e-5 of 1 - p).  Where the generator, not the memory system, bounds a
// kernel -- the attention kernels draw one mask bit per probability, the matrix kernels' dropout epilogues one per output -- a
// thread that holds both elements of a pair hashes once (dropout_pair), and two lanes that hold one each hash half of their
// registers and exchange (dropout_lane_pairs); every other caller asks per element (uniform01): all three give the same bits.
__device__ __forceinline__ unsigned dropout_hash(unsigned long long seed, unsigned j_lo, unsigned j_hi) {
  const unsigned k0 = (unsigned)seed * 0x9E3779B1u + 0x7F4A7C15u;
  unsigned k1 = (unsigned)(seed >> 32) ^ (k0 >> 15);
  k1 = k1 * 0x85EBCA6Bu + 0xC2B2AE35u;
  unsigned h = j_lo ^ k0;
  h += j_hi * 0x27D4EB2Fu;  // pair indices past 2^32 (not reached by these models) still get their own draws
  h ^= h >> 16;
  h *= 0x7FEB352Du;
  h ^= k1;
  h ^= h >> 15;
  h *= 0x846CA68Bu;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float dropout_u16(unsigned h, unsigned odd) { return (float)(odd ? h >> 16 : h & 0xFFFFu) * (1.0f / 65536.0f); }
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long i) {
  const unsigned long long j = i >> 1;
  return dropout_u16(dropout_hash(seed, (unsigned)j, (unsigned)(j >> 32)), (unsigned)i & 1u);
}
// elements 2 j and 2 j + 1 (j < 2^32: the caller's host side checks the tensor's size)
__device__ __forceinline__ void dropout_pair(unsigned long long seed, unsigned j, float& u_even, float& u_odd) {
  const unsigned h = dropout_hash(seed, j, 0u);
  u_even = dropout_u16(h, 0u);
  u_odd = dropout_u16(h, 1u);
}


// Sum over the 64 lanes of a wave, the same value returned in every lane: four DPP adds inside the rows of 16 lanes (quad swaps, half
// mirror, mirror), then the four row sums through the scalar registers.  (`__shfl_xor` compiles to ds_bpermute_b32 -- an LDS-crossbar
// round trip with its own s_waitcnt per step: the LayerNorm backward's 64 reductions per thread were 384 of them in a dependent
// chain, a ~10 us floor under a kernel that moves 5-27 MB.)
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1, 0, 3, 2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2, 3, 0, 1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return (r0 + r1) + (r2 + r3);
}

// ... and in double precision (the two halves of the value move together)
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xFFFFFFFFll), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}
__device__ __forceinline__ double wave_sum_dpp_f64(double v) {
  v += dpp_move_f64<0xB1>(v);
  v += dpp_move_f64<0x4E>(v);
  v += dpp_move_f64<0x141>(v);
  v += dpp_move_f64<0x140>(v);
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = (int)(b & 0xFFFFFFFFll), hi = (int)(b >> 32);
  double r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int l = __builtin_amdgcn_readlane(lo, 16 * q), h = __builtin_amdgcn_readlane(hi, 16 * q);
    r[q] = __builtin_bit_cast(double, ((long long)h << 32) | (long long)(unsigned)l);
  }
  return (r[0] + r[1]) + (r[2] + r[3]);
}

// wait until at most n (wave-uniform) vector-memory operations of this wave are outstanding
__device__ __forceinline__ void wait_vmcnt_le(int n) {
  n = __builtin_amdgcn_readfirstlane(n);
#define EVMI_VMCASE(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    EVMI_VMCASE(0) EVMI_VMCASE(1) EVMI_VMCASE(2) EVMI_VMCASE(3) EVMI_VMCASE(4) EVMI_VMCASE(5) EVMI_VMCASE(6)
    EVMI_VMCASE(7) EVMI_VMCASE(8) EVMI_VMCASE(9) EVMI_VMCASE(10) EVMI_VMCASE(11) EVMI_VMCASE(12) EVMI_VMCASE(13)
    EVMI_VMCASE(14) EVMI_VMCASE(15) EVMI_VMCASE(16) EVMI_VMCASE(17) EVMI_VMCASE(18) EVMI_VMCASE(19) EVMI_VMCASE(20)
    EVMI_VMCASE(21) EVMI_VMCASE(22) EVMI_VMCASE(23) EVMI_VMCASE(24) EVMI_VMCASE(25) EVMI_VMCASE(26) EVMI_VMCASE(27)
    EVMI_VMCASE(28) EVMI_VMCASE(29) EVMI_VMCASE(30) EVMI_VMCASE(31) EVMI_VMCASE(32) EVMI_VMCASE(33) EVMI_VMCASE(34)
    EVMI_VMCASE(35) EVMI_VMCASE(36) EVMI_VMCASE(37) EVMI_VMCASE(38) EVMI_VMCASE(39) EVMI_VMCASE(40) EVMI_VMCASE(41)
    EVMI_VMCASE(42) EVMI_VMCASE(43) EVMI_VMCASE(44) EVMI_VMCASE(45) EVMI_VMCASE(46) EVMI_VMCASE(47) EVMI_VMCASE(48)
    EVMI_VMCASE(49) EVMI_VMCASE(50) EVMI_VMCASE(51) EVMI_VMCASE(52) EVMI_VMCASE(53) EVMI_VMCASE(54) EVMI_VMCASE(55)
    EVMI_VMCASE(56) EVMI_VMCASE(57) EVMI_VMCASE(58) EVMI_VMCASE(59) EVMI_VMCASE(60) EVMI_VMCASE(61) EVMI_VMCASE(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef EVMI_VMCASE
}


// attention_train.hip: bf16 flash-style forward without dropout (lse_or_null: the training path's saved log-sum-exp, NULL for
// inference); returns nonzero for an unsupported head dimension (32 / 64 / 128 are built)
int launch_mha_fwd_bf16_plain(const float* qkv, const int* lens, float* out, float* lse_or_null, int B, int T, int D, int heads,
                              hipStream_t s);

}  // namespace evmi
