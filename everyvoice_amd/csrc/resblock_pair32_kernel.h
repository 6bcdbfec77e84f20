// Fused residual pair for the 32-channel stage, as a two-stage pipeline inside one persistent workgroup:
//
//   out = post( [accumulate ? out : 0] + out_scale * ( x + b2 + conv2( lrelu( b1 + conv1_dil( lrelu(x) ) ) ) ) )
//
// Same contract and arguments as resblock_pair_kernel.h.  What differs:
//   * waves 0-3 compute conv1 of tile k while waves 4-7 compute conv2 of tile k - 1 (one s_barrier per tile), so a SIMD always
//     has one wave of each convolution: the epilogue / activation work of one team runs under the other team's MFMAs;
//   * every wave keeps ITS convolution's weights in registers for the life of the workgroup (A fragments: KS x 2 x 4 VGPRs,
//     88 at KS = 11) -- at 32 channels the weight fragments were half of all LDS reads (one A + two B reads per two MFMAs);
//     the LDS now serves activation fragments only, one ds_read_b128 per MFMA;
//   * the raw rows of tile k + 2 arrive by LDS-DMA (global_load_lds_dwordx4) while tiles k + 1 and k are being computed: no
//     staging registers, no commit pass; rows are 64 B = four 16-byte slots, slot p of row r holds channel vector
//     p ^ ((r >> 2) & 3) (the 16-lane groups of ds_read_b128 then hit 16 distinct slots); rows outside the sequence read a
//     zero line;
//   * epilogues straight from registers (v_permlane32_swap -> 16-byte stores; residual from the raw tile in LDS).
//
// LDS (KS = 11): raw tiles XR[4] (tile k is read by the activation pass in phase k - 1 and by conv2's residual in phase k + 1),
// activated tiles XA[2], intermediates T1[2]: 4 x 20 + 2 x 20 + 2 x 17 KB = 154 KB.
//
// Measured (MI355X, forward of the bench): k = 11 1.085 vs 1.133 ms per three launches, k = 7 0.989 vs 0.94, whole forward 17.82 vs
// 17.63 ms -- the per-tile fixed work (activation pass, two epilogues, one barrier) of 256-row tiles outweighs the saved weight
// fragments at the short kernels, so the single-team 512-row kernels stay the default (EVMI_PAIR32=1 selects this one).
//
// Phase p (between two barriers), tiles counted along the workgroup's own walk:
//   all waves     : request the raw rows of tile p + 2                       -> XR[(p + 2) % 4]
//   waves 0-3     : T1[p & 1] = lrelu(b1 + conv1(XA[p & 1])) ; XA[(p + 1) & 1] = lrelu(XR[(p + 1) % 4])
//   waves 4-7     : out(tile p - 1) = conv2(T1[(p - 1) & 1]) + b2 + XR[(p - 1) % 4] rows ...
#pragma once

#include "resblock_pair_kernel.h"

namespace evmi {

static __device__ __attribute__((aligned(128))) bf16_t g_pair32_zero_row[32];  // zero-initialised: source of out-of-range rows

template <int KS_, int MAXDIL_>
struct Pair32Cfg {
  static constexpr int C = 32, KS = KS_, MAXDIL = MAXDIL_, BN = 256, TT = BN - (KS - 1);
  static constexpr int NTHREADS = 512, NT = 2;  // 4 + 4 waves, 64 rows per wave
  static constexpr int RA_MAX = BN + (KS - 1) * MAXDIL;
  static constexpr int XR_PIECES = (RA_MAX + 15) / 16;  // 16 rows x 64 B = 1 KiB = one wave-instruction
  static constexpr int XR_BYTES = XR_PIECES * 1024;
  static constexpr int T1_ROWS = BN + KS - 1;
  static constexpr int T1_BYTES = (T1_ROWS + 15) / 16 * 1024;
  static constexpr int OFF_XA = 4 * XR_BYTES, OFF_T1 = OFF_XA + 2 * XR_BYTES;
  static constexpr size_t LDS = size_t(OFF_T1) + 2 * T1_BYTES;
  static constexpr int A_VEC_PER_THREAD = (XR_PIECES * 64 + 255) / 256;  // activation pass: 256 threads (the conv1 team)
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <class P>
__global__ __launch_bounds__(P::NTHREADS, 2) void resblock_pair32_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  constexpr int C = 32, KS = P::KS, H2 = (KS - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int team = wave >> 2, wl = wave & 3;  // team 0: conv1, team 1: conv2; wl: which 64 rows
  const int h = lane >> 5, ln = lane & 31;
  const int h1 = a.dil1 * (KS - 1) / 2;
  const int ra = P::BN + (KS - 1) * a.dil1;   // raw rows a tile needs
  const int xr_pieces = (ra + 15) >> 4;

  // XCD-aware tile walk (as resblock_pair_kernel): each XCD takes one contiguous range of tiles
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_wg = (nwg + 7) >> 3;
  const int tiles_per_xcd = (a.n_tiles + 7) >> 3;
  const int tile_lo = xcd * tiles_per_xcd;
  const int tile_hi = min(a.n_tiles, tile_lo + tiles_per_xcd);
  const int first = tile_lo + slot;
  if (first >= tile_hi) return;
  const int nt_wg = (tile_hi - first + per_xcd_wg - 1) / per_xcd_wg;  // tiles of this workgroup
  auto tile_of = [&](int k) { return first + k * per_xcd_wg; };

  // ---- this wave's weights (A fragments) and bias, in registers for the life of the workgroup ----------------------
  // w [tap][32 co][32 ci] bf16: lane (co = ln, half h) takes ci = 16 ks + 8 h .. + 7 of tap j
  const bf16_t* wsrc = team ? a.w2 : a.w1;
  bf16x8 wreg[KS][2];
#pragma unroll
  for (int j = 0; j < KS; ++j)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wreg[j][ks] = *reinterpret_cast<const bf16x8*>(wsrc + ((long long)j * C + ln) * C + 16 * ks + 8 * h);
  float bias_r[16];  // accumulator layout: channel 8q + 4h + i
  {
    const float* bsrc = team ? a.b2 : a.b1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(bsrc + 8 * q + 4 * h);
#pragma unroll
      for (int i = 0; i < 4; ++i) bias_r[4 * q + i] = bv[i];
    }
  }

  // ---- LDS-DMA request of the raw rows of the workgroup's k-th tile -------------------------------------------------
  auto issue_rows = [&](int k) {
    const int tile = tile_of(k);
    const int item = tile / a.tiles_per_item, rt = tile - item * a.tiles_per_item;
    const int g0 = rt * P::TT - H2 - h1;  // global row of tile row 0
    const bf16_t* xb = a.x + (long long)item * a.T * C;
    char* dst = smem + (k & 3) * P::XR_BYTES;
    const int sl = lane & 3;
    for (int p = wl; p < xr_pieces; p += 4) {  // (called by the conv1 team only: its four waves share the pieces)
      const int row = p * 16 + (lane >> 2);
      const int g = g0 + row;
      const int c4 = sl ^ ((row >> 2) & 3);
      const bf16_t* src = (row < ra && g >= 0 && g < a.T) ? xb + (long long)g * C + c4 * 8 : g_pair32_zero_row + sl * 8;
      __builtin_amdgcn_global_load_lds((__attribute__((address_space(1))) const void*)src,
                                       (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
  };
  // XA[k & 1] = lrelu(XR[k & 3]) by `nthr` threads (thread index t)
  auto activate = [&](int k, int t, int nthr) {
    const char* src = smem + (k & 3) * P::XR_BYTES;
    char* dst = smem + P::OFF_XA + (k & 1) * P::XR_BYTES;
    const float sl = a.slope;
    for (int v = t; v < xr_pieces * 64; v += nthr) {
      bf16x8 val = *reinterpret_cast<const bf16x8*>(src + v * 16);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float f = (float)val[e];
        val[e] = (bf16_t)fmaxf(f, f * sl);
      }
      *reinterpret_cast<bf16x8*>(dst + v * 16) = val;
    }
  };
  // one convolution of this wave's 64 rows: B fragments from `tile` (64-byte swizzled rows), taps `tap_stride` rows apart.
  // The KS x 2 x NT fragment reads run PF steps ahead of the MFMAs that consume them, through a ring of PF fragment registers
  // (left to itself the compiler reuses ONE fragment register: read, wait for it, one MFMA, read ... -- a full LDS round trip
  // per MFMA); the order is pinned with sched_group_barrier: one MFMA, one LDS read, ...
  auto conv = [&](const char* tile, int tap_stride, f32x16 (&acc)[P::NT]) {
#pragma unroll
    for (int nt = 0; nt < P::NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[nt][r] = bias_r[r];
    constexpr int NF = KS * 2 * P::NT, PF = 4;
    auto frag = [&](int f) -> bf16x8 {  // f = (j * 2 + ks) * NT + nt
      const int nt = f % P::NT, ks = (f / P::NT) % 2, j = f / (2 * P::NT);
      const int q = wl * 64 + ln + j * tap_stride;  // this lane's row (n-tile 0); + 32 rows leaves the swizzle term unchanged
      return *reinterpret_cast<const bf16x8*>(tile + q * 64 + (((2 * ks + h) ^ ((q >> 2) & 3)) << 4) + nt * (32 * 64));
    };
    bf16x8 ring[PF];
#pragma unroll
    for (int f = 0; f < PF; ++f) ring[f] = frag(f);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      const int nt = f % P::NT, ks = (f / P::NT) % 2, j = f / (2 * P::NT);
      acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[j][ks], ring[f % PF], acc[nt], 0, 0, 0);
      if (f + PF < NF) ring[f % PF] = frag(f + PF);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
    }
  };

  // ---- prologue: rows of tiles 0 and 1 requested, T1 tails zeroed, tile 0 activated -------------------------------------
  // Only the conv1 team requests rows, and only it waits for them (s_waitcnt vmcnt(0) in front of the phase barrier); the
  // conv2 team passes the barriers with its LDS operations completed but its output STORES still in flight -- a
  // __syncthreads() would drain them at every phase.
  auto phase_barrier = [&]() {
    if (team == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
  };
  if (team == 0) {
    issue_rows(0);
    if (nt_wg > 1) issue_rows(1);
  }
  for (int v = tid; v < 2 * (P::T1_BYTES / 16); v += P::NTHREADS) {  // rows BN .. of T1 only feed discarded outputs: keep them finite
    u32x4 z = {0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(smem + P::OFF_T1 + v * 16) = z;
  }
  phase_barrier();  // both tiles have landed
  activate(0, tid, P::NTHREADS);

  for (int p = 0; p <= nt_wg; ++p) {
    phase_barrier();  // requests issued a phase ago have landed; last phase's LDS writes are visible; its reads are done
    if (team == 0) {
      if (p + 2 < nt_wg) issue_rows(p + 2);
      if (p < nt_wg) {
        // ---- conv1 of tile p: T1[n] = lrelu(b1 + conv1) for global row r0 - H2 + n, zero outside the sequence --------
        const int tile = tile_of(p);
        const int rt = tile % a.tiles_per_item;
        const int r0 = rt * P::TT;
        f32x16 acc[P::NT];
        conv(smem + P::OFF_XA + (p & 1) * P::XR_BYTES, a.dil1, acc);
        char* T1 = smem + P::OFF_T1 + (p & 1) * P::T1_BYTES;
        const float sl = a.slope;
#pragma unroll
        for (int nt = 0; nt < P::NT; ++nt) {
          const int n = wl * 64 + nt * 32 + ln;
          const int g = r0 - H2 + n;
          const float mask = (g >= 0 && g < a.T) ? 1.f : 0.f;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            bf16x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float f = acc[nt][4 * q + i];
              pk[i] = (bf16_t)(fmaxf(f, f * sl) * mask);
            }
            *reinterpret_cast<bf16x4*>(T1 + n * 64 + ((q ^ ((n >> 2) & 3)) << 4) + 8 * h) = pk;
          }
        }
      }
      if (p + 1 < nt_wg) activate(p + 1, tid, 256);  // (threads 0 .. 255 are the conv1 team; on the conv2 team it measured slower)
    } else if (p >= 1) {
      // ---- conv2 of tile p - 1, + residual (raw tile in LDS, accumulator layout) -> registers -> 16-byte stores --------
      const int k = p - 1;
      const int tile = tile_of(k);
      const int item = tile / a.tiles_per_item, rt = tile - item * a.tiles_per_item;
      const int r0 = rt * P::TT;
      f32x16 acc[P::NT];
      conv(smem + P::OFF_T1 + (k & 1) * P::T1_BYTES, 1, acc);
      const char* XR = smem + (k & 3) * P::XR_BYTES;
      bf16_t* ob = a.out + (long long)item * a.T * C;
      const float scale = a.out_scale, post = a.post_slope;
#pragma unroll
      for (int nt = 0; nt < P::NT; ++nt) {
        const int n = wl * 64 + nt * 32 + ln;
        const int r = r0 + n;
        const bool ok = n < P::TT && r < a.T;
        bf16_t* dst = ob + (long long)(ok ? r : 0) * C + 8 * h;
        const int xrow = n + H2 + h1;  // the raw tile's row of output row n
        u32x4 pv[2];
        if (a.accumulate) {  // wave-uniform; both vectors of the lane are requested before the first is consumed
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) pv[p2] = *reinterpret_cast<const u32x4*>(dst + 16 * p2);
        }
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2) {
          float f[8];
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const int q = 2 * p2 + qq;
            const bf16x4 rv = *reinterpret_cast<const bf16x4*>(XR + xrow * 64 + ((q ^ ((xrow >> 2) & 3)) << 4) + 8 * h);
#pragma unroll
            for (int i = 0; i < 4; ++i) f[4 * qq + i] = (acc[nt][4 * q + i] + (float)rv[i]) * scale;
          }
          if (a.accumulate) {
            const u32x4 d = swap_quads_bf16(pv[p2]);
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              f[2 * w] += bf16_lo(d[w]);
              f[2 * w + 1] += bf16_hi(d[w]);
            }
          }
          u32x4 o;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const float lo = post != 1.f ? fmaxf(f[2 * w], f[2 * w] * post) : f[2 * w];
            const float hi = post != 1.f ? fmaxf(f[2 * w + 1], f[2 * w + 1] * post) : f[2 * w + 1];
            o[w] = pack_bf16x2(lo, hi);
          }
          o = swap_quads_bf16(o);
          if (ok) *reinterpret_cast<u32x4*>(dst + 16 * p2) = o;
        }
      }
    }
  }
}

template <class P>
static PairLaunch make_pair32_launch(const char* name) {
  PairLaunch l;
  l.kernel = resblock_pair32_kernel<P>;
  l.c = P::C;
  l.ks = P::KS;
  l.bn = P::BN;
  l.tt = P::TT;
  l.threads = P::NTHREADS;
  l.max_dil = P::MAXDIL;
  l.lds_bytes = P::LDS;
  l.name = name;
  l.kc = P::C;
  l.wg_per_cu = 1;
  return l;
}

}  // namespace evmi
