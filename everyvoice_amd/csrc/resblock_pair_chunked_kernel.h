// Fused residual pair for wide layers (C = 128): same contract as resblock_pair_kernel.h, but the
// conv1 operand tile holds one 64-channel chunk at a time (the whole-C tile does not fit next to the
// intermediate), the intermediate T1 holds all C channels, and the waves tile the output 2 (channels) x
// 4 (rows) with 64 x 64 per wave.  Per tile: 2*NCH*KS weight steps, one activation load per chunk, one
// epilogue — twice the K depth per tile of the unfused convolutions, and no T1 round trip through HBM.
#pragma once

#include "resblock_pair_kernel.h"

namespace evmi {

template <int C_, int KC_, int KS_, int BN_, int MAXDIL_, int WM_, int WN_>
struct PairChunkedCfg {
  static constexpr int C = C_, KC = KC_, KS = KS_, BN = BN_, MAXDIL = MAXDIL_, WM = WM_, WN = WN_;
  static constexpr int NTHREADS = WM * WN * 64;
  static constexpr int NCH = C / KC;
  static constexpr int MT = C / (WM * 32), NT = BN / (WN * 32);
  static constexpr int SX = KC + 8, ST = C + 8, SW = KC + 8;
  static constexpr int TT = BN - (KS - 1);
  static constexpr int RA_MAX = BN + (KS - 1) * MAXDIL;
  static constexpr int T1_ROWS = BN + KS - 1;
  static constexpr int STEPS_PER_CONV = NCH * KS;
  static constexpr int NSTEP = 2 * STEPS_PER_CONV;
  static constexpr int W_TILE = C * SW;
  static constexpr int W_VECS = C * (KC / 8);
  static constexpr int W_PER_THREAD = W_VECS / NTHREADS;
  static constexpr int XV = (RA_MAX * (KC / 8) + NTHREADS - 1) / NTHREADS;
  static constexpr int RV = (BN * (C / 8)) / NTHREADS;  // epilogue vectors per thread (over all BN rows)
  static constexpr size_t OFF_XA = 0;
  static constexpr size_t OFF_T1 = OFF_XA + size_t(RA_MAX) * SX;
  static constexpr size_t OFF_WS = OFF_T1 + size_t(T1_ROWS) * ST;
  static constexpr size_t LDS = (OFF_WS + 2 * size_t(W_TILE)) * 2;
  static_assert(W_VECS % NTHREADS == 0 && (BN * (C / 8)) % NTHREADS == 0, "even split over the threads");
  static_assert(C % (WM * 32) == 0 && BN % (WN * 32) == 0 && C % KC == 0, "tiling");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <class P>
// (one workgroup per CU by its LDS: the register budget is the full 256 per wave at two waves per SIMD)
__global__ __launch_bounds__(P::NTHREADS) void resblock_pair_chunked_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* XA = reinterpret_cast<bf16_t*>(smem) + P::OFF_XA;
  bf16_t* T1 = reinterpret_cast<bf16_t*>(smem) + P::OFF_T1;
  bf16_t* WS = reinterpret_cast<bf16_t*>(smem) + P::OFF_WS;
  bf16_t* OS = T1;  // epilogue staging [BN][ST] reuses the intermediate once conv2 has consumed it

  constexpr int C = P::C, KC = P::KC, KS = P::KS, H2 = (KS - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / P::WN, wn = wave % P::WN;
  const int h1 = a.dil1 * (KS - 1) / 2;
  const int ra = P::BN + (KS - 1) * a.dil1;
  const int x_nvec = ra * (KC / 8);

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_wg = (gridDim.x + 7) >> 3;
  const int tiles_per_xcd = (a.n_tiles + 7) >> 3;
  const int tile_lo = xcd * tiles_per_xcd;
  const int tile_hi = min(a.n_tiles, tile_lo + tiles_per_xcd);
  int tile = tile_lo + slot;
  if (tile >= tile_hi) return;

  bf16x8 xreg[P::XV];
  // weight images are requested TWO steps ahead (a ring of three register sets): a step is 16 MFMAs per wave, far shorter than the
  // L2 round trip that a one-step-ahead prefetch had to cover
  constexpr int WRING = 3;
  bf16x8 wreg[WRING][P::W_PER_THREAD];
  bf16x8 rreg[P::RV];

  auto x_issue = [&](int t, int chunk) {
    const int item = t / a.tiles_per_item, rt = t % a.tiles_per_item;
    const int g0 = rt * P::TT - H2 - h1;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x + (long long)item * a.T * C), 0,
                                                        a.T * C * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < P::XV; ++i) {
      const int v = tid + i * P::NTHREADS;
      const int row = v / (KC / 8), c8 = v % (KC / 8);
      const int g = g0 + row;
      // rows outside [0, T) fall outside the descriptor and read as zero (the convolution's zero padding)
      const unsigned voff = g < 0 ? 0xfffffff0u : (unsigned)((g * C + chunk * KC + c8 * 8) * 2);
      xreg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, 0, 0));
    }
  };
  auto x_commit = [&]() {
    const float sl = a.slope;
#pragma unroll
    for (int i = 0; i < P::XV; ++i) {
      const int v = tid + i * P::NTHREADS;
      if (v < x_nvec) {
        const int row = v / (KC / 8), c8 = v % (KC / 8);
        bf16x8 val = xreg[i];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)val[e];
          val[e] = (bf16_t)fmaxf(f, f * sl);
        }
        *reinterpret_cast<bf16x8*>(XA + row * P::SX + c8 * 8) = val;
      }
    }
  };
  // step s: conv = s / STEPS_PER_CONV; within a conv: chunk-major, tap-minor.  Global layout per conv:
  // [chunk][tap][C][KC] (the conv_tc layout with BM = C)
  auto w_prefetch = [&](int s, int slot) {
    const int conv = s / P::STEPS_PER_CONV, rem = s % P::STEPS_PER_CONV;
    const bf16_t* src = (conv ? a.w2 : a.w1) + (long long)rem * C * KC;
#pragma unroll
    for (int i = 0; i < P::W_PER_THREAD; ++i)
      wreg[slot][i] = *reinterpret_cast<const bf16x8*>(src + (long long)(tid + i * P::NTHREADS) * 8);
  };
  auto w_commit = [&](int s, int slot) {
    bf16_t* dst = WS + (s & 1) * P::W_TILE;
#pragma unroll
    for (int i = 0; i < P::W_PER_THREAD; ++i) {
      const int v = tid + i * P::NTHREADS;
      *reinterpret_cast<bf16x8*>(dst + (v / (KC / 8)) * P::SW + (v % (KC / 8)) * 8) = wreg[slot][i];
    }
  };
  auto out_offset = [&](int r0, int i) -> unsigned {
    const int v = tid + i * P::NTHREADS;
    const int n = v / (C / 8), c8 = v % (C / 8);
    const int r = r0 + n;
    return (n < P::TT && r < a.T) ? (unsigned)((r * C + c8 * 8) * 2) : 0xfffffff0u;
  };

  x_issue(tile, 0);
#pragma unroll
  for (int s = 0; s < WRING - 1; ++s) w_prefetch(s, s);
  static_assert(P::NSTEP % WRING == 0, "the ring position of a step is the same in every tile");

  for (; tile < tile_hi; tile += per_xcd_wg) {
    const int item = tile / a.tiles_per_item, rt = tile % a.tiles_per_item;
    const int r0 = rt * P::TT;
    const int next = tile + per_xcd_wg < tile_hi ? tile + per_xcd_wg : tile;  // last tile: a harmless repeat

    f32x16 acc[P::MT][P::NT];
    auto zero_acc = [&]() {
#pragma unroll
      for (int i = 0; i < P::MT; ++i)
#pragma unroll
        for (int j = 0; j < P::NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    };

    // ---------------- conv1 (dilated) over the activated residual stream ----------------
    zero_acc();
    static_assert(KS % WRING == 0, "the ring slot of a step is its tap: static under the unrolled tap loop");
#pragma unroll 1
    for (int chunk = 0; chunk < P::NCH; ++chunk) {
      if (chunk > 0) lds_barrier();  // done reading the previous chunk's rows (chunk 0: barrier at the loop end)
      x_commit();
#pragma unroll
      for (int tap = 0; tap < KS; ++tap) {
        const int s = chunk * KS + tap;
        w_commit(s, tap % WRING);
        lds_barrier();
        w_prefetch((s + WRING - 1) % P::NSTEP, (tap + WRING - 1) % WRING);
        if (tap == 0) {
          if (chunk + 1 < P::NCH) x_issue(tile, chunk + 1);
          else x_issue(next, 0);
        }
        const bf16_t* Arow = WS + (s & 1) * P::W_TILE + (wm * P::MT * 32 + (lane & 31)) * P::SW + (lane >> 5) * 8;
        const bf16_t* Brow = XA + (wn * P::NT * 32 + (lane & 31) + tap * a.dil1) * P::SX + (lane >> 5) * 8;
        mma_tap_group<P::MT, P::NT, KC / 16, 1, 0, 32 * P::SW, 32 * P::SX>(Arow, Brow, 0, acc);
      }
    }
    {  // T1[n][c] = lrelu(conv1 + b1) for global row r0 - H2 + n, zero outside the sequence
      const float sl = a.slope;
#pragma unroll
      for (int mt = 0; mt < P::MT; ++mt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = wm * P::MT * 32 + mt * 32 + 8 * q + 4 * (lane >> 5);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b1 + c);
#pragma unroll
          for (int nt = 0; nt < P::NT; ++nt) {
            const int n = wn * P::NT * 32 + nt * 32 + (lane & 31);
            const int g = r0 - H2 + n;
            const float mask = (g >= 0 && g < a.T) ? 1.f : 0.f;
            bf16x4 pk;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float f = acc[mt][nt][4 * q + i] + bv[i];
              pk[i] = (bf16_t)(fmaxf(f, f * sl) * mask);
            }
            *reinterpret_cast<bf16x4*>(T1 + n * P::ST + c) = pk;
          }
        }
      for (int v = tid; v < (KS - 1) * (C / 8); v += P::NTHREADS) {  // rows that only feed discarded outputs
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
        *reinterpret_cast<bf16x8*>(T1 + (P::BN + v / (C / 8)) * P::ST + (v % (C / 8)) * 8) = z;
      }
    }
    // ---------------- conv2 (dense) over the intermediate in LDS ----------------
    zero_acc();
#pragma unroll 1
    for (int chunk = 0; chunk < P::NCH; ++chunk) {
#pragma unroll
      for (int tap = 0; tap < KS; ++tap) {
        const int s = P::STEPS_PER_CONV + chunk * KS + tap;
        w_commit(s, tap % WRING);
        lds_barrier();
        w_prefetch((s + WRING - 1) % P::NSTEP, (tap + WRING - 1) % WRING);  // (wraps to the next tile's first images)
        const bf16_t* Arow = WS + (s & 1) * P::W_TILE + (wm * P::MT * 32 + (lane & 31)) * P::SW + (lane >> 5) * 8;
        const bf16_t* Brow = T1 + (wn * P::NT * 32 + (lane & 31) + tap) * P::ST + chunk * KC + (lane >> 5) * 8;
        mma_tap_group<P::MT, P::NT, KC / 16, 1, 0, 32 * P::SW, 32 * P::ST>(Arow, Brow, 0, acc);
      }
    }
    // ---------------- epilogue ----------------
    const auto xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.x + (long long)item * a.T * C), 0,
                                                         a.T * C * 2, 0x00020000);
    const auto orsrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (long long)item * a.T * C, 0, a.T * C * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < P::RV; ++i)  // residual rows (raw x): L2-resident, issued before the staging pass
      rreg[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, out_offset(r0, i), 0, 0));
    lds_barrier();  // every wave is done reading T1: it becomes the staging tile
#pragma unroll
    for (int mt = 0; mt < P::MT; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = wm * P::MT * 32 + mt * 32 + 8 * q + 4 * (lane >> 5);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(a.b2 + c);
#pragma unroll
        for (int nt = 0; nt < P::NT; ++nt) {
          const int n = wn * P::NT * 32 + nt * 32 + (lane & 31);
          bf16x4 pk;
#pragma unroll
          for (int i = 0; i < 4; ++i) pk[i] = (bf16_t)(acc[mt][nt][4 * q + i] + bv[i]);
          *reinterpret_cast<bf16x4*>(OS + n * P::ST + c) = pk;
        }
      }
    lds_barrier();
    {
      const float scale = a.out_scale, post = a.post_slope;
#pragma unroll
      for (int i = 0; i < P::RV; ++i) {
        const int v = tid + i * P::NTHREADS;
        const unsigned off = out_offset(r0, i);
        const bf16x8 o = *reinterpret_cast<const bf16x8*>(OS + (v / (C / 8)) * P::ST + (v % (C / 8)) * 8);
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = ((float)o[e] + (float)rreg[i][e]) * scale;
        if (a.accumulate) {
          const bf16x8 pv = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(orsrc, off, 0, 0));
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] += (float)pv[e];
        }
        bf16x8 res;
#pragma unroll
        for (int e = 0; e < 8; ++e) res[e] = (bf16_t)fmaxf(f[e], f[e] * post);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, res), orsrc, off, 0, 0);
      }
    }
    lds_barrier();  // staging consumed: the next tile may overwrite T1 / XA
  }
}

template <class P>
static PairLaunch make_pair_chunked_launch(const char* name) {
  PairLaunch l;
  l.kernel = resblock_pair_chunked_kernel<P>;
  l.c = P::C;
  l.ks = P::KS;
  l.bn = P::BN;
  l.tt = P::TT;
  l.threads = P::NTHREADS;
  l.max_dil = P::MAXDIL;
  l.lds_bytes = P::LDS;
  l.name = name;
  l.kc = P::KC;
  return l;
}

}  // namespace evmi
