// One whole branch of a HiFiGAN multi-receptive-field stage -- up to three residual pairs of ResBlock1, dilations (d0, d1, d2) -- on
// one row tile, the running value of the branch kept in LDS between the pairs:
//
//   y0 = x;   y(p+1) = y(p) + b2p + conv2p( lrelu( b1p + conv1p_dil(dp)( lrelu(y(p)) ) ) );   out = post([out +] scale * y(np))
//
// The pair kernel (resblock_pair_kernel.h) reads and writes the residual stream once per pair: three round trips of a 403 MB tensor per
// branch, and the k = 3 / k = 7 pairs of the 32- and 64-channel stages are bound by exactly that (0.8 GB in 216 us).  Here HBM sees the
// branch input once and the branch output once.  Arithmetic and rounding points are those of the pair kernel (y(p) is rounded to bf16
// where the pair kernel stores it): the two paths give the same bits.
//
// Geometry.  All LDS tiles share ONE row coordinate: LDS row i is global row g0 + i.  H = (KS-1)/2, h1p = dp * H, M0 = 0,
// M(p+1) = Mp + h1p + H, Mtot = M(np).  The tile loads R0 = BN + 2 h1_0 rows of x from g0 = r0 - Mtot; every convolution computes BN rows:
//   conv1 of pair p: rows Mp + h1p + n  (n < BN)  <-  XA rows Mp + n + j dp
//   conv2 of pair p: rows M(p+1) + n              <-  T1 rows Mp + h1p + n + j
// and what lies outside the shrinking valid range [M(p+1), R0 - M(p+1)) is finite garbage that only ever feeds garbage rows (a row of the
// B operand touches one output row).  TT = R0 - 2 Mtot rows are stored per tile (dilations 1, 3, 5: BN - 22 H).  Rows outside the
// sequence are forced to zero at every stage (each convolution of the unfused chain zero-pads ITS input).
//   XA [RB][C+8]  lrelu(y(p))      RS [RB][C+8]  y(p) (the residual; updated in place)      T1 [RB][C+8]  lrelu(conv1 + b1)
//   RB = BN + 16 H rows (the reach of pair 2 of a (1, 3, 5) branch); WS / BIAS as in the pair kernel.
#pragma once

#include "resblock_pair_kernel.h"

namespace evmi {

struct BranchArgs {
  const bf16_t* x;    // [B][T][C]
  bf16_t* out;        // [B][T][C]
  const bf16_t* w[6];  // conv1, conv2 of pair 0, 1, 2: [KS][C][C] bf16, tap-major (the pair kernel's layout)
  const float* b[6];
  int T;
  int np;             // pairs (1..3)
  int dil[3];
  int tt;             // valid rows per tile (host: BN + 2 h1_0 - 2 Mtot)
  int tiles_per_item;
  int n_tiles;
  float slope, post_slope, out_scale;
  int accumulate;
};

struct BranchLaunch {
  void (*kernel)(BranchArgs);
  int c, ks, bn, rb, threads;
  size_t lds_bytes;
  const char* name;
};

// WRES_: the weights of all 2 NP convolutions stay in registers for the life of the persistent workgroup (loaded once) and are committed to
// the LDS from there every step -- a k = 3 step is 12 MFMAs per wave, far shorter than the L2 round trip of the next step's weights that
// the pair kernel's one-step-ahead prefetch has to cover
// WM_: waves along the output channels (the others along the rows)
template <int C_, int KS_, int BN_, int TAPS_, int WAVES_, int NWBUF_, int NP_ = 3, int WRES_ = 1, int WM_ = 1>
struct BranchCfg {
  static constexpr int C = C_, KS = KS_, BN = BN_, TAPS = TAPS_, WAVES = WAVES_, NWBUF = NWBUF_, NP = NP_, WRES = WRES_;
  static constexpr int WM = WM_, WN = WAVES / WM;
  static constexpr int NTHREADS = WAVES * 64;
  static constexpr int MT = C / (WM * 32), NT = BN / (WN * 32);
  static constexpr int S = C + 8;
  static constexpr int H = (KS - 1) / 2;
  static constexpr int RB = BN + 16 * H;
  static constexpr int NG = (KS + TAPS - 1) / TAPS;
  static constexpr int LAST_TAPS = KS - (NG - 1) * TAPS;
  static constexpr int W_TILE = TAPS * C * S;
  static constexpr int W_VECS = TAPS * C * (C / 8);
  static constexpr int W_PER_THREAD = (W_VECS + NTHREADS - 1) / NTHREADS;
  static constexpr bool W_EXACT = (KS % TAPS == 0) && (W_VECS % NTHREADS == 0);
  static constexpr int X_PER_THREAD = (RB * (C / 8) + NTHREADS - 1) / NTHREADS;
  static constexpr size_t OFF_XA = 0;
  static constexpr size_t OFF_RS = OFF_XA + size_t(RB) * S;
  static constexpr size_t OFF_T1 = OFF_RS + size_t(RB) * S;
  static constexpr size_t OFF_WS = OFF_T1 + size_t(RB) * S;
  static constexpr size_t OFF_BIAS = OFF_WS + NWBUF * size_t(W_TILE);  // 2 copies x 6 x C floats (in bf16 units: 24 C)
  static constexpr size_t LDS = (OFF_BIAS + 24 * size_t(C)) * 2;
  static_assert(BN % (WN * 32) == 0 && C % (WM * 32) == 0 && WAVES % WM == 0, "tiling");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <class P>
__global__ __launch_bounds__(P::NTHREADS) void resblock_branch_kernel(BranchArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* XA = reinterpret_cast<bf16_t*>(smem) + P::OFF_XA;
  bf16_t* RS = reinterpret_cast<bf16_t*>(smem) + P::OFF_RS;
  bf16_t* T1 = reinterpret_cast<bf16_t*>(smem) + P::OFF_T1;
  bf16_t* WS = reinterpret_cast<bf16_t*>(smem) + P::OFF_WS;
  float* BIAS = reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(smem) + P::OFF_BIAS);

  constexpr int C = P::C, S = P::S, KS = P::KS, H = P::H;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / P::WN, wn = wave % P::WN;
  const int cb = wm * P::MT * 32;  // first output channel of this wave
  constexpr int np = P::NP;
  int mtot = 0;
#pragma unroll
  for (int p = 0; p < np; ++p) mtot += a.dil[p] * H + H;
  const int r_in = P::BN + 2 * a.dil[0] * H;  // rows of x a tile loads
  const int x_nvec = r_in * (C / 8);
  constexpr int nstep = 2 * np * P::NG;

  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_wg = (nwg + 7) >> 3;
  const int tiles_per_xcd = (a.n_tiles + 7) >> 3;
  const int tile_lo = xcd * tiles_per_xcd;
  const int tile_hi = min(a.n_tiles, tile_lo + tiles_per_xcd);

  bf16x8 xreg[P::X_PER_THREAD];
  bf16x8 wreg[P::WRES ? nstep : 1][P::W_PER_THREAD];

  auto x_issue = [&](int tile) {
    const int item = tile / a.tiles_per_item, rt = tile % a.tiles_per_item;
    const int g0 = rt * a.tt - mtot;
    const bf16_t* xb = a.x + (long long)item * a.T * C;
#pragma unroll
    for (int i = 0; i < P::X_PER_THREAD; ++i) {
      const int v = tid + i * P::NTHREADS;
      const int row = v / (C / 8), c8 = v % (C / 8);
      const int g = g0 + row;
      bf16x8 val;
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] = (bf16_t)0.f;
      if (v < x_nvec && g >= 0 && g < a.T) val = *reinterpret_cast<const bf16x8*>(xb + (long long)g * C + c8 * 8);
      xreg[i] = val;
    }
  };
  auto x_commit = [&]() {
    const float sl = a.slope;
#pragma unroll
    for (int i = 0; i < P::X_PER_THREAD; ++i) {
      const int v = tid + i * P::NTHREADS;
      if (v < x_nvec) {
        const int row = v / (C / 8), c8 = v % (C / 8);
        const bf16x8 raw = xreg[i];
        *reinterpret_cast<bf16x8*>(XA + row * S + c8 * 8) = lrelu8_bf16(raw, sl);
        *reinterpret_cast<bf16x8*>(RS + row * S + c8 * 8) = raw;
      }
    }
  };
  // step s in [0, nstep): convolution s / NG (conv1, conv2 of pair 0, conv1 of pair 1, ...), tap group s % NG
  auto w_prefetch = [&](int s) {
    const int ci = s / P::NG, grp = s % P::NG;
    const bf16_t* src = a.w[ci] + (long long)grp * P::TAPS * C * C;
    const int ntaps = (KS - grp * P::TAPS) < P::TAPS ? (KS - grp * P::TAPS) : P::TAPS;
    const int nvec = ntaps * C * (C / 8);
#pragma unroll
    for (int i = 0; i < P::W_PER_THREAD; ++i) {
      const int v = tid + i * P::NTHREADS;
      if (P::W_EXACT || v < nvec) wreg[P::WRES ? s : 0][i] = *reinterpret_cast<const bf16x8*>(src + (long long)v * 8);
    }
  };
  auto w_commit = [&](int s) {
    bf16_t* dst = WS + (s & (P::NWBUF - 1)) * P::W_TILE;
    const int grp = s % P::NG;
    const int ntaps = (KS - grp * P::TAPS) < P::TAPS ? (KS - grp * P::TAPS) : P::TAPS;
    const int nvec = ntaps * C * (C / 8);
#pragma unroll
    for (int i = 0; i < P::W_PER_THREAD; ++i) {
      const int v = tid + i * P::NTHREADS;
      if (P::W_EXACT || v < nvec) {
        const int row = v / (C / 8), c8 = v % (C / 8);
        *reinterpret_cast<bf16x8*>(dst + row * S + c8 * 8) = wreg[P::WRES ? s : 0][i];
      }
    }
  };

  int tile = tile_lo + slot;
  if (tile >= tile_hi) return;
  // the three activation tiles start finite (what a tile never writes is read by garbage rows only, but must not be a NaN pattern
  // that another row's MFMA could not survive -- it cannot: a B row feeds one output row -- nor trap the activation arithmetic)
  {
    bf16x8 z;
#pragma unroll
    for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
    for (int v = tid; v < 3 * P::RB * S / 8; v += P::NTHREADS) *reinterpret_cast<bf16x8*>(XA + v * 8) = z;
  }
  for (int i = tid; i < 4 * np * C; i += P::NTHREADS) BIAS[i] = a.b[(i % (2 * np * C)) / C][i % C];  // (two copies: see the pair kernel)
  lds_barrier();
  x_issue(tile);
  if (P::WRES) {
#pragma unroll
    for (int s = 0; s < nstep; ++s) w_prefetch(s);
  } else {
    w_prefetch(0);
  }

  for (; tile < tile_hi; tile += per_xcd_wg) {
    const int item = tile / a.tiles_per_item, rt = tile % a.tiles_per_item;
    const int r0 = rt * a.tt;       // first output row of this tile
    const int g0 = r0 - mtot;       // global row of LDS row 0
    const bool edge = g0 < 0 || g0 + P::RB > a.T;  // (wave-uniform) the tile holds rows outside the sequence
    const int next = tile + per_xcd_wg;
    x_commit();

    f32x16 acc[P::MT][P::NT];
    int mp = 0;  // Mp
#pragma unroll
    for (int p = 0; p < np; ++p) {
      const int dil = a.dil[p];
      const int h1 = dil * H;
#pragma unroll
      for (int conv = 0; conv < 2; ++conv) {
        {  // accumulators start at the bias (accumulator layout: channels 8q + 4h .. + 3 per register quad): the epilogues only
           // activate / add the residual -- they are VALU-bound (a k = 3 tile: 1600 vector ops per wave against 72 MFMAs)
          const float* bias = BIAS + (2 * p + conv) * C;
#pragma unroll
          for (int i = 0; i < P::MT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int j = 0; j < P::NT; ++j) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (j & 1) * 2 * np * C + cb + i * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] = bv[r];
              }
        }
        // conv1: XA rows Mp + n + j d;  conv2: T1 rows Mp + h1 + n + j
        const bf16_t* Bsrc = conv ? T1 + (mp + h1) * S : XA + mp * S;
        const int b_tap_stride = (conv ? 1 : dil) * S;
#pragma unroll
        for (int grp = 0; grp < P::NG; ++grp) {
          const int s = (2 * p + conv) * P::NG + grp;
          if (P::NWBUF == 1 && s > 0) lds_barrier();  // everyone is done reading the single weight buffer (s = 0: the barrier that ends a tile)
          w_commit(s);
          lds_barrier();
          if (!P::WRES) w_prefetch(s + 1 == nstep ? 0 : s + 1);  // wraps to the next tile's first group
          if (s == 0 && next < tile_hi) x_issue(next);
          const bf16_t* Arow = WS + (s & (P::NWBUF - 1)) * P::W_TILE + (cb + (lane & 31)) * S + (lane >> 5) * 8;
          const bf16_t* Brow = Bsrc + (wn * P::NT * 32 + (lane & 31)) * S + grp * P::TAPS * b_tap_stride + (lane >> 5) * 8;
          if (grp + 1 < P::NG || P::LAST_TAPS == P::TAPS)
            mma_tap_group<P::MT, P::NT, C / 16, P::TAPS, C * S, 32 * S, 32 * S>(Arow, Brow, b_tap_stride, acc);
          else
            mma_tap_group<P::MT, P::NT, C / 16, P::LAST_TAPS, C * S, 32 * S, 32 * S>(Arow, Brow, b_tap_stride, acc);
        }
        if (conv == 0) {
          // T1[Mp + h1 + n] = lrelu(conv1 + b1), zero outside the sequence (only a tile at an end of the sequence has such rows)
          const float sl = a.slope;
#pragma unroll
          for (int mt = 0; mt < P::MT; ++mt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int c = cb + mt * 32 + 8 * q + 4 * (lane >> 5);
#pragma unroll
              for (int nt = 0; nt < P::NT; ++nt) {
                const int n = wn * P::NT * 32 + nt * 32 + (lane & 31);
                const int row = mp + h1 + n;
                bf16x4 pk = lrelu4_bf16(f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]}, sl);
                if (edge) {
                  const int g = g0 + row;
                  if (g < 0 || g >= a.T) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) pk[i] = (bf16_t)0.f;
                  }
                }
                *reinterpret_cast<bf16x4*>(T1 + row * S + c) = pk;
              }
            }
          }
        }
      }
      const int mnext = mp + h1 + H;  // M(p+1): LDS row of conv2's output row 0
      if (p + 1 < np) {
        // y(p+1) = y(p) + b2 + conv2, rounded to bf16 as the pair kernel stores it; raw into RS (in place: a lane reads and writes
        // its own elements), its activation into XA (dead since conv1 of this pair); zero outside the sequence
        const float sl = a.slope;
#pragma unroll
        for (int mt = 0; mt < P::MT; ++mt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = cb + mt * 32 + 8 * q + 4 * (lane >> 5);
#pragma unroll
            for (int nt = 0; nt < P::NT; ++nt) {
              const int n = wn * P::NT * 32 + nt * 32 + (lane & 31);
              const int row = mnext + n;
              const bf16x4 rv = *reinterpret_cast<const bf16x4*>(RS + row * S + c);
              bf16x4 raw, act;
#pragma unroll
              for (int i = 0; i < 4; ++i) raw[i] = (bf16_t)(acc[mt][nt][4 * q + i] + (float)rv[i]);
              if (edge) {
                const int g = g0 + row;
                if (g < 0 || g >= a.T) {
#pragma unroll
                  for (int i = 0; i < 4; ++i) raw[i] = (bf16_t)0.f;
                }
              }
              act = lrelu4_bf16(f32x4{(float)raw[0], (float)raw[1], (float)raw[2], (float)raw[3]}, sl);
              *reinterpret_cast<bf16x4*>(RS + row * S + c) = raw;
              *reinterpret_cast<bf16x4*>(XA + row * S + c) = act;
            }
          }
        }
      } else {
        // ---- final epilogue: conv2 + b2 + residual (LDS) -> registers -> 16-byte stores (as the pair kernel's) ------------------
        bf16_t* ob = a.out + (long long)item * a.T * C;
        const float scale = a.out_scale, post = a.post_slope;
        const int hh = lane >> 5;
#pragma unroll
        for (int nt = 0; nt < P::NT; ++nt) {
          const int n = wn * P::NT * 32 + nt * 32 + (lane & 31);
          const int r = r0 + n;
          const bool ok = n < a.tt && r < a.T;
          bf16_t* dst = ob + (long long)(ok ? r : 0) * C + 8 * hh;
          u32x4 pv[P::MT][2];
          if (a.accumulate) {
#pragma unroll
            for (int mt = 0; mt < P::MT; ++mt)
#pragma unroll
              for (int p2 = 0; p2 < 2; ++p2) pv[mt][p2] = *reinterpret_cast<const u32x4*>(dst + cb + mt * 32 + 16 * p2);
          }
#pragma unroll
          for (int mt = 0; mt < P::MT; ++mt)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
              float f[8];
#pragma unroll
              for (int qq = 0; qq < 2; ++qq) {
                const int c = cb + mt * 32 + 8 * (2 * p2 + qq) + 4 * hh;
                const bf16x4 rv = *reinterpret_cast<const bf16x4*>(RS + (mnext + n) * S + c);
#pragma unroll
                for (int i = 0; i < 4; ++i) f[4 * qq + i] = (acc[mt][nt][4 * (2 * p2 + qq) + i] + (float)rv[i]) * scale;
              }
              if (a.accumulate) {
                const u32x4 d = swap_quads_bf16(pv[mt][p2]);
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
              if (ok) *reinterpret_cast<u32x4*>(dst + cb + mt * 32 + 16 * p2) = o;
            }
        }
      }
      mp = mnext;
    }
    lds_barrier();  // the activation tiles are free again for the next tile's commit
  }
}

template <class P>
static BranchLaunch make_branch_launch(const char* name) {
  BranchLaunch l;
  l.kernel = resblock_branch_kernel<P>;
  l.c = P::C;
  l.ks = P::KS;
  l.bn = P::BN;
  l.rb = P::RB;
  l.threads = P::NTHREADS;
  l.lds_bytes = P::LDS;
  l.name = name;
  return l;
}

// nullptr when no instantiation takes this branch (channels, kernel size, the dilations' reach inside the LDS tiles)
const BranchLaunch* find_resblock_branch(int c, int ks, int np, const int* dil);
int launch_resblock_branch(const BranchLaunch* L, BranchArgs a, int B, int n_cu, hipStream_t stream);

}  // namespace evmi
