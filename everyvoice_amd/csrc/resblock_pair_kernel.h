// Fused residual pair of a HiFiGAN ResBlock1 on MFMA, persistent over row tiles:
//
//   out = post( [accumulate ? out : 0] + out_scale * ( x + b2 + conv2( lrelu( b1 + conv1_dil( lrelu(x) ) ) ) ) )
//
// Both convolutions (C -> C channels, KS taps; conv1 dilated, conv2 dense) run back to back on one
// row tile with the intermediate kept in LDS, so per pair the residual stream is read once and
// written once (the unfused path moves five tensors).  Only used where the per-tile working set
// fits LDS (C <= 64).
//
// Per tile of BN rows computed per convolution (TT = BN - (KS-1) of them valid after conv2):
//   XA [BN + (KS-1)*dil][C+8]  lrelu(x), zero outside [0, T)      (B operand of conv1)
//   RS [BN][C+8]               raw x of the output rows            (residual, no second HBM read)
//   T1 [BN + KS-1][C+8]        lrelu(conv1 + b1), zero outside [0, T)  (B operand of conv2)
//   WS 2 x [TAPS][C][C+8]      weight tap groups, streamed global -> regs -> LDS one group ahead
// The workgroup is persistent: it walks tiles tile0, tile0 + grid, ... and keeps the NEXT tile's
// activation rows in flight in registers while the current tile computes, so the only exposed
// HBM latency is the first tile's.
#pragma once

#include "common.h"

namespace evmi {

struct PairArgs {
  const bf16_t* x;   // [B][T][C] residual stream (raw)
  const bf16_t* w1;  // [KS][C][C] bf16, tap-major (kernel layout)
  const bf16_t* w2;
  const float* b1;
  const float* b2;
  bf16_t* out;       // [B][T][C]
  int T;
  int dil1;
  int tiles_per_item;
  int n_tiles;       // B * tiles_per_item
  float slope;       // leaky-relu slope on x (load) and on conv1's output
  float post_slope;
  float out_scale;
  int accumulate;
  long long* timeline;  // debug builds only (PairCfg::DBG): cycle stamps of workgroup 0, else nullptr
};

struct PairLaunch {
  void (*kernel)(PairArgs);
  int c, ks, bn, tt, threads, max_dil;
  int wg_per_cu = 1;  // persistent workgroups per CU (LDS-bound residency)
  int kc;  // channel chunk of the weight layout [chunk][tap][C][kc] (== c when the layer is not chunked)
  size_t lds_bytes;
  const char* name;
};

// OVL_: must be 0 (round 2's form with T1 overlaying the conv1 operand tile, two workgroups per CU, was measured slower and removed in round 5)
// WRES_: the weights of both convolutions stay in registers for the life of the persistent workgroup (NSTEP x W_PER_THREAD vectors,
// loaded once) and are committed to the LDS from there: a two-tap step at 64 channels is 16 MFMAs per wave, shorter than the L2 round
// trip of the next step's weights that the one-step-ahead prefetch has to cover.
// WM_: waves along the output channels (the others along the rows): 64 channels on sixteen waves = 2 x 8 of 32 x 32 tiles
template <int C_, int KS_, int BN_, int TAPS_, int MAXDIL_, int WAVES_, int DBG_ = 0, int NWBUF_ = 2, int OVL_ = 0, int WRES_ = 0, int WM_ = 1>
struct PairCfg {
  static constexpr int C = C_, KS = KS_, BN = BN_, TAPS = TAPS_, MAXDIL = MAXDIL_, WAVES = WAVES_, DBG = DBG_, OVL = OVL_, WRES = WRES_;
  static constexpr int WM = WM_, WN = WAVES / WM;
  // weight tap-group buffers in LDS: 2 = commit the next group while the current one is read; 1 = one
  // buffer (lets a whole convolution's taps sit in LDS at once) at the price of a barrier before each commit
  static constexpr int NWBUF = NWBUF_;
  static constexpr int NTHREADS = WAVES * 64;
  static constexpr int MT = C / (WM * 32), NT = BN / (WN * 32);
  static constexpr int S = C + 8;  // LDS row stride (elements): odd multiple of 16 B -> conflict-free b128 reads
  static constexpr int TT = BN - (KS - 1);
  static constexpr int RA_MAX = BN + (KS - 1) * MAXDIL;
  static constexpr int T1_ROWS = BN + KS - 1;
  static constexpr int NG = (KS + TAPS - 1) / TAPS;  // tap groups per convolution
  static constexpr int LAST_TAPS = KS - (NG - 1) * TAPS;  // taps in the final (possibly short) group
  static constexpr int NSTEP = 2 * NG;
  static constexpr int W_TILE = TAPS * C * S;
  static constexpr int W_VECS = TAPS * C * (C / 8);
  static constexpr int W_PER_THREAD = (W_VECS + NTHREADS - 1) / NTHREADS;
  static constexpr bool W_EXACT = (KS % TAPS == 0) && (W_VECS % NTHREADS == 0);
  static constexpr int X_VECS_MAX = RA_MAX * (C / 8);
  static constexpr int X_PER_THREAD = (X_VECS_MAX + NTHREADS - 1) / NTHREADS;
  static constexpr size_t OFF_XA = 0;
  static constexpr size_t OFF_RS = OFF_XA + size_t(RA_MAX) * S;
  static constexpr size_t OFF_T1 = OFF_RS + size_t(BN) * S;
  static constexpr size_t OFF_WS = OFF_T1 + size_t(T1_ROWS) * S;
  static constexpr size_t OFF_BIAS = OFF_WS + NWBUF * size_t(W_TILE);  // 2 copies x 2 x C floats (in bf16 units: 8 C)
  static constexpr size_t LDS = (OFF_BIAS + 8 * size_t(C)) * 2;
  static constexpr int WG_PER_CU = (2 * LDS <= 160 * 1024 && WAVES <= 4) ? 2 : 1;
  static_assert(BN % (WN * 32) == 0 && C % (WM * 32) == 0 && WAVES % WM == 0, "tiling");
  static_assert(OVL == 0, "the overlay form was removed");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <class P>
__global__ __launch_bounds__(P::NTHREADS) void resblock_pair_kernel(PairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* XA = reinterpret_cast<bf16_t*>(smem) + P::OFF_XA;
  bf16_t* RS = reinterpret_cast<bf16_t*>(smem) + P::OFF_RS;
  bf16_t* T1 = reinterpret_cast<bf16_t*>(smem) + P::OFF_T1;
  bf16_t* WS = reinterpret_cast<bf16_t*>(smem) + P::OFF_WS;

  constexpr int C = P::C, S = P::S, KS = P::KS, H2 = (KS - 1) / 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / P::WN, wn = wave % P::WN;
  const int cb = wm * P::MT * 32;  // first output channel of this wave
  const int h1 = a.dil1 * (KS - 1) / 2;
  const int ra = P::BN + (KS - 1) * a.dil1;  // rows of XA actually needed
  const int x_nvec = ra * (C / 8);

  // XCD-aware tile walk: workgroup b runs on XCD b % 8; give each XCD one contiguous range of
  // tiles so neighbouring tiles (which share halo rows) meet in the same L2.
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd_wg = (nwg + 7) >> 3;
  const int tiles_per_xcd = (a.n_tiles + 7) >> 3;
  const int tile_lo = xcd * tiles_per_xcd;
  const int tile_hi = min(a.n_tiles, tile_lo + tiles_per_xcd);

  bf16x8 xreg[P::X_PER_THREAD];
  bf16x8 wreg[P::WRES ? P::NSTEP : 1][P::W_PER_THREAD];

  auto x_issue = [&](int tile) {
    const int item = tile / a.tiles_per_item, rt = tile % a.tiles_per_item;
    const int g0 = rt * P::TT - H2 - h1;  // global row of XA row 0
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
        const bf16x8 act = lrelu8_bf16(raw, sl);
        *reinterpret_cast<bf16x8*>(XA + row * S + c8 * 8) = act;
        const int n = row - H2 - h1;
        if (n >= 0 && n < P::BN) *reinterpret_cast<bf16x8*>(RS + n * S + c8 * 8) = raw;
      }
    }
  };
  // step s in [0, NSTEP): conv = s / NG, tap group = s % NG
  auto w_prefetch = [&](int s) {
    const int conv = s / P::NG, grp = s % P::NG;
    const bf16_t* src = (conv ? a.w2 : a.w1) + (long long)grp * P::TAPS * C * C;
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

  int n_stamp = 0;
  auto stamp = [&]() {
    if (P::DBG && a.timeline && blockIdx.x == 0 && tid == 0 && n_stamp < 256)
      a.timeline[n_stamp++] = (long long)__builtin_readcyclecounter();
  };
  int tile = tile_lo + slot;
  if (tile >= tile_hi) return;
  // both bias vectors sit in LDS for the life of the (persistent) workgroup: accumulators start at the bias, read in the
  // accumulator layout (channels 8q + 4h .. + 3 per register quad) under the first weight commit of each convolution
  float* BIAS = reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(smem) + P::OFF_BIAS);
  // (two copies: the accumulators of the second row tile are initialised by loads of their own -- from one copy the compiler shares the
  //  loads and pays a v_mov per accumulator register, on the VALU these kernels are bound by)
  for (int i = tid; i < 4 * C; i += P::NTHREADS) BIAS[i] = (i % (2 * C)) < C ? a.b1[i % (2 * C)] : a.b2[i % (2 * C) - C];
  lds_barrier();  // (the first tile's accumulators are initialised from it before the first step's barrier)
  x_issue(tile);
  if (P::WRES) {
#pragma unroll
    for (int s = 0; s < P::NSTEP; ++s) w_prefetch(s);
  } else {
    w_prefetch(0);
  }

  for (; tile < tile_hi; tile += per_xcd_wg) {
    const int item = tile / a.tiles_per_item, rt = tile % a.tiles_per_item;
    const int r0 = rt * P::TT;  // first output row of this tile
    const int next = tile + per_xcd_wg;
    stamp();  // tile start
    x_commit();
    stamp();  // x committed

    f32x16 acc[P::MT][P::NT];
#pragma unroll
    for (int conv = 0; conv < 2; ++conv) {
      {  // accumulators start at the bias (accumulator layout: channels 8q + 4h .. + 3 per register quad): the epilogues, which are
         // VALU-bound at these channel counts, only activate / add the residual
        const float* bias = BIAS + conv * C;
#pragma unroll
        for (int i = 0; i < P::MT; ++i)
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < P::NT; ++j) {
              const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + (j & 1) * 2 * C + cb + i * 32 + 8 * q + 4 * (lane >> 5));
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[i][j][4 * q + r] = bv[r];
            }
      }
      const bf16_t* Bsrc = conv ? T1 : XA;
      const int b_tap_stride = (conv ? 1 : a.dil1) * S;
      // tap groups are fully unrolled so that the (possibly shorter) last group is static too
#pragma unroll
      for (int grp = 0; grp < P::NG; ++grp) {
        const int s = conv * P::NG + grp;
        if (P::NWBUF == 1 && s > 0) lds_barrier();  // everyone is done reading the single weight buffer
        w_commit(s);
        lds_barrier();
        if (!P::WRES) w_prefetch(s + 1 == P::NSTEP ? 0 : s + 1);  // wraps to the next tile's first group
        if (s == 0 && next < tile_hi) x_issue(next);  // after the weight loads: they stay in flight
        const bf16_t* Arow = WS + (s & (P::NWBUF - 1)) * P::W_TILE + (cb + (lane & 31)) * S + (lane >> 5) * 8;
        const bf16_t* Brow = Bsrc + (wn * P::NT * 32 + (lane & 31)) * S + grp * P::TAPS * b_tap_stride + (lane >> 5) * 8;
        if (grp + 1 < P::NG || P::LAST_TAPS == P::TAPS)
          mma_tap_group<P::MT, P::NT, C / 16, P::TAPS, C * S, 32 * S, 32 * S>(Arow, Brow, b_tap_stride, acc);
        else
          mma_tap_group<P::MT, P::NT, C / 16, P::LAST_TAPS, C * S, 32 * S, 32 * S>(Arow, Brow, b_tap_stride, acc);
        stamp();  // step done
      }
      if (conv == 0) {
        // T1[n] = lrelu(conv1 + b1) for global row r0 - H2 + n, zero outside the sequence
        const float sl = a.slope;
        const bool edge = r0 - H2 < 0 || r0 - H2 + P::BN > a.T;  // (wave-uniform) rows outside the sequence in this tile
#pragma unroll
        for (int mt = 0; mt < P::MT; ++mt) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = cb + mt * 32 + 8 * q + 4 * (lane >> 5);
#pragma unroll
            for (int nt = 0; nt < P::NT; ++nt) {
              const int n = wn * P::NT * 32 + nt * 32 + (lane & 31);
              bf16x4 pk = lrelu4_bf16(f32x4{acc[mt][nt][4 * q], acc[mt][nt][4 * q + 1], acc[mt][nt][4 * q + 2], acc[mt][nt][4 * q + 3]}, sl);
              if (edge) {
                const int g = r0 - H2 + n;
                if (g < 0 || g >= a.T) {
#pragma unroll
                  for (int i = 0; i < 4; ++i) pk[i] = (bf16_t)0.f;
                }
              }
              *reinterpret_cast<bf16x4*>(T1 + n * S + c) = pk;
            }
          }
        }
        // rows BN .. BN+KS-2 of T1 only feed discarded outputs; keep them finite
        for (int v = tid; v < (KS - 1) * (C / 8); v += P::NTHREADS) {
          bf16x8 z;
#pragma unroll
          for (int e = 0; e < 8; ++e) z[e] = (bf16_t)0.f;
          *reinterpret_cast<bf16x8*>(T1 + (P::BN + v / (C / 8)) * S + (v % (C / 8)) * 8) = z;
        }
      }
    }

    stamp();  // conv loops done (includes the T1 epilogue of conv1)
    // ---- epilogue: conv2 + b2 + residual (LDS, accumulator layout) -> registers -> 16-byte stores ---------------
    // No staging pass and no barrier: every wave retires its rows on its own.  Lane (n, h) holds channels 8q + 4h .. + 3 of row
    // n per register quad q; swap_quads_bf16 turns the packed quads (2p, 2p + 1) into 16 contiguous bytes per lane.
    {
      bf16_t* ob = a.out + (long long)item * a.T * C;
      const float scale = a.out_scale, post = a.post_slope;
      const int hh = lane >> 5;
#pragma unroll
      for (int nt = 0; nt < P::NT; ++nt) {
        const int n = wn * P::NT * 32 + nt * 32 + (lane & 31);
        const int r = r0 + n;
        const bool ok = n < P::TT && r < a.T;
        bf16_t* dst = ob + (long long)(ok ? r : 0) * C + 8 * hh;
        u32x4 pv[P::MT][2];
        if (a.accumulate) {  // wave-uniform; all of the lane's vectors are requested before the first is consumed
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
              const bf16x4 rv = *reinterpret_cast<const bf16x4*>(RS + n * S + c);
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
    stamp();  // stores issued
    lds_barrier();  // staging / residual tiles are free again for the next tile's commit
  }
}

template <class P>
static PairLaunch make_pair_launch(const char* name) {
  PairLaunch l;
  l.kernel = resblock_pair_kernel<P>;
  l.c = P::C;
  l.ks = P::KS;
  l.bn = P::BN;
  l.tt = P::TT;
  l.threads = P::NTHREADS;
  l.max_dil = P::MAXDIL;
  l.lds_bytes = P::LDS;
  l.name = name;
  l.kc = P::C;
  l.wg_per_cu = P::WG_PER_CU;
  return l;
}

const PairLaunch* find_resblock_pair(int c, int ks, int dil);
int launch_resblock_pair(const PairLaunch* L, PairArgs a, int B, int n_cu, hipStream_t stream);

}  // namespace evmi
