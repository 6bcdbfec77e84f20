// "Ping-pong" MFMA implicit-GEMM convolution for the wide residual-stack layers of the generator (C_in = C_out in {128, 256},
// k in {7, 11}): same contract and arguments as conv_tc_dma_kernel (conv_tc_mfma.h), a different execution structure.
//
// conv_tc_dma_kernel runs its eight waves in lockstep (one barrier per (chunk, tap) step; everybody requests weights, reads
// fragments, multiplies) and relies on a SECOND workgroup on the CU for phase diversity: the matrix pipes are busy 57.6 % of the
// clocked cycles (profiles/r04z_pmc_summary.json), the tile prologue / epilogue and the start of every step are exposed.  Here ONE
// persistent workgroup per CU alternates roles inside each SIMD (MI355X_MICROARCH.md "Two waves per SIMD"):
//
//   * 8 waves, wave w and w + 4 share a SIMD.  Waves 0-3 (group 0) and 4-7 (group 1) run ONE instruction stream -- per phase:
//     the fragment reads L(k), then the MFMA segment M(k) (16 x v_mfma_f32_32x32x16_bf16 from registers, s_setprio 1) with the
//     phase's side work sliced between the MFMAs -- and ONE barrier per phase whose POSITION depends on the group: group 0 waits
//     behind M(k), group 1 between L(k) and M(k).  Between two barriers group 0 therefore runs L(k), M(k) and group 1 runs
//     M(k - 1), L(k): on every SIMD one wave has MFMAs to issue while its partner waits for the LDS, and the register allocator
//     sees straight-line code (the same two roles written as two branches spilled the accumulators).
//   * workgroup tile 128 channels x 512 rows, wave tile 128 x 64 (MT = 4, NT = 2: 0.75 LDS fragment reads per MFMA instead of 1),
//     K in sub-chunks of 32 channels: a phase = (sub-chunk, tap) = 2 k-steps of 16 = 16 MFMAs per wave.
//   * LDS: weight images of 8 KB (128 rows x 64 B) per phase in a ring of four, requested three phases ahead BY GROUP 0 (two
//     pieces per wave and phase: they come out of the L2 and land within an interval); the activation rows of a sub-chunk
//     ((512 + (KS - 1) dil) rows x 64 B) double buffered and requested BY GROUP 1 during the first taps of the previous sub-chunk
//     -- of the previous TILE for a tile's first sub-chunk -- with D intervals to land: they come from HBM (1-2 us under load),
//     and vmcnt retires in order, so a wave that requested both kinds would wait for the slow rows every time it needs a weight
//     piece (measured: +1000 cycles per phase with a request one interval ahead).  Rows are 64 B = four 16-byte slots, slot p of
//     row r holds channel vector p ^ ((r >> 2) & 3): the 16-lane groups of ds_read_b128 (16 consecutive rows, one vector) hit 16
//     distinct slots of the 256-byte bank row; the permutation is applied on the source side of the LDS-DMA (weights: by the
//     host, layout 3) and again on the fragment read.
//   * every LDS-DMA request is waited for by the wave that issued it with a COUNTED vmcnt in front of its barrier, and read
//     by others one barrier later at the earliest.  The leaky ReLU on load is applied in place by the requesting lane to its own
//     16 bytes once they have landed (no barrier between landing and the pass: same wave), sliced between MFMAs as well.
//   * epilogue straight from registers (bias from LDS into the initial accumulators; scale, residual / running sum fetched in the
//     accumulator layout, activation, v_permlane32_swap, 16-byte stores) in front of the next tile's first fragment reads: for
//     both groups that is the same interval.
#pragma once

#include <type_traits>
#include <utility>

#include "conv_tc_dma_kernel.h"

namespace evmi {

// DBG: s_memtime stamps of workgroup 3 into args.timeline (tools/conv_timeline.py).  VAR (A/B, tools/sweep_conv.py):
//   1 = no s_setprio around the MFMA segments   2 = no group skew (every wave waits behind its MFMA segment: lockstep)
//   ablations (results wrong, timing only): 16 = no epilogue, 32 = no activation-row requests after the first sub-chunk,
//   64 = no weight requests after the prologue
template <int CIN_, int KS_, int MAXDIL_, int DBG_ = 0, int VAR_ = 0>
struct ConvPpCfg {
  static constexpr int CIN = CIN_, KS = KS_, MAXDIL = MAXDIL_, DBG = DBG_, VAR = VAR_;
  static constexpr int KC = 32, BM = 128, BN = 512, NWAVES = 8, NTHREADS = 512, MT = 4, NT = 2;
  static constexpr int NSUB = CIN / KC, NP = NSUB * KS;
  static constexpr int R_MAX = BN + (KS - 1) * MAXDIL;
  static constexpr int A_BYTES = BM * KC * 2, A_RING = 4;
  static constexpr int BIAS_FLOATS = 512;
  // Activation pieces (16 rows x 64 B) of the next sub-chunk: requested by the four waves of group 1 in the side work of taps
  // 0 .. XTAPS - 1, NXW per wave and tap.  Group 1 runs the side work of phase k in interval k + 1; a piece is waited for XD
  // intervals later (counted vmcnt: the requests of the last XD phases stay in flight), activated in the side work of phase
  // k + XD + 1 = interval k + XD + 2, and first read in interval KS: XTAPS - 1 + XD + 3 <= KS.
  static constexpr int XD = KS >= 11 ? 4 : 2;
  static constexpr int XTAPS = KS - XD - 2;
  static constexpr int NXW = ((R_MAX + 15) / 16 + 4 * XTAPS - 1) / (4 * XTAPS);
  static constexpr int PW = NXW * XTAPS;  // pieces per wave of group 1 and sub-chunk
  static constexpr int X_PIECES = 4 * PW, X_BYTES = X_PIECES * 1024;
  static constexpr size_t LDS = size_t(A_RING) * A_BYTES + 2 * size_t(X_BYTES) + BIAS_FLOATS * 4;
  static constexpr int xcnt(int tap) { return (tap >= 0 && tap < XTAPS) ? NXW : 0; }
  static constexpr int xfly(int tap) {  // requests of the XD phases up to `tap` (taps of the previous sub-chunk below 0)
    int s = 0;
    for (int d = 0; d < XD; ++d) s += xcnt((tap - d + KS) % KS);
    return s;
  }
  static_assert(CIN % KC == 0 && NSUB % 2 == 0, "sub-chunks alternate between two buffers per tile");
  static_assert(NP % A_RING == 0, "the weight ring position is the phase index of the tile");
  static_assert(XTAPS >= 1 && XTAPS + XD < KS, "in-place pass of the last request ends before the sub-chunk does");
  static_assert(NXW >= 1 && NXW <= 3, "side-work slices for up to three requests / in-place passes per phase");
  static_assert(X_PIECES * 16 >= R_MAX, "every row has a piece");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <int N, class F, int... I>
__device__ __forceinline__ void pp_static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void pp_static_for(F&& f) {
  pp_static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt immediate");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class C>
__global__ __launch_bounds__(C::NTHREADS, 2) void conv_tc_pp_kernel(ConvTcArgs a) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* const As = smem;
  char* const Xs = smem + C::A_RING * C::A_BYTES;
  float* const Bs = reinterpret_cast<float*>(smem + C::A_RING * C::A_BYTES + 2 * C::X_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                          // who requests what
  const int late = (C::VAR & 2) ? 0 : (wave >> 2);    // where the barrier of a phase sits
  const int l5 = lane & 31, h = lane >> 5;

  int n_stamp = 0;
  auto stamp = [&]() __attribute__((always_inline)) {
    if (C::DBG && a.timeline && blockIdx.x == 3 && lane == 0 && n_stamp < 126)
      a.timeline[wave * 128 + n_stamp++] = (long long)__builtin_readcyclecounter();
  };
  stamp();

  // ---- tiles: (item, row tile) pairs in groups of eight consecutive workgroups = one per XCD; the m-tiles of a pair follow each
  // other eight tile indices apart, i.e. on the SAME XCD: its L2 serves the second m-tile's activation rows
  const int n_rt = (a.n_rows + C::BN - 1) / C::BN;
  const int n_mt = a.c_out / C::BM;
  const int n_rb = a.n_items * n_rt;
  const int total = ((n_rb + 7) / 8) * 8 * n_mt;
  struct Tile {
    int idx, r0, m0;
    const bf16_t* xb;
    const bf16_t* wb;
    long long ob;
  };
  auto tile_from = [&](int idx) __attribute__((always_inline)) -> Tile {  // first valid tile at or behind idx (idx >= total: none)
    Tile t;
    t.r0 = t.m0 = 0;
    t.xb = a.x;
    t.wb = a.w;
    t.ob = 0;
    for (;; idx += gridDim.x) {
      t.idx = idx;
      if (idx >= total) break;
      const int within = idx % (8 * n_mt);
      const int rb = (idx / (8 * n_mt)) * 8 + (within & 7);
      if (rb >= n_rb) continue;
      const int mt = within >> 3, b = rb / n_rt;
      t.r0 = (rb % n_rt) * C::BN;
      t.m0 = mt * C::BM;
      t.xb = a.x + (long long)b * a.x_batch_stride;
      t.wb = a.w + (long long)mt * C::NP * (C::BM * C::KC);  // layout 3: [mtile][sub-chunk][tap][BM rows][4 slots]
      t.ob = (long long)b * a.out_batch_stride;
      break;
    }
    return t;
  };

  const int rows_needed = C::BN + (C::KS - 1) * a.dil;
  const float pre = a.pre_slope;

  // ---- LDS-DMA requests (one wave-instruction = 1 KiB, lane-linear in the LDS) -------------------------------------------------
  auto issue_a = [&](const Tile& t, int phase, int slot, int piece) __attribute__((always_inline)) {  // 16 rows of a weight image
    if (C::VAR & 64) return;
    lds_dma_b128(t.wb + (long long)phase * (C::BM * C::KC) + piece * 512 + lane * 8, As + slot * C::A_BYTES + piece * 1024);
  };
  const int x_slot = lane & 3, x_c8 = (lane & 3) ^ ((lane >> 4) & 3);  // piece row r = 16 p + (lane >> 2): (r >> 2) & 3 = (lane >> 4) & 3
  auto issue_x = [&](const Tile& t, int sub, int buf, int p) __attribute__((always_inline)) {  // piece p of sub-chunk `sub` of tile t
    int lrow = lane >> 2;
    asm volatile("" : "+v"(lrow));  // keeps the per-piece address arithmetic where it is used (hoisted out of the sub-chunk loop it
                                    // holds ~20 registers for the whole tile)
    const int row = p * 16 + lrow;
    const int rr = t.r0 - a.pad + row;
    const bf16_t* src = (row < rows_needed && rr >= 0 && rr < a.t_in) ? t.xb + (long long)rr * C::CIN + sub * C::KC + x_c8 * 8
                                                                      : g_conv_dma_zero_row + x_slot * 8;
    lds_dma_b128(src, Xs + buf * C::X_BYTES + p * 1024);
  };
  // leaky ReLU in place on the 16 bytes this lane requested (they have landed), in stages that fit between two MFMAs
  auto act_ptr = [&](int buf, int p) __attribute__((always_inline)) -> bf16x8* {
    return reinterpret_cast<bf16x8*>(Xs + buf * C::X_BYTES + p * 1024 + lane * 16);
  };
  auto act_half = [&](bf16x8& v, int e) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float f = (float)v[e + i];
      v[e + i] = (bf16_t)fmaxf(f, f * pre);
    }
  };

  // ---- fragment offsets ------------------------------------------------------------------------------------------------------------
  // row r of a 64-byte-row image: byte r * 64 + ((vector ^ ((r >> 2) & 3)) << 4); vector = 2 ks + h of the 32-channel sub-chunk
  int off_a[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) off_a[ks] = l5 * 64 + (((2 * ks + h) ^ ((l5 >> 2) & 3)) << 4);
  const int row_b0 = wave * (C::NT * 32) + l5;

  f32x16 acc[C::MT][C::NT];
  bf16x8 af[2][C::MT], bfr[2][C::NT];

  Tile cur = tile_from(blockIdx.x);
  if (cur.idx >= total) return;
  Tile nxt = tile_from(cur.idx + gridDim.x);
  Tile done = cur;  // the tile whose accumulators are waiting for their epilogue

  // ---- prologue: biases, the first sub-chunk's rows, the first three weight images ---------------------------------------------
  for (int i = tid; i < a.c_out && i < C::BIAS_FLOATS; i += C::NTHREADS) Bs[i] = a.bias[i];
  for (int p = wave; p < C::X_PIECES; p += C::NWAVES) issue_x(cur, 0, 0, p);
#pragma unroll
  for (int ph = 0; ph < 3; ++ph) issue_a(cur, ph, ph, wave);
  pp_wait_vmcnt<0>();
  if (pre != 1.f) {
    for (int p = wave; p < C::X_PIECES; p += C::NWAVES) {
      bf16x8 v = *act_ptr(0, p);
      act_half(v, 0);
      act_half(v, 4);
      *act_ptr(0, p) = v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stamp();

  // fragment reads of phase (c, tap): 4 weight + 2 activation fragments per k-step of 16
  auto load_frags = [&](int c, int tap, int q) __attribute__((always_inline)) {
    const char* Ab = As + (q & (C::A_RING - 1)) * C::A_BYTES;
    const char* Xb = Xs + (c & 1) * C::X_BYTES;
    int rb = row_b0;
    asm volatile("" : "+v"(rb));  // same: the taps' fragment offsets are cheaper to recompute than to keep
    rb += tap * a.dil;
    const int swz = (rb >> 2) & 3;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
        bfr[ks][nt] = *reinterpret_cast<const bf16x8*>(Xb + rb * 64 + (((2 * ks + h) ^ swz) << 4) + nt * (32 * 64));
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) af[ks][mt] = *reinterpret_cast<const bf16x8*>(Ab + off_a[ks] + mt * (32 * 64));
    }
  };
  // the 16 MFMAs of the fragments in registers; behind MFMA i the i-th slice of the phase's side work (requests, in-place
  // activation): each slice is a handful of instructions that issue in the shadow of the 32-cycle MFMA in front of it
  // (MI355X_MICROARCH.md: <= 5 fillers per MFMA slot are free).  sched_barrier pins the interleaving: left alone the compiler puts
  // the whole side work in front of the first MFMA, where it delays group 1's MFMAs and with them the interval.
  auto mma16 = [&](auto&& slice) __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if (!(C::VAR & 1)) __builtin_amdgcn_s_setprio(1);
    pp_static_for<16>([&](auto i_c) __attribute__((always_inline)) {
      constexpr int I = decltype(i_c)::value, ks = I / 8, mt = (I % 8) / 2, nt = I % 2;
      acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][mt], bfr[ks][nt], acc[mt][nt], 0, 0, 0);
      slice(i_c);
      __builtin_amdgcn_sched_barrier(0);
    });
    if (!(C::VAR & 1)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto epilogue = [&](const Tile& t) __attribute__((always_inline)) {
    if (C::VAR & 16) {
      float sacc = 0.f;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) sacc += acc[mt][nt][r];
      if (sacc == 12345.678f) a.out[0] = (bf16_t)sacc;  // keeps the accumulators (and the MFMAs) alive
      return;
    }
    const float scale = a.out_scale, post = a.post_slope;
    // vector (nt, channel offset co) of this lane lives at base[nt] + co; it exists if lo[nt] <= co <= hi[nt] (row inside the
    // tensor, flat index inside [0, out_limit): the polyphase placement of the transposed convolutions needs the second test)
    long long base[C::NT];
    int lo[C::NT], hi[C::NT];
    int el5 = l5, eh = h;
    asm volatile("" : "+v"(el5), "+v"(eh));  // the address arithmetic stays here (hoisted out of the tile loop it is spilled)
#pragma unroll
    for (int nt = 0; nt < C::NT; ++nt) {
      const int r = t.r0 + wave * (C::NT * 32) + nt * 32 + el5;
      base[nt] = (long long)r * a.out_row_stride + t.m0 + 8 * eh + a.out_shift;
      const long long l = -base[nt], u = a.out_limit - 8 - base[nt];
      lo[nt] = l < 0 ? 0 : (l > 4096 ? 4096 : (int)l);
      hi[nt] = r < a.n_rows ? (u > 4096 ? 4096 : (u < -1 ? -1 : (int)u)) : -1;
      base[nt] += t.ob;
    }
    auto body = [&](auto has_res, auto has_acc) __attribute__((always_inline)) {
      constexpr bool RES = decltype(has_res)::value, ACC = decltype(has_acc)::value;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        u32x4 rv[C::NT][2], pv[C::NT][2];
        if (RES || ACC) {
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
              const int co = mt * 32 + 16 * p2;
              const long long f = (co >= lo[nt] && co <= hi[nt]) ? base[nt] + co : t.ob;
              if (RES) rv[nt][p2] = *reinterpret_cast<const u32x4*>(a.res + f);
              if (ACC) pv[nt][p2] = *reinterpret_cast<const u32x4*>(a.out + f);
            }
        }
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            float f[8];  // quads 2 p2 and 2 p2 + 1 of this lane
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc[mt][nt][8 * p2 + e];
            if (RES) {
              const u32x4 d = swap_quads_bf16(rv[nt][p2]);
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                f[2 * w] += bf16_lo(d[w]);
                f[2 * w + 1] += bf16_hi(d[w]);
              }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] *= scale;
            if (ACC) {
              const u32x4 d = swap_quads_bf16(pv[nt][p2]);
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                f[2 * w] += bf16_lo(d[w]);
                f[2 * w + 1] += bf16_hi(d[w]);
              }
            }
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float lo_ = post != 1.f ? fmaxf(f[2 * w], f[2 * w] * post) : f[2 * w];
              const float hi_ = post != 1.f ? fmaxf(f[2 * w + 1], f[2 * w + 1] * post) : f[2 * w + 1];
              o[w] = pack_bf16x2(lo_, hi_);
            }
            o = swap_quads_bf16(o);
            const int co = mt * 32 + 16 * p2;
            if (co >= lo[nt] && co <= hi[nt]) *reinterpret_cast<u32x4*>(a.out + base[nt] + co) = o;
          }
      }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    if (a.res) {
      if (a.accumulate) body(T_{}, T_{});
      else body(T_{}, F_{});
    } else if (a.accumulate) body(F_{}, T_{});
    else body(F_{}, F_{});
  };
  auto init_acc = [&](const Tile& t) __attribute__((always_inline)) {  // accumulators start at the bias
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + t.m0 + mt * 32 + 8 * q4 + 4 * h);
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[mt][nt][4 * q4 + i] = bv[i];
      }
  };
  // End of an interval of this wave.  Its requests have landed except the youngest `nfly` (group 0: the two weight pieces of this
  // interval's side work; group 1: the activation pieces of its last XD side works); its LDS reads and writes are complete.
  auto end_interval = [&](auto fly0, auto fly1, bool counted) __attribute__((always_inline)) {
    if (!counted) pp_wait_vmcnt<0>();
    else if (grp == 0) pp_wait_vmcnt<decltype(fly0)::value>();
    else pp_wait_vmcnt<decltype(fly1)::value>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    stamp();
  };

  bool first_tile = true;
  while (true) {
    const bool has_next = nxt.idx < total;
#pragma unroll 1
    for (int c = 0; c < C::NSUB; ++c) {
      const bool last_sub = c == C::NSUB - 1;
      const bool more_x = !last_sub || has_next;  // there is a sub-chunk behind this one to request rows for
      const bool counted = more_x && !(C::VAR & (32 | 64));
      pp_static_for<C::KS>([&](auto tap_c) __attribute__((always_inline)) {
        constexpr int TAP = decltype(tap_c)::value;
        const int q = c * C::KS + TAP;
        constexpr int NX = C::xcnt(TAP);                 // requests of this phase's side work (per wave of group 1)
        constexpr int TR = TAP - C::XD - 1;              // the tap whose requests this phase's side work activates
        constexpr int NR = C::xcnt(TR);
        if (TAP == 0 && c == 0) {
          if (!first_tile) epilogue(done);
          init_acc(cur);
        }
        load_frags(c, TAP, q);
        // group 1's barrier: its last side work was the one of phase q - 1
        if (late == 1) end_interval(std::integral_constant<int, 2>{}, std::integral_constant<int, C::xfly(TAP - 1)>{}, counted);
        // group 0: the two pieces (wave, wave + 4) of the weight image of phase q + 3, requested while its fragment reads are in
        // flight and group 1 has the matrix pipe to itself (an LDS-DMA request holds its wave for 100-185 cycles: among group 0's
        // own MFMAs that is time the pipe idles, because group 1 has finished its segment by then)
        if (grp == 0) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const int piece = wave + 4 * i;
            if (TAP + 3 < C::KS || !last_sub) issue_a(cur, q + 3, (q + 3) & (C::A_RING - 1), piece);
            else if (has_next) issue_a(nxt, TAP + 3 - C::KS, (q + 3) & (C::A_RING - 1), piece);
          }
        }
        const bool act = NR > 0 && grp == 1 && more_x && pre != 1.f && !(C::VAR & 32);
        const bool req = NX > 0 && grp == 1 && more_x && !(C::VAR & 32);
        const int xbuf = (c + 1) & 1;
        bf16x8 av[NR > 0 ? NR : 1];
        mma16([&](auto i_c) __attribute__((always_inline)) {
          constexpr int I = decltype(i_c)::value;
          // group 1: requests of the next sub-chunk's pieces (piece index = (tap * NXW + j) * 4 + wave - 4), late in its MFMA
          // segment: group 0 is multiplying by then and covers the request's issue time ...
          if (I >= 12 && I <= 14 && I - 12 < NX && req) {
            const int p = (TAP * C::NXW + (I - 12)) * 4 + (wave - 4);
            if (!last_sub) issue_x(cur, c + 1, xbuf, p);
            else issue_x(nxt, 0, 0, p);
          }
          // ... and the in-place activation of the pieces requested XD + 1 phases ago: read | two halves | write per piece
          if (I == 1 && act) {
#pragma unroll
            for (int j = 0; j < NR; ++j) av[j] = *act_ptr(xbuf, ((TR < 0 ? 0 : TR) * C::NXW + j) * 4 + (wave - 4));
          }
          if (I >= 3 && I < 3 + 3 * 3 && act) {
            constexpr int J = (I - 3) / 3, ST = (I - 3) % 3;
            if (J < NR) {
              if (ST < 2) act_half(av[J < NR ? J : 0], 4 * ST);
              else *act_ptr(xbuf, ((TR < 0 ? 0 : TR) * C::NXW + J) * 4 + (wave - 4)) = av[J < NR ? J : 0];
            }
          }
        });
        stamp();
        if (late == 0) end_interval(std::integral_constant<int, 2>{}, std::integral_constant<int, C::xfly(TAP)>{}, counted);
      });
    }
    done = cur;
    first_tile = false;
    if (!has_next) break;
    cur = nxt;
    nxt = tile_from(cur.idx + gridDim.x);
  }
  epilogue(done);
}

template <class C>
static ConvTcLaunch make_conv_pp_launch(const char* name) {
  ConvTcLaunch l;
  l.kernel = conv_tc_pp_kernel<C>;
  l.bm = C::BM;
  l.bn = C::BN;
  l.kc = C::KC;
  l.threads = C::NTHREADS;
  l.lds_bytes = C::LDS;
  l.name = name;
  l.wlayout = 3;
  l.persistent = 1;
  return l;
}

}  // namespace evmi
