// Device template of the MFMA implicit-GEMM convolution (included by conv_tc_mfma.hip, which
// instantiates the product table, and by bench_kernels.hip, which instantiates tuning variants).
#pragma once

#include <type_traits>

#include "conv_tc_mfma.h"

namespace evmi {

// ABL: ablation bits for the micro-benchmark only (tools/sweep_conv.py); 0 in the product table.
//   1 = no weight streaming after step 0   2 = activation tile loaded for chunk 0 only
//   4 = no epilogue                        8 = no MFMA (data movement only)
//   128 = s_memtime stamps of one wave into args.timeline (start, X tile in LDS, per step: barrier passed / MFMAs issued, ..., stores issued)
//   64 = activation tile of the NEXT channel chunk prefetched into registers during the current chunk's taps (round 2: measured
//        0.665 vs 0.600 ms on c128/k11 -- 20 more live VGPRs and the extra address arithmetic cost more than the exposed load)
template <int CIN_, int KC_, int BM_, int BN_, int WM_, int WN_, int KS_, int TAPS_, int MAXDIL_, int ABL_ = 0, int OCC_ = 0>
struct ConvTcCfg {
  static constexpr int CIN = CIN_, KC = KC_, BM = BM_, BN = BN_, WM = WM_, WN = WN_, KS = KS_,
                       TAPS = TAPS_, MAXDIL = MAXDIL_, ABL = ABL_;
  static constexpr int NTHREADS = WM * WN * 64;
  static constexpr int MT = BM / (WM * 32), NT = BN / (WN * 32);
  static constexpr int XS = KC + 8, AS = KC + 8, OS = BM + 8;
  static constexpr int R_MAX = BN + (KS - 1) * MAXDIL;
  static constexpr int NCHUNK = CIN / KC;
  static constexpr int NGROUP = (KS + TAPS - 1) / TAPS;
  static constexpr int NSTEP = NCHUNK * NGROUP;
  static constexpr int NABUF = NSTEP > 1 ? 2 : 1;
  static constexpr int A_TILE = TAPS * BM * AS;       // LDS elements per buffer
  static constexpr int A_VECS = TAPS * BM * (KC / 8); // 16-B vectors per tap group
  static constexpr int A_PER_THREAD = (A_VECS + NTHREADS - 1) / NTHREADS;
  // every tap group is full and divides evenly over the threads: no per-vector guards needed
  static constexpr bool A_EXACT = (KS % TAPS == 0) && (A_VECS % NTHREADS == 0);
  // activation tile of the NEXT chunk kept in flight in registers while the current chunk's taps run
  static constexpr int X_VECS_MAX = R_MAX * (KC / 8);
  static constexpr int X_PER_THREAD = (X_VECS_MAX + NTHREADS - 1) / NTHREADS;
  static constexpr bool X_PREFETCH = (ABL & 64) != 0;
  static constexpr size_t LDS_MAIN = size_t(R_MAX * XS + NABUF * A_TILE) * 2;
  static constexpr size_t LDS_OUT = size_t(BN) * OS * 2;
  static constexpr size_t LDS = LDS_MAIN > LDS_OUT ? LDS_MAIN : LDS_OUT;
  // waves per SIMD the register allocator must leave room for.  Default: as many workgroups per CU as the LDS admits (up to 8
  // waves per SIMD) -- two co-resident 8-wave workgroups (<= 128 VGPRs) are what hides one workgroup's tile load / epilogue
  // behind the other's MFMAs (s_memtime timeline, tools/conv_timeline.py).
  static constexpr int WG_PER_CU_LDS = int((160 * 1024) / LDS);
  static constexpr int OCC_AUTO = (WM_ * WN_ + 3) / 4 * (WG_PER_CU_LDS > 2 ? 2 : WG_PER_CU_LDS);
  static constexpr int OCC = OCC_ > 0 ? OCC_ : (OCC_AUTO > 8 ? 8 : OCC_AUTO);
  static_assert(CIN % KC == 0 && KC % 16 == 0, "channel chunking");
  static_assert(KS % TAPS == 0, "tap groups must be full");
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiling");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, C::OCC) void conv_tc_kernel(ConvTcArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);
  bf16_t* As = Xs + C::R_MAX * C::XS;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int r0 = blockIdx.x * C::BN;
  const int b = blockIdx.y;
  const int mtile = blockIdx.z;
  const int m0 = mtile * C::BM;

  const bf16_t* __restrict__ xb = a.x + (long long)b * a.x_batch_stride;
  // weights pre-laid-out by the host as [mtile][chunk][tap][BM][KC] (see relayout_conv_tc_weights)
  const bf16_t* __restrict__ wb = a.w + (long long)mtile * C::NCHUNK * C::KS * C::BM * C::KC;

  f32x16 acc[C::MT][C::NT];
#pragma unroll
  for (int i = 0; i < C::MT; ++i)
#pragma unroll
    for (int j = 0; j < C::NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  bf16x8 areg[C::A_PER_THREAD];
  int n_stamp = 0;
  auto stamp = [&]() {
    if ((C::ABL & 128) && a.timeline && blockIdx.x == 5 && blockIdx.y == 1 && blockIdx.z == 0 && (tid & 63) == 0 && n_stamp < 120)
      a.timeline[(tid >> 6) * 128 + n_stamp++] = (long long)__builtin_readcyclecounter();
  };
  stamp();

  auto a_prefetch = [&](int step) {
    const int chunk = step / C::NGROUP, grp = step % C::NGROUP;
    const bf16_t* src = wb + ((long long)chunk * C::KS + grp * C::TAPS) * C::BM * C::KC;
    const int ntaps = (C::KS - grp * C::TAPS) < C::TAPS ? (C::KS - grp * C::TAPS) : C::TAPS;
    const int nvec = ntaps * C::BM * (C::KC / 8);
#pragma unroll
    for (int i = 0; i < C::A_PER_THREAD; ++i) {
      const int v = tid + i * C::NTHREADS;
      if (C::A_EXACT || v < nvec) areg[i] = *reinterpret_cast<const bf16x8*>(src + (long long)v * 8);
    }
  };
  auto a_commit = [&](int step) {
    bf16_t* dst = As + (step & (C::NABUF - 1)) * C::A_TILE;
    const int grp = step % C::NGROUP;
    const int ntaps = (C::KS - grp * C::TAPS) < C::TAPS ? (C::KS - grp * C::TAPS) : C::TAPS;
    const int nvec = ntaps * C::BM * (C::KC / 8);
#pragma unroll
    for (int i = 0; i < C::A_PER_THREAD; ++i) {
      const int v = tid + i * C::NTHREADS;
      if (C::A_EXACT || v < nvec) {
        const int row = v / (C::KC / 8);  // tap*BM + m
        const int c8 = v % (C::KC / 8);
        *reinterpret_cast<bf16x8*>(dst + row * C::AS + c8 * 8) = areg[i];
      }
    }
  };

  const int rows_needed = C::BN + (C::KS - 1) * a.dil;
  const float pre = a.pre_slope;

  // Variant (ABL bit 64): the activation rows of chunk c + 1 are requested (global -> registers) right after chunk c's first
  // barrier and written to LDS at the chunk boundary, so their latency runs under chunk c's tap steps.  Not in the product
  // table: slower than the synchronous load on every shape measured (see the bit's description above).
  bf16x8 xreg[C::X_PER_THREAD];
  const int x_nvec = rows_needed * (C::KC / 8);
  auto x_prefetch = [&](int chunk) {
#pragma unroll
    for (int i = 0; i < C::X_PER_THREAD; ++i) {
      const int v = tid + i * C::NTHREADS;
      const int row = v / (C::KC / 8), c8 = v % (C::KC / 8);
      const int rr = r0 - a.pad + row;
      bf16x8 val;
#pragma unroll
      for (int e = 0; e < 8; ++e) val[e] = (bf16_t)0.f;
      if (v < x_nvec && rr >= 0 && rr < a.t_in) val = *reinterpret_cast<const bf16x8*>(xb + (long long)rr * C::CIN + chunk * C::KC + c8 * 8);
      xreg[i] = val;
    }
  };
  auto x_commit = [&]() {
#pragma unroll
    for (int i = 0; i < C::X_PER_THREAD; ++i) {
      const int v = tid + i * C::NTHREADS;
      if (v >= x_nvec) continue;
      const int row = v / (C::KC / 8), c8 = v % (C::KC / 8);
      bf16x8 val = xreg[i];
      if (pre != 1.f) {  // slope in [0, 1]: lrelu(x) = max(x, slope * x)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = (float)val[e];
          val[e] = (bf16_t)fmaxf(f, f * pre);
        }
      }
      *reinterpret_cast<bf16x8*>(Xs + row * C::XS + c8 * 8) = val;
    }
  };

  a_prefetch(0);
  if (C::X_PREFETCH) x_prefetch(0);
#pragma unroll 1
  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
    if (chunk > 0) __syncthreads();  // everyone is done reading the previous X chunk
    // ---- activation tile: rows [r0 - pad, r0 - pad + rows_needed) x channels [chunk*KC, +KC)
    if (C::X_PREFETCH) {
      x_commit();
    } else if (!(C::ABL & 2) || chunk == 0) {
      // every row segment of the tile is requested before the first one is consumed: ONE memory round trip per chunk
      // (a loop that loads, activates and stores vector by vector pays one round trip per iteration: measured 9.6k + 6.5k
      // cycles for the two 39 KB tiles of a c128 block, s_memtime stamps of tools/conv_timeline.py)
      bf16x8 xv[C::X_PER_THREAD];
#pragma unroll
      for (int i = 0; i < C::X_PER_THREAD; ++i) {
        const int v = tid + i * C::NTHREADS;
        const int row = v / (C::KC / 8), c8 = v % (C::KC / 8);
        const int rr = r0 - a.pad + row;
        bf16x8 val;
#pragma unroll
        for (int e = 0; e < 8; ++e) val[e] = (bf16_t)0.f;
        if (v < x_nvec && rr >= 0 && rr < a.t_in) val = *reinterpret_cast<const bf16x8*>(xb + (long long)rr * C::CIN + chunk * C::KC + c8 * 8);
        xv[i] = val;
      }
#pragma unroll
      for (int i = 0; i < C::X_PER_THREAD; ++i) {
        const int v = tid + i * C::NTHREADS;
        if (v >= x_nvec) continue;
        const int row = v / (C::KC / 8), c8 = v % (C::KC / 8);
        bf16x8 val = xv[i];
        if (pre != 1.f) {  // slope in [0, 1]: lrelu(x) = max(x, slope * x)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)val[e];
            val[e] = (bf16_t)fmaxf(f, f * pre);
          }
        }
        *reinterpret_cast<bf16x8*>(Xs + row * C::XS + c8 * 8) = val;
      }
    }
    stamp();  // X tile written
#pragma unroll 1
    for (int grp = 0; grp < C::NGROUP; ++grp) {
      const int step = chunk * C::NGROUP + grp;
      if (!(C::ABL & 1) || step == 0) a_commit(step);
      __syncthreads();
      stamp();  // step barrier passed
      if (step + 1 < C::NSTEP && !(C::ABL & 1)) a_prefetch(step + 1);
      if (C::X_PREFETCH && grp == 0 && chunk + 1 < C::NCHUNK) x_prefetch(chunk + 1);
      const bf16_t* Ab = As + ((C::ABL & 1) ? 0 : (step & (C::NABUF - 1))) * C::A_TILE;
      {
        const bf16_t* Arow = Ab + (wm * C::MT * 32 + (lane & 31)) * C::AS + (lane >> 5) * 8;
        const bf16_t* Brow = Xs + (wn * C::NT * 32 + (lane & 31) + grp * C::TAPS * a.dil) * C::XS + (lane >> 5) * 8;
        if (!(C::ABL & 8))
          mma_tap_group<C::MT, C::NT, C::KC / 16, C::TAPS, C::BM * C::AS, 32 * C::AS, 32 * C::XS, (C::ABL >> 5) & 1>(
              Arow, Brow, a.dil * C::XS, acc);
        else
          acc[0][0][0] += (float)Arow[0] * (float)Brow[0];
      }
      stamp();  // MFMAs of the step issued
    }
  }

  // ---- epilogue: acc + bias -> bf16 -> LDS [BN][BM+8] -> coalesced fused store -----------------
  if (C::ABL & 4) {
    float sacc = 0.f;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc += acc[mt][nt][r];
    if (sacc == 12345.678f) a.out[0] = (bf16_t)sacc;  // keep the accumulators live
    return;
  }
  __syncthreads();
  stamp();  // main loop done everywhere
  bf16_t* Os = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = wm * C::MT * 32 + mt * 32 + 8 * q + 4 * (lane >> 5);
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + m0 + c);
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) {
        const int n = wn * C::NT * 32 + nt * 32 + (lane & 31);
        bf16x4 pk;
#pragma unroll
        for (int i = 0; i < 4; ++i) pk[i] = (bf16_t)(acc[mt][nt][4 * q + i] + bv[i]);
        *reinterpret_cast<bf16x4*>(Os + n * C::OS + c) = pk;
      }
    }
  }
  __syncthreads();
  stamp();  // staged
  {
    const long long ob = (long long)b * a.out_batch_stride;
    const float scale = a.out_scale, post = a.post_slope;
    constexpr int VPR = C::BM / 8;
    constexpr int OPT = (C::BN * VPR + C::NTHREADS - 1) / C::NTHREADS;  // output vectors per thread
    constexpr int EB = OPT >= 4 ? 4 : OPT;                                // vectors per batch (register budget: 2 x 4 x EB)
    // residual / running-sum rows are requested for a whole batch of the thread's vectors before the first is consumed:
    // one memory round trip per batch instead of one per vector (measured 13.5k cycles for 8 dependent trips on a c128 block)
    auto flat_index = [&](int v) -> long long {
      const int n = v / VPR, c8 = v % VPR;
      const int r = r0 + n;
      const long long flat = (long long)r * a.out_row_stride + m0 + c8 * 8 + a.out_shift;
      return (v >= C::BN * VPR || r >= a.n_rows || flat < 0 || flat >= a.out_limit) ? -1 : flat;
    };
    // the (residual, running sum) cases are separate straight-line bodies: with the conditions inside, the compiler joins the
    // paths with s_waitcnt vmcnt(0) after every store
    const float mslope = a.mask_slope;
    auto body = [&](auto has_res, auto has_acc, auto has_mask) {
      constexpr bool RES = decltype(has_res)::value, ACC = decltype(has_acc)::value, MSK = decltype(has_mask)::value;
#pragma unroll
      for (int i0 = 0; i0 < OPT; i0 += EB) {
        bf16x8 rv[EB], pv[EB], mv[EB];
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          // unconditional loads (vectors that are not stored read element 0 of the item): a load under a per-lane condition
          // makes the compiler branch around it and drain the memory counter at every join
          const long long flat = flat_index(tid + (i0 + i) * C::NTHREADS);
          const long long safe = flat < 0 ? 0 : flat;
          if (RES && i0 + i < OPT) rv[i] = *reinterpret_cast<const bf16x8*>(a.res + ob + safe);
          if (ACC && i0 + i < OPT) pv[i] = *reinterpret_cast<const bf16x8*>(a.out + ob + safe);
          if (MSK && i0 + i < OPT) mv[i] = *reinterpret_cast<const bf16x8*>(a.mask + ob + safe);
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          const int v = tid + (i0 + i) * C::NTHREADS;
          const long long flat = flat_index(v);
          if (i0 + i >= OPT) continue;
          const int n = (v / VPR) % C::BN, c8 = v % VPR;
          const bf16x8 o = *reinterpret_cast<const bf16x8*>(Os + n * C::OS + c8 * 8);
          float f[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = (float)o[e];
          if (MSK) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = (float)mv[i][e] > 0.f ? f[e] : f[e] * mslope;
          }
          if (RES) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)rv[i][e];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] *= scale;
          if (ACC) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] += (float)pv[i][e];
          }
          bf16x8 res;
#pragma unroll
          for (int e = 0; e < 8; ++e) res[e] = (bf16_t)(post != 1.f ? lrelu(f[e], post) : f[e]);
          if (flat >= 0) *reinterpret_cast<bf16x8*>(a.out + ob + flat) = res;
        }
      }
    };
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    if (a.mask) {  // (training launches: never with a running sum)
      if (a.res) body(T_{}, F_{}, T_{});
      else body(F_{}, F_{}, T_{});
    } else if (a.res) {
      if (a.accumulate) body(T_{}, T_{}, F_{});
      else body(T_{}, F_{}, F_{});
    } else {
      if (a.accumulate) body(F_{}, T_{}, F_{});
      else body(F_{}, F_{}, F_{});
    }
  }
  stamp();  // stores issued
}


template <class C>
static ConvTcLaunch make_conv_tc_launch(const char* name) {
  ConvTcLaunch l;
  l.kernel = conv_tc_kernel<C>;
  l.bm = C::BM;
  l.bn = C::BN;
  l.kc = C::KC;
  l.threads = C::NTHREADS;
  l.lds_bytes = C::LDS;
  l.name = name;
  return l;
}

}  // namespace evmi
