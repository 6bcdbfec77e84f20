// MFMA implicit-GEMM convolution for the wide layers (C_in a multiple of 64, C_out a multiple of 128), LDS filled by
// LDS-DMA (global_load_lds_dwordx4) instead of global -> registers -> ds_write.  Same contract and arguments as
// conv_tc_kernel (conv_tc_mfma.h); what differs is how the operands reach the LDS and how they are laid out there:
//
//   * LDS images are UNPADDED rows of 64 channels (128 B = eight 16-byte slots).  An LDS-DMA instruction writes
//     base + lane * 16, so a padded row stride cannot be produced; bank conflicts of the ds_read_b128 fragment reads are
//     avoided by an XOR swizzle instead: slot p of row r holds channel vector p ^ ((r >> 1) & 7).  The 16-lane groups of
//     ds_read_b128 read 16 consecutive rows (mod 16) at one channel vector: 16 distinct 16-byte slots of the 256-byte bank row.
//     The permutation is applied on the SOURCE side (each lane picks the global vector that belongs in its slot; the weight
//     images are stored pre-swizzled by the host) and again on the fragment read.
//   * weights: one 16 KB image per (channel chunk, tap), double buffered; the image of step s + 1 is requested right after
//     the barrier that opens step s and has a whole step of MFMAs to land: no staging registers, no ds_write pass.
//   * activations: the (BN + (KS-1)*dil) x 64 tile of a channel chunk is requested in 8-row pieces; rows outside the
//     sequence read a zero line.  A leaky-ReLU on load (pre_slope != 1) is applied by one pass over the landed tile.
//   * 512 threads, 128 channels x 256 rows per workgroup, 64 x 64 per wave; <= 128 VGPRs and ~71 KB of LDS so that TWO
//     workgroups share a CU: one's tile load / epilogue runs under the other's MFMAs (tools/conv_timeline.py).
#pragma once

#include <type_traits>

#include "conv_tc_mfma.h"

namespace evmi {

static __device__ __attribute__((aligned(128))) bf16_t g_conv_dma_zero_row[64];  // zero-initialised: source of out-of-range rows

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
__device__ __forceinline__ void lds_dma_b128(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((glb_void_t*)g, (lds_void_t*)l, 16, 0, 0);
}

// DBG: s_memtime stamps of workgroup (5, 1, 0) into args.timeline (tools/conv_timeline.py)
// VAR: schedule variants for A/B runs (tools/sweep_conv.py):
//   1 = static priority 1 for the younger half of the workgroup (waves 4-7 lose the age-based issue arbitration otherwise)
//   2 = fragment reads of k-step k + 1 pinned in front of the MFMAs of k-step k (two fragment register sets)
//   4 = s_setprio 1 around every MFMA cluster
//   ablations (results are wrong; timing only): 16 = no step barriers and no weight requests after step 0 (fragment reads +
//   MFMAs on whatever the LDS holds); 32 = additionally no fragment reads inside the k-loop (MFMAs on the first fragments);
//   64 = activation tile requested for chunk 0 only; 128 = no epilogue (accumulators kept live); 256 = no step barriers
//   (weight requests kept); 512 = no weight requests after step 0 (barriers kept)
// WN_ = 8: sixteen waves on 512 rows, ONE workgroup per CU (every weight image feeds twice the MFMAs); A/B variant
// WN_ = 2: four waves on 128 rows -- the NARROW tile for problems too small to fill the chip with 256-row tiles (one utterance; the GAN
// step's generator at 16 x 256 .. 2048 rows: 32 .. 128 workgroups of 256 rows on 256 CUs); same K order per output: same bits
template <int CIN_, int KS_, int MAXDIL_, int DBG_ = 0, int VAR_ = 0, int WN_ = 4>
struct ConvDmaCfg {
  static constexpr int CIN = CIN_, KS = KS_, MAXDIL = MAXDIL_, DBG = DBG_, VAR = VAR_;
  static constexpr int KC = 64, BM = 128, WM = 2, WN = WN_, MT = 2, NT = 2, BN = WN * NT * 32;
  static constexpr int NTHREADS = WM * WN * 64, NWAVES = WM * WN;
  static constexpr int NCHUNK = CIN / KC, NSTEP = NCHUNK * KS;
  static constexpr int R_MAX = BN + (KS - 1) * MAXDIL;
  static constexpr int X_PIECES = (R_MAX + 7) / 8;  // 8 rows x 128 B = 1 KiB = one wave-instruction
  static constexpr int X_BYTES = X_PIECES * 1024;
  static constexpr int A_PIECES = BM / 8;
  static constexpr int A_BYTES = A_PIECES * 1024;
  static constexpr int OS = BM + 8;
  static constexpr size_t LDS_MAIN = size_t(X_BYTES) + 2 * A_BYTES;
  static constexpr size_t LDS_OUT = size_t(BN) * OS * 2;  // staging of the residual layers' epilogue
  static constexpr size_t LDS = LDS_MAIN > LDS_OUT ? LDS_MAIN : LDS_OUT;
  static constexpr int X_VEC_PER_THREAD = (X_PIECES * 64 + NTHREADS - 1) / NTHREADS;
  static_assert(CIN % KC == 0, "channel chunking");
  static_assert((WN == 4 ? 2 : 1) * LDS <= 160 * 1024, "two workgroups per CU (one for the 16-wave form)");
  static_assert(A_PIECES % NWAVES == 0 || NWAVES == 16, "weight pieces per wave");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS, 4) void conv_tc_dma_kernel(ConvTcArgs a) {
  static_assert(C::NWAVES == 2 || C::NWAVES == 4 || C::NWAVES == 8 || C::NWAVES == 16, "2, 4, 8 or 16 waves");
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  char* Xs = smem;
  char* As = smem + C::X_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / C::WN, wn = wave % C::WN;
  const int r0 = blockIdx.x * C::BN;
  const int b = blockIdx.y;
  const int mtile = blockIdx.z;
  const int m0 = mtile * C::BM;

  const bf16_t* __restrict__ xb = a.x + (long long)b * a.x_batch_stride;
  // pre-swizzled LDS images [mtile][chunk][tap][BM rows][8 slots] (relayout_conv, layout 1)
  const bf16_t* __restrict__ wb = a.w + (long long)mtile * C::NSTEP * C::BM * C::KC;

  int n_stamp = 0;
  auto stamp = [&]() {
    if (C::DBG && a.timeline && blockIdx.x == 5 && blockIdx.y == 1 && blockIdx.z == 0 && lane == 0 && n_stamp < 120)
      a.timeline[wave * 128 + n_stamp++] = (long long)__builtin_readcyclecounter();
  };
  stamp();

  // ---- output geometry of this lane ---------------------------------------------------------------------------------
  // After a 32 x 32 MFMA chain lane (n, h) holds, for every register quad q, channels 8q + 4h .. + 3 of row n.  One
  // v_permlane32_swap per dword between quads 2p and 2p + 1 leaves lane (n, 0) with channels 16p .. 16p + 7 and lane (n, 1) with
  // 16p + 8 .. 16p + 15 of its row: 16 bytes per lane, stored (and, for the residual / running sum, loaded) straight from / to
  // registers -- no LDS staging pass, no barrier, every wave retires its tile on its own while the others keep the MFMAs going.
  const int h = lane >> 5;
  const long long ob = (long long)b * a.out_batch_stride;
  // flat index of (row of n-tile nt, channel m0 + wm*64 + 8h); which of the row's vectors are stored is decided per vector:
  // in the polyphase (transposed convolution) placement the channels of one GEMM row are different output rows
  long long flat_nt[C::NT];
  bool row_ok[C::NT];
#pragma unroll
  for (int nt = 0; nt < C::NT; ++nt) {
    const int r = r0 + wn * (C::NT * 32) + nt * 32 + (lane & 31);
    flat_nt[nt] = (long long)r * a.out_row_stride + m0 + wm * (C::MT * 32) + 8 * h + a.out_shift;
    row_ok[nt] = r < a.n_rows;
  }
  // vector (nt, channel offset co): its flat index, or -1 (offsets and limits are multiples of 8: a vector never straddles)
  auto vec_index = [&](int nt, int co) -> long long {
    const long long f = flat_nt[nt] + co;
    return (row_ok[nt] && f >= 0 && f + 8 <= a.out_limit) ? f : -1;
  };
  auto swap_quads = [](u32x4 d) -> u32x4 { return swap_quads_bf16(d); };

  // accumulators start at the bias: the direct epilogue only scales, activates and stores
  f32x16 acc[C::MT][C::NT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const f32x4 bv = *reinterpret_cast<const f32x4*>(a.bias + m0 + wm * (C::MT * 32) + mt * 32 + 8 * q + 4 * h);
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[mt][nt][4 * q + i] = bv[i];
    }
  const int rows_needed = C::BN + (C::KS - 1) * a.dil;
  const int x_pieces = (rows_needed + 7) >> 3;
  const float pre = a.pre_slope;

  // ---- LDS-DMA requests ------------------------------------------------------------------------------------------
  auto issue_a = [&](int step) {  // weight image of `step` -> slot step & 1: 16 pieces
    const bf16_t* src = wb + (long long)step * (C::BM * C::KC) + lane * 8;
    char* dst = As + (step & 1) * C::A_BYTES;
    if (C::VAR & 1024) {
      // only the older half of the workgroup requests (four pieces per wave): waves 0-3 win the issue arbitration, finish their
      // MFMAs 600-900 cycles before waves 4-7 and would wait at the barrier anyway; an LDS-DMA instruction holds its wave
      // for 100-185 cycles, which the younger waves -- the ones that set the step time -- no longer pay
      if (wave < C::NWAVES / 2) {
#pragma unroll
        for (int i = 0; i < 2 * C::A_PIECES / C::NWAVES; ++i) {
          const int p = wave + i * (C::NWAVES / 2);
          lds_dma_b128(src + p * 512, dst + p * 1024);
        }
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < (C::A_PIECES + C::NWAVES - 1) / C::NWAVES; ++i) {
      const int p = wave + i * C::NWAVES;
      if (p < C::A_PIECES) lds_dma_b128(src + p * 512, dst + p * 1024);
    }
  };
  auto issue_x = [&](int chunk) {  // activation rows [r0 - pad, + rows_needed) x channels [chunk * 64, + 64), 8-row pieces
    const int slot = lane & 7;
    for (int p = wave; p < x_pieces; p += C::NWAVES) {
      const int row = p * 8 + (lane >> 3);
      const int rr = r0 - a.pad + row;
      const int c8 = slot ^ ((row >> 1) & 7);
      const bf16_t* src = (row < rows_needed && rr >= 0 && rr < a.t_in) ? xb + (long long)rr * C::CIN + chunk * C::KC + c8 * 8
                                                                        : g_conv_dma_zero_row + slot * 8;
      lds_dma_b128(src, Xs + p * 1024);
    }
  };

  // ---- fragment addresses (bytes): row * 128 + ((channel vector ^ ((row >> 1) & 7)) << 4) --------------------------
  const int row_a = wm * (C::MT * 32) + (lane & 31);
  const int s_a = (row_a >> 1) & 7;
  int off_a[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) off_a[ks] = row_a * 128 + (((2 * ks + h) ^ s_a) << 4);

  auto mma_step = [&](const char* Ab, int tap, int next_step) {
    const int q = wn * (C::NT * 32) + (lane & 31) + tap * a.dil;  // this lane's activation row (n-tile 0)
    const int s_b = (q >> 1) & 7;
    int off_b[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) off_b[ks] = q * 128 + (((2 * ks + h) ^ s_b) << 4);
    bf16x8 af[2][C::MT], bfr[2][C::NT];
    auto load = [&](int ks, int buf) {
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) af[buf][mt] = *reinterpret_cast<const bf16x8*>(Ab + off_a[ks] + mt * (32 * 128));
#pragma unroll
      for (int nt = 0; nt < C::NT; ++nt) bfr[buf][nt] = *reinterpret_cast<const bf16x8*>(Xs + off_b[ks] + nt * (32 * 128));
    };
    load(0, 0);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = (C::VAR & 32) ? 0 : (ks & 1);
      if (ks + 1 < 4 && !(C::VAR & 32)) load(ks + 1, cur ^ 1);
      if ((C::VAR & 2048) && ks == 0 && next_step >= 0) issue_a(next_step);  // requests among the MFMAs instead of behind the barrier
      if (C::VAR & 2) __builtin_amdgcn_sched_barrier(0);
      if (C::VAR & 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][mt], bfr[cur][nt], acc[mt][nt], 0, 0, 0);
      if (C::VAR & 4) __builtin_amdgcn_s_setprio(0);
      if (C::VAR & 2) __builtin_amdgcn_sched_barrier(0);
    }
  };
  if ((C::VAR & 1) && wave >= C::NWAVES / 2) __builtin_amdgcn_s_setprio(1);

  issue_x(0);
  issue_a(0);
#pragma unroll 1
  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
    if (chunk > 0 && !(C::VAR & 64)) {
      __syncthreads();  // every wave is done reading the previous chunk's rows
      issue_x(chunk);
    }
    if (pre != 1.f && !((C::VAR & 64) && chunk > 0)) {
      lds_dma_barrier();  // vmcnt(0) of every wave (explicit: common.h), then the barrier: the whole tile has landed
      // leaky-ReLU in place (slope in [0, 1]: lrelu(x) = max(x, slope * x)); the swizzle is a permutation inside a row
#pragma unroll
      for (int i = 0; i < C::X_VEC_PER_THREAD; ++i) {
        const int v = tid + i * C::NTHREADS;
        if (v < x_pieces * 64) {
          bf16x8 val = *reinterpret_cast<const bf16x8*>(Xs + v * 16);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)val[e];
            val[e] = (bf16_t)fmaxf(f, f * pre);
          }
          *reinterpret_cast<bf16x8*>(Xs + v * 16) = val;
        }
      }
    }
    stamp();  // tile requested / activated
#pragma unroll 1
    for (int tap = 0; tap < C::KS; ++tap) {
      const int step = chunk * C::KS + tap;
      if (!(C::VAR & (16 | 256)) || step == 0) lds_dma_barrier();  // weight image of this step (and, at tap 0, the tile) landed and visible; slot (step + 1) & 1 is free
      stamp();
      const bool more = step + 1 < C::NSTEP && !(C::VAR & (16 | 512));
      if (more && !(C::VAR & 2048)) issue_a(step + 1);
      mma_step(As + (step & 1) * C::A_BYTES, tap, (more && (C::VAR & 2048)) ? step + 1 : -1);
      stamp();
    }
  }

  if (C::VAR & 128) {
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
  stamp();
  if (a.res || a.mask) {
    // ---- residual (and training-mask) layers: acc -> bf16 -> LDS [BN][BM+8] -> full-row fused stores ----------------------
    // The residual rows have to be read in full 128-byte lines: fetched in the accumulator layout (32 rows x 32 B per
    // instruction) they cost 17-21k cycles per tile (measured), against 3-5k for the staged pass.
    __syncthreads();
    bf16_t* Os = reinterpret_cast<bf16_t*>(smem);
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = wm * C::MT * 32 + mt * 32 + 8 * q + 4 * h;
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt) {
          const int n = wn * C::NT * 32 + nt * 32 + (lane & 31);
          bf16x4 pk;
#pragma unroll
          for (int i = 0; i < 4; ++i) pk[i] = (bf16_t)acc[mt][nt][4 * q + i];
          *reinterpret_cast<bf16x4*>(Os + n * C::OS + c) = pk;
        }
      }
    __syncthreads();
    const float scale = a.out_scale, post = a.post_slope;
    constexpr int VPR = C::BM / 8;
    constexpr int OPT = C::BN * VPR / C::NTHREADS;  // output vectors per thread
    constexpr int EB = 4;                            // vectors per batch (register budget: 2 x 4 x EB)
    static_assert(C::BN * VPR % C::NTHREADS == 0 && OPT % EB == 0, "epilogue tiling");
    auto flat_index = [&](int v) -> long long {
      const int n = v / VPR, c8 = v % VPR;
      const int r = r0 + n;
      const long long flat = (long long)r * a.out_row_stride + m0 + c8 * 8 + a.out_shift;
      return (r >= a.n_rows || flat < 0 || flat >= a.out_limit) ? -1 : flat;
    };
    // residual / running-sum rows of a batch are requested before the first is consumed; unconditional loads and separate
    // straight-line bodies per case (conditions around the loads make the compiler drain the memory counter at every join)
    const float mslope = a.mask_slope;
    auto body = [&](auto has_acc, auto has_res, auto has_mask) {
      constexpr bool ACC = decltype(has_acc)::value, RES = decltype(has_res)::value, MSK = decltype(has_mask)::value;
#pragma unroll
      for (int i0 = 0; i0 < OPT; i0 += EB) {
        bf16x8 rv[EB], pv[EB], mv[EB];
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          const long long flat = flat_index(tid + (i0 + i) * C::NTHREADS);
          const long long safe = flat < 0 ? 0 : flat;
          if (RES) rv[i] = *reinterpret_cast<const bf16x8*>(a.res + ob + safe);
          if (ACC) pv[i] = *reinterpret_cast<const bf16x8*>(a.out + ob + safe);
          if (MSK) mv[i] = *reinterpret_cast<const bf16x8*>(a.mask + ob + safe);
        }
#pragma unroll
        for (int i = 0; i < EB; ++i) {
          const int v = tid + (i0 + i) * C::NTHREADS;
          const long long flat = flat_index(v);
          const int n = v / VPR, c8 = v % VPR;
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
      if (a.res) body(F_{}, T_{}, T_{});
      else body(F_{}, F_{}, T_{});
    } else if (a.accumulate) body(T_{}, T_{}, F_{});
    else body(F_{}, T_{}, F_{});
  } else {
    // ---- epilogue: registers -> (x scale, + running sum, leaky-ReLU) -> bf16 -> 16-byte stores -------------------------
    const float scale = a.out_scale, post = a.post_slope;
    auto body = [&](auto has_acc) {
      constexpr bool ACC = decltype(has_acc)::value;
#pragma unroll
      for (int mt = 0; mt < C::MT; ++mt) {
        u32x4 pv[C::NT][2];
        if (ACC) {
#pragma unroll
          for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
            for (int p2 = 0; p2 < 2; ++p2) {
              const long long f = vec_index(nt, mt * 32 + 16 * p2);
              pv[nt][p2] = *reinterpret_cast<const u32x4*>(a.out + ob + (f < 0 ? 0 : f));
            }
        }
#pragma unroll
        for (int nt = 0; nt < C::NT; ++nt)
#pragma unroll
          for (int p2 = 0; p2 < 2; ++p2) {
            float f[8];  // quads 2p and 2p + 1 of this lane
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = acc[mt][nt][8 * p2 + e] * scale;
            if (ACC) {
              const u32x4 d = swap_quads(pv[nt][p2]);
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                f[2 * w] += __builtin_bit_cast(float, d[w] << 16);
                f[2 * w + 1] += __builtin_bit_cast(float, d[w] & 0xffff0000u);
              }
            }
            u32x4 o;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const float lo = post != 1.f ? lrelu(f[2 * w], post) : f[2 * w];
              const float hi = post != 1.f ? lrelu(f[2 * w + 1], post) : f[2 * w + 1];
              typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
              bf16x2_t pk;
              pk[0] = (bf16_t)lo;
              pk[1] = (bf16_t)hi;
              o[w] = __builtin_bit_cast(unsigned, pk);
            }
            o = swap_quads(o);
            const long long fo = vec_index(nt, mt * 32 + 16 * p2);
            if (fo >= 0) *reinterpret_cast<u32x4*>(a.out + ob + fo) = o;
          }
      }
    };
    if (a.accumulate) body(std::integral_constant<bool, true>{});
    else body(std::integral_constant<bool, false>{});
  }
  stamp();
}

template <class C>
static ConvTcLaunch make_conv_dma_launch(const char* name) {
  ConvTcLaunch l;
  l.kernel = conv_tc_dma_kernel<C>;
  l.bm = C::BM;
  l.bn = C::BN;
  l.kc = C::KC;
  l.threads = C::NTHREADS;
  l.lds_bytes = C::LDS;
  l.name = name;
  l.wlayout = 1;
  return l;
}

}  // namespace evmi
