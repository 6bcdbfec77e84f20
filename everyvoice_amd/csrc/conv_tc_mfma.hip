// MFMA implicit-GEMM 1-D convolution, gfx950.  See conv_tc_mfma.h for the contract.
//
// Tiling (per workgroup of WM*WN wavefronts of 64):
//   D[BM x BN] fp32 in AGPR/VGPR accumulators, v_mfma_f32_32x32x16_bf16; wave (wm, wn) owns
//   (MT*32) channels x (NT*32) rows.
//   A fragment (weights):  lane l -> row l&31 of the 32-channel tile, 8 consecutive c_in at
//                          (l>>5)*8 of the 16-deep k-step: one ds_read_b128 from As[tap][m][KC+8].
//   B fragment (activations): lane l -> output row l&31 (+ tap*dil), same 8 channels: one
//                          ds_read_b128 from Xs[row][KC+8].  The +8 element pad makes the row
//                          stride an odd multiple of 16 B, so the 16-lane groups of ds_read_b128
//                          hit 16 distinct 16-B slots: conflict-free.
//   D: lane l holds output row l&31, channels 8*(reg>>2) + 4*(l>>5) + (reg&3): four consecutive
//      channels per register quad -> packed to 4 x bf16 and staged through LDS so the global
//      stores (and the residual / accumulate loads) are full 16-B, row-contiguous accesses.
#include "conv_tc_mfma.h"

namespace evmi {

template <int CIN_, int KC_, int BM_, int BN_, int WM_, int WN_, int KS_, int TAPS_, int MAXDIL_>
struct ConvTcCfg {
  static constexpr int CIN = CIN_, KC = KC_, BM = BM_, BN = BN_, WM = WM_, WN = WN_, KS = KS_,
                       TAPS = TAPS_, MAXDIL = MAXDIL_;
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
  static constexpr size_t LDS_MAIN = size_t(R_MAX * XS + NABUF * A_TILE) * 2;
  static constexpr size_t LDS_OUT = size_t(BN) * OS * 2;
  static constexpr size_t LDS = LDS_MAIN > LDS_OUT ? LDS_MAIN : LDS_OUT;
  static_assert(CIN % KC == 0 && KC % 16 == 0, "channel chunking");
  static_assert(BM % (WM * 32) == 0 && BN % (WN * 32) == 0, "wave tiling");
  static_assert(LDS <= 160 * 1024, "LDS budget");
};

template <class C>
__global__ __launch_bounds__(C::NTHREADS) void conv_tc_kernel(ConvTcArgs a) {
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

  auto a_prefetch = [&](int step) {
    const int chunk = step / C::NGROUP, grp = step % C::NGROUP;
    const bf16_t* src = wb + ((long long)chunk * C::KS + grp * C::TAPS) * C::BM * C::KC;
    const int ntaps = (C::KS - grp * C::TAPS) < C::TAPS ? (C::KS - grp * C::TAPS) : C::TAPS;
    const int nvec = ntaps * C::BM * (C::KC / 8);
#pragma unroll
    for (int i = 0; i < C::A_PER_THREAD; ++i) {
      const int v = tid + i * C::NTHREADS;
      if (v < nvec) areg[i] = *reinterpret_cast<const bf16x8*>(src + (long long)v * 8);
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
      if (v < nvec) {
        const int row = v / (C::KC / 8);  // tap*BM + m
        const int c8 = v % (C::KC / 8);
        *reinterpret_cast<bf16x8*>(dst + row * C::AS + c8 * 8) = areg[i];
      }
    }
  };

  const int rows_needed = C::BN + (C::KS - 1) * a.dil;
  const float pre = a.pre_slope;

  a_prefetch(0);
#pragma unroll 1
  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
    if (chunk > 0) __syncthreads();  // everyone is done reading the previous X chunk
    // ---- activation tile: rows [r0 - pad, r0 - pad + rows_needed) x channels [chunk*KC, +KC)
    {
      const int nvec = rows_needed * (C::KC / 8);
      for (int v = tid; v < nvec; v += C::NTHREADS) {
        const int i = v / (C::KC / 8), c8 = v % (C::KC / 8);
        const int rr = r0 - a.pad + i;
        bf16x8 val;
        if (rr >= 0 && rr < a.t_in) {
          val = *reinterpret_cast<const bf16x8*>(xb + (long long)rr * C::CIN + chunk * C::KC + c8 * 8);
          if (pre != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) val[e] = (bf16_t)lrelu((float)val[e], pre);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) val[e] = (bf16_t)0.f;
        }
        *reinterpret_cast<bf16x8*>(Xs + i * C::XS + c8 * 8) = val;
      }
    }
#pragma unroll 1
    for (int grp = 0; grp < C::NGROUP; ++grp) {
      const int step = chunk * C::NGROUP + grp;
      a_commit(step);
      __syncthreads();
      if (step + 1 < C::NSTEP) a_prefetch(step + 1);
      const bf16_t* Ab = As + (step & (C::NABUF - 1)) * C::A_TILE;
#pragma unroll
      for (int jj = 0; jj < C::TAPS; ++jj) {
        const int j = grp * C::TAPS + jj;
        if (j < C::KS) {
          const bf16_t* Arow = Ab + (jj * C::BM + wm * C::MT * 32 + (lane & 31)) * C::AS + (lane >> 5) * 8;
          const bf16_t* Brow = Xs + (wn * C::NT * 32 + (lane & 31) + j * a.dil) * C::XS + (lane >> 5) * 8;
#pragma unroll
          for (int ks = 0; ks < C::KC / 16; ++ks) {
            bf16x8 af[C::MT], bfr[C::NT];
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
              af[mt] = *reinterpret_cast<const bf16x8*>(Arow + mt * 32 * C::AS + ks * 16);
#pragma unroll
            for (int nt = 0; nt < C::NT; ++nt)
              bfr[nt] = *reinterpret_cast<const bf16x8*>(Brow + nt * 32 * C::XS + ks * 16);
#pragma unroll
            for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < C::NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mt], bfr[nt], acc[mt][nt], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: acc + bias -> bf16 -> LDS [BN][BM+8] -> coalesced fused store -----------------
  __syncthreads();
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
  {
    const long long ob = (long long)b * a.out_batch_stride;
    const float scale = a.out_scale, post = a.post_slope;
    constexpr int VPR = C::BM / 8;
    for (int v = tid; v < C::BN * VPR; v += C::NTHREADS) {
      const int n = v / VPR, c8 = v % VPR;
      const int r = r0 + n;
      if (r >= a.n_rows) continue;
      const long long flat = (long long)r * a.out_row_stride + m0 + c8 * 8 + a.out_shift;
      if (flat < 0 || flat >= a.out_limit) continue;
      const bf16x8 o = *reinterpret_cast<const bf16x8*>(Os + n * C::OS + c8 * 8);
      float f[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = (float)o[e];
      if (a.res) {
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(a.res + ob + flat);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)rv[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] *= scale;
      if (a.accumulate) {
        const bf16x8 pv = *reinterpret_cast<const bf16x8*>(a.out + ob + flat);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] += (float)pv[e];
      }
      bf16x8 res;
#pragma unroll
      for (int e = 0; e < 8; ++e) res[e] = (bf16_t)(post != 1.f ? lrelu(f[e], post) : f[e]);
      *reinterpret_cast<bf16x8*>(a.out + ob + flat) = res;
    }
  }
}

// ---- instantiation table ------------------------------------------------------------------------
struct ConvTcEntry {
  int c_in, ks, max_dil, kc;
  ConvTcLaunch launch;
};

template <class C>
static ConvTcEntry make_entry(const char* name) {
  ConvTcEntry e;
  e.c_in = C::CIN;
  e.ks = C::KS;
  e.max_dil = C::MAXDIL;
  e.kc = C::KC;
  e.launch.kernel = conv_tc_kernel<C>;
  e.launch.bm = C::BM;
  e.launch.bn = C::BN;
  e.launch.threads = C::NTHREADS;
  e.launch.lds_bytes = C::LDS;
  e.launch.name = name;
  e.launch.kc = C::KC;
  return e;
}

//                       CIN  KC   BM   BN  WM WN KS TAPS MAXDIL
#define EVMI_CONV_TC_TABLE(X)                                                        \
  /* conv_pre 80 -> 512, k7 */                                                       \
  X(80, 80, 128, 128, 2, 2, 7, 1, 1)                                                 \
  /* transposed-conv upsamplers in polyphase form (2 taps) */                        \
  X(512, 64, 128, 128, 2, 2, 2, 2, 1)                                                \
  X(256, 64, 128, 128, 2, 2, 2, 2, 1)                                                \
  X(128, 64, 128, 128, 2, 2, 2, 2, 1)                                                \
  X(64, 64, 64, 128, 1, 4, 2, 2, 1)                                                  \
  /* MRF residual-block convolutions, dilation <= 5 */                               \
  X(256, 64, 128, 128, 2, 2, 3, 1, 5)                                                \
  X(256, 64, 128, 128, 2, 2, 7, 1, 5)                                                \
  X(256, 64, 128, 128, 2, 2, 11, 1, 5)                                               \
  X(128, 64, 128, 128, 2, 2, 3, 1, 5)                                                \
  X(128, 64, 128, 128, 2, 2, 7, 1, 5)                                                \
  X(128, 64, 128, 128, 2, 2, 11, 1, 5)                                               \
  X(64, 64, 64, 128, 1, 4, 3, 3, 5)                                                  \
  X(64, 64, 64, 128, 1, 4, 7, 2, 5)                                                  \
  X(64, 64, 64, 128, 1, 4, 11, 2, 5)                                                 \
  X(32, 32, 32, 256, 1, 4, 3, 3, 5)                                                  \
  X(32, 32, 32, 256, 1, 4, 7, 7, 5)                                                  \
  X(32, 32, 32, 256, 1, 4, 11, 11, 5)

static const ConvTcEntry* conv_tc_table(int* n) {
#define X(cin, kc, bm, bn, wm, wn, ks, taps, md)                                               \
  make_entry<ConvTcCfg<cin, kc, bm, bn, wm, wn, ks, taps, md>>(                                \
      "conv_tc_mfma<c" #cin ",k" #ks ",bm" #bm ",bn" #bn ",kc" #kc ",t" #taps ">"),
  static const ConvTcEntry table[] = {EVMI_CONV_TC_TABLE(X)};
#undef X
  *n = (int)(sizeof(table) / sizeof(table[0]));
  return table;
}

const ConvTcLaunch* find_conv_tc(int c_in, int c_out, int ks, int dil) {
  int n = 0;
  const ConvTcEntry* t = conv_tc_table(&n);
  for (int i = 0; i < n; ++i) {
    if (t[i].c_in == c_in && t[i].ks == ks && dil <= t[i].max_dil && c_out % t[i].launch.bm == 0)
      return &t[i].launch;
  }
  return nullptr;
}

int launch_conv_tc(const ConvTcLaunch* L, const ConvTcArgs& a, int B, hipStream_t stream) {
  static thread_local const void* configured[64];
  static thread_local int n_configured = 0;
  bool seen = false;
  for (int i = 0; i < n_configured; ++i) seen |= (configured[i] == (const void*)L->kernel);
  if (!seen) {
    EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)L->kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)L->lds_bytes));
    if (n_configured < 64) configured[n_configured++] = (const void*)L->kernel;
  }
  dim3 grid((a.n_rows + L->bn - 1) / L->bn, B, a.c_out / L->bm);
  hipLaunchKernelGGL(L->kernel, grid, dim3(L->threads), L->lds_bytes, stream, a);
  EVMI_LAUNCH_CHECK(L->name);
  return EVMI_OK;
}

}  // namespace evmi
