// fp32 implicit-GEMM 1-D convolution on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact fp32
// fmaf chains, 157 TFLOP/s class) for the channel-major training layout x[c][b][t].
//
//   y[co][b][to*os + oo] (+)= bias[co] + sum_{ci in group} sum_{j<k} w[co][ci][j] * x[ci][b][to*s + j*d - p]
//
// GEMM per workgroup: D[BM out-channels x BN columns], columns = the flattened (b, to) index n = b*n_out + to,
// so short sequences (the period discriminators' H = 4..50 rows) fill a tile with several batch items instead
// of wasting it.  K runs over (channel pair, tap, channel parity): the two K slots of one MFMA are the even and
// the odd channel of a pair at the same tap, so the per-lane part of every operand address is a constant and
// the per-K part a wave-uniform scalar (A: +1 fragment, X: + the dilation per tap).
//
// Weights are first re-laid (wfrag_kernel, a few microseconds) into MFMA fragment order
//   wf[g][m-block of 32 rows][channel pair][tap][parity][32 rows]      (zero padded),
// i.e. one 256-byte fragment per (m-block, K pair) that a wave reads with lane-linear ds_read_b32 and that the
// loader copies as contiguous kilobytes.  Both operands travel global -> LDS with the LDS-direct loads of gfx950
// (global_load_lds_dwordx4 for the weight fragments, global_load_lds_dword for the input windows): no staging
// registers, so the K loop is a three-slot LDS ring -- the loads of step t+2 are in flight while step t feeds the
// matrix cores (one barrier per step, vmcnt counted so that only the older group is waited for), and the waves
// keep their registers for a 64x64 (or 32x128) accumulator tile at two workgroups per CU.  Small workgroup tiles
// (needed to fill 256 CUs on the short layers) keep that wave tile by splitting the K pairs of each step over the
// waves (KS) and adding the partial tiles through LDS at the end.  Any stride / dilation / groups; `os`, `oo`
// place the outputs on a strided grid so that the gradient of a strided convolution runs as `stride` polyphase
// stride-1 convolutions through this same kernel.
#include <algorithm>
#include <cstdint>
#include <type_traits>
#include <cstdio>
#include <cstdlib>

#include "common.h"
#include "conv_cbt_direct.h"

namespace evmi {

struct ConvF32Args {
  const float* x;     // [c_in][B][t_in]
  const float* wf;    // fragment-ordered weights (wfrag_kernel)
  const float* bias;  // [c_out] or nullptr
  float* y;           // [c_out][B][t_out_total]
  int B, t_in, t_out_total;
  int n_out;          // output positions computed per (co, b): to in [0, n_out)
  int cin_g, cout_g, k, stride, dil, pad;
  int out_stride, out_offset;  // y index = to * out_stride + out_offset
  int accumulate;              // y += instead of y =
  int act;                     // epilogue on (conv + bias): 0 none, 1 leaky-relu(act_param), 2 SiLU, 3 ReLU, 4 tanh
  float act_param;
  int mtiles_per_group;
  int mblocks, pairs;          // wf dims: ceil(cout_g / 32), ceil(cin_g / 2)
  // tiling chosen by the host
  int ps;       // channel pairs per step
  int xrow;     // LDS row stride of the staged input rows (odd)
  int pieces;   // ceil(xrow / 64)
  int stage;    // floats per ring slot (weights fragments, then input rows)
  int nst;      // ring slots: 3 (loads two steps ahead) or 2 (one step ahead, twice the step depth in the same LDS)
  // polyphase input gradients: grid.z = phase, every phase its own padding / extent / output offset / weight fragments
  int phases;
  int n_out_min;  // planning: shortest phase (most items per tile); 0 = n_out
  long long wf_phase_stride;
  int ph_pad[8], ph_nout[8], ph_off[8];
  // bf16-operand mode (BF kernels): K runs over blocks of 16 channels at one tap; LDS still holds fp32 rows, the consumer
  // gathers 8 channels per lane, rounds them to bf16 and feeds v_mfma_f32_32x32x16_bf16 (fp32 accumulation)
  int bf;       // 1 = bf16 operands
  int cblocks;  // ceil(cin_g / 16): 16-channel blocks of the weight fragments
  int tj, nch;  // taps per step and steps per channel step (long kernels are split over the ring steps)
  int xcd_remap;  // 1 = XCD-aware tile order (EVMI_F32_XCD=0 switches it off for A/B runs)
  int ablate;   // timing experiments only (EVMI_F32_ABLATE): 1 no input loads, 2 no weight loads, 4 no MFMA
  long long* tl;  // timing experiments only (EVMI_F32_TL): s_memtime stamps of workgroup (0, 0), [step][wave][4]
};

// activation epilogue shared by the convolution kernels (the code is wave-uniform: one branch)
template <int ACT>
__device__ __forceinline__ float conv_act(float v, float p) {
  if (ACT == 1) return v > 0.f ? v : v * p;
  if (ACT == 2) return v / (1.f + expf(-v));
  if (ACT == 3) return fmaxf(v, 0.f);
  if (ACT == 4) return tanhf(v);
  return v;
}
// run `body(integral_constant<ACT>)` for the runtime activation code: one wave-uniform switch outside the store loops
template <class F>
__device__ __forceinline__ void with_act(int act, F&& body) {
  switch (act) {
    case 1: body(std::integral_constant<int, 1>{}); break;
    case 2: body(std::integral_constant<int, 2>{}); break;
    case 3: body(std::integral_constant<int, 3>{}); break;
    case 4: body(std::integral_constant<int, 4>{}); break;
    default: body(std::integral_constant<int, 0>{}); break;
  }
}

constexpr int SMALLCO_DIRECT = 4;  // output channels up to which conv_cbt_direct.hip takes the shape
constexpr int F32_PMAX = 10;  // 64-column pieces of a staged row (xrow <= 640)

__device__ float g_zero_line[64];  // source of the zeros staged for padding / out-of-range columns

typedef __attribute__((address_space(3))) float lds_float_t;
typedef __attribute__((address_space(1))) const float glb_float_t;
__device__ __forceinline__ void lds_direct_b32(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((glb_float_t*)g, (lds_float_t*)l, 4, 0, 0);
}
__device__ __forceinline__ void lds_direct_b128(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((glb_float_t*)g, (lds_float_t*)l, 16, 0, 0);
}

// wf[(((g*MB + mb)*P + p)*k + j)*64 + kh*32 + mi] = w[g*cout_g + mb*32 + mi][2p + kh][j]  (0 outside)
// grid (P, groups*MB): one workgroup per (m-block, channel pair) = k fragments of 64 floats
__global__ __launch_bounds__(256) void wfrag_kernel(const float* __restrict__ w, float* __restrict__ wf, int cout_g,
                                                    int cin_g, int k, int MB, int P) {
  const int p = blockIdx.x, gmb = blockIdx.y;
  const int g = gmb / MB, mb = gmb - g * MB;
  float* dst = wf + ((long long)gmb * P + p) * k * 64;
  for (int e = threadIdx.x; e < k * 64; e += 256) {
    const int j = e >> 6, kh = (e >> 5) & 1, mi = e & 31;
    const int m = mb * 32 + mi, ci = 2 * p + kh;
    dst[e] = (m < cout_g && ci < cin_g) ? w[((long long)(g * cout_g + m) * cin_g + ci) * k + j] : 0.f;
  }
}

// Fragment-ordered weights of the polyphase input-gradient convolutions, straight from w [c_out][cin_g][k]:
// the gradient of y = conv(x, w, stride s) with respect to x is, for every phase phi < min(s, k), a stride-1 convolution of
// dy (c_out channels in, c_in channels out) with the taps j = phi + s*m of w in reverse order.  All phases get the same
// tap count M = ceil(k / s) (phases with fewer taps are zero padded in front), so they share one launch geometry:
//   wfd[ph][(((g*MB + mb)*P + p)*M + m)*64 + kh*32 + mi] = w[g*cout_g + 2p+kh][mb*32+mi][phi + s*(M_phi - 1 - (m - (M - M_phi)))]
// with MB = ceil(cin_g / 32), P = ceil(cout_g / 2).  grid (P, groups*MB, phases)
__global__ __launch_bounds__(256) void wfrag_dgrad_kernel(const float* __restrict__ w, float* __restrict__ wfd, int cout_g, int cin_g,
                                                          int k, int stride, int M, int MB, int P, long long phase_stride) {
  const int p = blockIdx.x, gmb = blockIdx.y, phi = blockIdx.z;
  const int g = gmb / MB, mb = gmb - g * MB;
  const int m_phi = (k - phi + stride - 1) / stride, lead = M - m_phi;
  float* dst = wfd + phi * phase_stride + ((long long)gmb * P + p) * M * 64;
  for (int e = threadIdx.x; e < M * 64; e += 256) {
    const int m = e >> 6, kh = (e >> 5) & 1, mi = e & 31;
    const int ci = mb * 32 + mi, co = 2 * p + kh;
    float v = 0.f;
    if (ci < cin_g && co < cout_g && m >= lead) {
      const int j = phi + stride * (m_phi - 1 - (m - lead));
      v = w[((long long)(g * cout_g + co) * cin_g + ci) * k + j];
    }
    dst[e] = v;
  }
}

// bf16 fragments (v_mfma_f32_32x32x16_bf16 A operand): one 1 KB fragment per (m-block, 16-channel block, tap); lane
// (mi = lane & 31, kh = lane >> 5) holds the 8 channels cb*16 + kh*8 + 0..7 of row mb*32 + mi as four 32-bit words:
//   wfb[(((g*MB + mb)*CB + cb)*k + j)*256 + lane*4 + q] = bf16x2(w[.][cb*16 + kh*8 + 2q][j], w[.][.. + 2q + 1][j])
// grid (CB, groups*MB)
__global__ __launch_bounds__(256) void wfrag_bf16_kernel(const float* __restrict__ w, unsigned* __restrict__ wf, int cout_g,
                                                         int cin_g, int k, int MB, int CB) {
  const int cb = blockIdx.x, gmb = blockIdx.y;
  const int g = gmb / MB, mb = gmb - g * MB;
  unsigned* dst = wf + ((long long)gmb * CB + cb) * k * 256;
  for (int e = threadIdx.x; e < k * 256; e += 256) {
    const int j = e >> 8, ln = (e >> 2) & 63, q = e & 3;
    const int m = mb * 32 + (ln & 31), ci = cb * 16 + (ln >> 5) * 8 + 2 * q;
    float lo = 0.f, hi = 0.f;
    if (m < cout_g) {
      const float* wr = w + (long long)(g * cout_g + m) * cin_g * k + j;
      if (ci < cin_g) lo = wr[(long long)ci * k];
      if (ci + 1 < cin_g) hi = wr[(long long)(ci + 1) * k];
    }
    dst[e] = pack_bf16x2(lo, hi);
  }
}

// bf16 fragments of the polyphase input-gradient convolutions (see wfrag_dgrad_kernel): rows = x channels, K = dy channels.
// MB = ceil(cin_g / 32), CB = ceil(cout_g / 16); grid (CB, groups*MB, phases)
__global__ __launch_bounds__(256) void wfrag_dgrad_bf16_kernel(const float* __restrict__ w, unsigned* __restrict__ wfd, int cout_g,
                                                               int cin_g, int k, int stride, int M, int MB, int CB,
                                                               long long phase_stride) {
  const int cb = blockIdx.x, gmb = blockIdx.y, phi = blockIdx.z;
  const int g = gmb / MB, mb = gmb - g * MB;
  const int m_phi = (k - phi + stride - 1) / stride, lead = M - m_phi;
  unsigned* dst = wfd + phi * phase_stride + ((long long)gmb * CB + cb) * M * 256;
  for (int e = threadIdx.x; e < M * 256; e += 256) {
    const int m = e >> 8, ln = (e >> 2) & 63, q = e & 3;
    const int ci = mb * 32 + (ln & 31), co = cb * 16 + (ln >> 5) * 8 + 2 * q;
    float lo = 0.f, hi = 0.f;
    if (ci < cin_g && m >= lead) {
      const int j = phi + stride * (m_phi - 1 - (m - lead));
      if (co < cout_g) lo = w[((long long)(g * cout_g + co) * cin_g + ci) * k + j];
      if (co + 1 < cout_g) hi = w[((long long)(g * cout_g + co + 1) * cin_g + ci) * k + j];
    }
    dst[e] = pack_bf16x2(lo, hi);
  }
}

// One K block (16 channels at one tap) of a wave tile in bf16-operand mode: A fragments are 16-byte LDS reads, B fragments
// 8 fp32 reads (8 channel rows of the staged window at this lane's column) rounded to bf16.
template <int MT, int NT>
struct BfRaw {
  bf16x8 a[MT];
  float x[NT][8];
};
template <int MT, int NT>
__device__ __forceinline__ void bf_load(BfRaw<MT, NT>& f, const float* __restrict__ sm, const int (&abase)[MT], const int (&xbase)[NT],
                                        int off_a, int off_x, int xrow) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) f.a[mt] = *reinterpret_cast<const bf16x8*>(sm + abase[mt] + off_a);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const float* col = sm + xbase[nt] + off_x;
#pragma unroll
    for (int i = 0; i < 8; ++i) f.x[nt][i] = col[i * xrow];
  }
}
// K blocks [q_lo, q_hi) of a staged step (tjc taps per 16-channel block).  The LDS reads of block q+1 are issued before the
// MFMAs of block q and converted after them, so the matrix pipe runs while the next operands arrive (one loop, no early
// exits: the accumulators stay in place).  The block after the last one re-reads the last (in-bounds, unused).
template <int MT, int NT>
__device__ __forceinline__ void bf_consume(const float* __restrict__ sm, const int (&abase)[MT], const int (&xbase)[NT], int base,
                                           int q_lo, int q_hi, int tjc, int j0, int d, int xrow, f32x16 (&acc)[MT][NT]) {
  if (q_lo >= q_hi) return;
  int cb = q_lo / tjc, jj = q_lo - cb * tjc;
  int off_a = base + q_lo * 256;
  int off_x = base + cb * 16 * xrow + (j0 + jj) * d;
  BfRaw<MT, NT> nx;
  bf_load(nx, sm, abase, xbase, off_a, off_x, xrow);
  bf16x8 fa[MT], fb[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) fa[mt] = nx.a[mt];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int i = 0; i < 8; ++i) fb[nt][i] = (bf16_t)nx.x[nt][i];
#pragma unroll 2
  for (int q = q_lo; q < q_hi; ++q) {
    if (q + 1 < q_hi) {  // scalar bookkeeping only
      ++jj; off_a += 256; off_x += d;
      if (jj == tjc) { jj = 0; off_x += 16 * xrow - tjc * d; }
    }
    bf_load(nx, sm, abase, xbase, off_a, off_x, xrow);
    __builtin_amdgcn_sched_barrier(0);  // all LDS reads of block q+1 are in flight before the MFMAs of block q issue
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fa[mt] = nx.a[mt];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int i = 0; i < 8; ++i) fb[nt][i] = (bf16_t)nx.x[nt][i];
  }
}

template <int BM, int BN, int WM, int WN, int KS, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv_cbt_f32_mfma_kernel(ConvF32Args a) {
  constexpr int NTHREADS = 256;
  static_assert(WM * WN * KS == 4, "four waves");
  constexpr int MT = BM / (WM * 32), NT = BN / (WN * 32);
  constexpr int U = 2;              // K pairs per software-pipeline group
  constexpr int MBT = BM / 32;      // m-blocks of the tile
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;

  // XCD-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs (private L2s); give every XCD a
  // contiguous range of the (m-tile major) tile list instead, so the workgroups that share weight fragments share an L2
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd_remap) {
    const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
    if (nwg >= 16) {
      const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
      const unsigned L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
      by = L / gridDim.x;
      bx = L - by * gridDim.x;
    }
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave / (WM * WN), wmn = wave % (WM * WN), wm = wmn / WN, wn = wmn % WN;
  const int kh = lane >> 5, ln = lane & 31;
  const int g = by / a.mtiles_per_group, mt_idx = by % a.mtiles_per_group;
  const int co0 = g * a.cout_g + mt_idx * BM;
  const int k = a.k, s = a.stride, d = a.dil, xrow = a.xrow, ps = a.ps, pieces = a.pieces;
  const int nblk_step = ps >> 3;              // BF: 16-channel blocks per channel step
  const int nqa_pad = BF ? nblk_step * a.tj * 4 : (ps * k + 3) & ~3;  // 256-byte units per m-block per slot (whole 1 KB quads)
  const int a_floats = MBT * nqa_pad * 64;    // weight part of a slot
  const int halo = (k - 1) * d + 1;
  const int gap = max(halo - s, 0);  // extra columns between the staged segments of consecutive items
  const int ph = blockIdx.z;
  const int n_out = a.ph_nout[ph], pad_ = a.ph_pad[ph], out_off = a.ph_off[ph];
  if (n_out <= 0) return;  // a phase without outputs
  const float* wf_ = a.wf + (long long)ph * a.wf_phase_stride;
  const long long n_total = (long long)a.B * n_out;
  const long long n0 = (long long)bx * BN;
  if (n0 >= n_total) return;  // tiles past a (shorter) phase
  const int b_first = (int)(n0 / n_out);
  const int to_first = (int)(n0 - (long long)b_first * n_out);
  const int m_valid = min(BM, a.cout_g - mt_idx * BM);
  const long long ch_stride = (long long)a.B * a.t_in;
  const float* xgrp = a.x + (long long)g * a.cin_g * ch_stride;

  // per-lane column bookkeeping: tile column c -> (b, to); LDS column base = c*s + (b - b_first)*gap
  int xbase[NT], col_b[NT], col_to[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (wn * NT + nt) * 32 + ln;
    const long long n = n0 + c;
    if (n < n_total) {
      const int bb = (int)(n / n_out);
      col_b[nt] = bb;
      col_to[nt] = (int)(n - (long long)bb * n_out);
      xbase[nt] = a_floats + kh * (BF ? 8 : 1) * xrow + c * s + (bb - b_first) * gap;
    } else {
      col_b[nt] = -1;
      col_to[nt] = 0;
      xbase[nt] = a_floats + kh * (BF ? 8 : 1) * xrow;  // finite staged data; the column is never stored
    }
  }
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = (wm * MT + mt) * nqa_pad * 64 + (BF ? lane * 4 : lane);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // source offset (b*t_in + ti, -1 = zero) of the staged columns this lane loads: the same for every step and row
  int so[F32_PMAX];
  {
    int* Stab = reinterpret_cast<int*>(smem);
    for (int v = tid; v < pieces * 64; v += NTHREADS) Stab[v] = -1;
    lds_barrier();
    int c = 0, bb = b_first, to_lo = to_first;
    while (c < BN && bb < a.B) {
      const int cnt = min(n_out - to_lo, BN - c);
      const int seglen = (cnt - 1) * s + halo;
      const int ti0 = to_lo * s - pad_;
      int* seg = Stab + c * s + (bb - b_first) * gap;
      for (int i = tid; i < seglen; i += NTHREADS) {
        const int ti = ti0 + i;
        seg[i] = (ti >= 0 && ti < a.t_in) ? bb * a.t_in + ti : -1;
      }
      c += cnt;
      ++bb;
      to_lo = 0;
    }
    lds_barrier();
#pragma unroll
    for (int pi = 0; pi < F32_PMAX; ++pi) so[pi] = pi < pieces ? Stab[pi * 64 + lane] : -1;
    lds_barrier();
  }

  const int nsteps = ((a.pairs + ps - 1) / ps) * (BF ? a.nch : 1);
  if (a.ablate) {  // experiments read uninitialised LDS otherwise
    for (int v = tid; v < a.nst * a.stage; v += NTHREADS) smem[v] = 0.f;
    lds_barrier();
  }

  // ---- loader: every wave brings its share of the operands of step t -> ring slot t%3 (LDS-direct, no registers).
  // Weights: the 1 KB quads of the step round-robin over the waves (SGPR base + lane offset addressing);
  // inputs: rows round-robin over the waves.  Returns the number of loads this wave issued. ----
  const long long mb_stride = BF ? (long long)a.cblocks * k * 256 : (long long)a.pairs * k * 64;
  const float* wf_tile = wf_ + (long long)(g * a.mblocks + mt_idx * MBT) * mb_stride;
  const int mb_last = a.mblocks - 1 - mt_idx * MBT;  // m-blocks past the group re-read the last one (never stored)
  const unsigned lane16 = lane * 16;
  auto issue = [&](int t, int slot) -> int {
    const int tc = BF ? t / a.nch : t;   // channel step; BF: tap chunk t % nch of it
    const int c0 = tc * 2 * ps;
    const int cbcur = min(2 * ps, a.cin_g - c0);
    const int pairs_cur = (cbcur + 1) >> 1;
    const int nblk_cur = (cbcur + 15) >> 4;
    const int j0 = BF ? (t - tc * a.nch) * a.tj : 0;
    // f32: whole quads, the tail fragments belong to the next step (or the slack); BF: one 1 KB fragment per (block, tap)
    const int nquads = BF ? nblk_cur * min(a.tj, k - j0) : (pairs_cur * k + 3) >> 2;
    const long long a_step = BF ? ((long long)tc * nblk_step * k + j0) * 256 : (long long)t * ps * k * 64;
    const int nrows = BF ? 16 * nblk_cur : 2 * pairs_cur;
    float* sa = smem + slot * a.stage;
    float* sx = sa + a_floats;
    int issued = 0;
    if (!(a.ablate & 2)) {
      int u = wave;  // unit = (m-block, quad), round-robin over the four waves
#pragma unroll
      for (int mbi = 0; mbi < MBT; ++mbi) {
        const char* src = reinterpret_cast<const char*>(wf_tile + min(mbi, mb_last) * mb_stride + a_step);
        float* dst = sa + mbi * nqa_pad * 64;
        for (; u < nquads; u += 4) {
          lds_direct_b128(reinterpret_cast<const float*>(src + (size_t)u * 1024 + lane16), dst + u * 256);
          ++issued;
        }
        u -= nquads;
      }
    }
    if (!(a.ablate & 1)) {
      const float* xr = xgrp + (long long)(c0 + wave) * ch_stride;
      float* dstrow = sx + wave * xrow;
      for (int r = wave; r < nrows; r += 4) {
        const bool zero_row = r >= cbcur;
#pragma unroll
        for (int pi = 0; pi < F32_PMAX; ++pi) {
          if (pi >= pieces) break;
          const float* srcp = (zero_row || so[pi] < 0) ? g_zero_line + lane : xr + so[pi];
          if (pi * 64 + lane < xrow) lds_direct_b32(srcp, dstrow + pi * 64);
          ++issued;
        }
        xr += 4 * ch_stride;
        dstrow += 4 * xrow;
      }
    }
    return issued;
  };
  int n_next = 0;  // loads of the group issued after the one about to be consumed
  const int nst = a.nst;
  issue(0, 0);
  if (nst == 3 && nsteps > 1) n_next = issue(1, 1);

  int slot = -1;
  for (int t = 0; t < nsteps; ++t) {
    slot = slot + 1 == nst ? 0 : slot + 1;       // t % nst
    const int slot_ahead = slot == 0 ? nst - 1 : slot - 1;  // (t + nst - 1) % nst: the slot step t-1 just released
    const bool stamp = a.tl && bx == 0 && by == 0 && t < 24;
    long long* tl = a.tl + (t * 4 + wave) * 4;
    if (stamp && lane == 0) tl[0] = __builtin_readcyclecounter();
    wait_vmcnt_le(n_next);  // step t has landed (this wave's part); step t+1 may still be in flight
    lds_barrier();          // ... everyone's part; slot (t+2)%3 was last read in step t-1
    if (stamp && lane == 0) tl[1] = __builtin_readcyclecounter();
    {
      const int issued = t + nst - 1 < nsteps ? issue(t + nst - 1, slot_ahead) : 0;
      n_next = nst == 3 ? issued : 0;  // two slots: the step just issued is the next one consumed: wait for all of it
    }
    if (stamp && lane == 0) tl[2] = __builtin_readcyclecounter();
    if (BF) {
      const int tc = t / a.nch, j0 = (t - tc * a.nch) * a.tj, tjc = min(a.tj, k - j0);
      const int nkb = ((min(2 * ps, a.cin_g - tc * 2 * ps) + 15) >> 4) * tjc;
      const int q_lo = KS == 1 ? 0 : (ks * nkb) / KS, q_hi = KS == 1 ? nkb : ((ks + 1) * nkb) / KS;
      if (!(a.ablate & 4)) bf_consume<MT, NT>(smem, abase, xbase, slot * a.stage, q_lo, q_hi, tjc, j0, d, xrow, acc);
      if (stamp && lane == 0) tl[3] = __builtin_readcyclecounter();
      continue;
    }
    const int cbcur = min(2 * ps, a.cin_g - t * 2 * ps);
    const int nq_all = ((cbcur + 1) >> 1) * k;
    const int q_lo = KS == 1 ? 0 : (ks * nq_all) / KS;
    const int q_hi = KS == 1 ? nq_all : ((ks + 1) * nq_all) / KS;
    const int nq = q_hi - q_lo;
    // wave-uniform operand offsets of the current K pair
    int pl0 = 0, j = 0;
    if (KS > 1) { pl0 = q_lo / k; j = q_lo - pl0 * k; }
    int off_a = slot * a.stage + q_lo * 64;
    int off_x = slot * a.stage + pl0 * 2 * xrow + j * d;
    float fa[U][MT], fb[U][NT], ga[U][MT], gb[U][NT];
    auto load_group = [&](float (&da)[U][MT], float (&db)[U][NT]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) da[u][mt] = smem[abase[mt] + off_a];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) db[u][nt] = smem[xbase[nt] + off_x];
        ++j; off_a += 64; off_x += d;
        if (j == k) { j = 0; off_x += 2 * xrow - k * d; }
      }
    };
    load_group(fa, fb);
    const int nfull = nq / U, rem = nq - nfull * U;
    if (!(a.ablate & 4))
    for (int gi = 0; gi < nfull; ++gi) {
      load_group(ga, gb);  // the group after this one (past the end: in-bounds LDS, never used)
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][mt], fb[u][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < U; ++u) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[u][mt] = ga[u][mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[u][nt] = gb[u][nt];
      }
      // issue order: one operand read and its address arithmetic behind every MFMA
#pragma unroll
      for (int i = 0; i < U * MT * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    }
#pragma unroll
    for (int u = 0; u < U - 1; ++u)
      if (u < rem) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][mt], fb[u][nt], acc[mt][nt], 0, 0, 0);
      }
    if (stamp && lane == 0) tl[3] = __builtin_readcyclecounter();
  }

  // ---- K-split partial tiles: waves ks > 0 hand their accumulators to wave ks == 0 through LDS ----
  if (KS > 1) {
    float* R = smem;  // [KS-1][WM*WN][MT][NT][16][64]
    lds_barrier();
    if (ks > 0) {
      float* dst = R + ((ks - 1) * WM * WN + wmn) * (MT * NT * 16 * 64) + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((mt * NT + nt) * 16 + r) * 64] = acc[mt][nt][r];
    }
    lds_barrier();
    if (ks > 0) return;
#pragma unroll 1
    for (int kk = 0; kk < KS - 1; ++kk) {
      const float* src = R + (kk * WM * WN + wmn) * (MT * NT * 16 * 64) + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] += src[((mt * NT + nt) * 16 + r) * 64];
    }
  }

  // ---- epilogue: D layout: lane column = output position, registers = output channels ----
  with_act(a.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (col_b[nt] < 0) continue;
      float* ycol = a.y + (long long)col_b[nt] * a.t_out_total + (long long)col_to[nt] * a.out_stride + out_off;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        constexpr int EB = 4;  // (8 would push the 128-register kernels of this family over their occupancy step)
        // bias and previous values in batches of EB from clamped rows: a read under a per-element condition costs a drained
        // round trip each (64 in a row per lane); 0 / -0 stand in for an absent operand
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += EB) {
          int mrow[EB];
          float* dst[EB];
          float bv[EB], prev[EB];
#pragma unroll
          for (int e = 0; e < EB; ++e) {
            const int r = r0 + e;
            mrow[e] = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            dst[e] = ycol + (long long)(co0 + min(mrow[e], m_valid - 1)) * a.B * a.t_out_total;
            bv[e] = 0.f;
            prev[e] = -0.f;
          }
          if (a.bias) {
#pragma unroll
            for (int e = 0; e < EB; ++e) bv[e] = a.bias[co0 + min(mrow[e], m_valid - 1)];
          }
          if (a.accumulate) {
#pragma unroll
            for (int e = 0; e < EB; ++e) prev[e] = *dst[e];
          }
#pragma unroll
          for (int e = 0; e < EB; ++e) {
            const float v = conv_act<ACT>(a.bias ? acc[mt][nt][r0 + e] + bv[e] : acc[mt][nt][r0 + e], a.act_param);
            if (mrow[e] < m_valid) *dst[e] = prev[e] + v;
          }
        }
      }
    }
  });
}

// ---- wave-private variant (the K-split tiles) -------------------------------------------------------------------------
// In a K-split tile no two waves share an operand: wave (wn, ks) owns the columns [wn*BNW, +BNW) of the tile and every
// KS-th step of the contraction.  So each wave runs its own two-slot LDS ring -- its own LDS-direct loads, its own
// vmcnt waits -- with NO workgroup barrier in the K loop; the partial tiles meet in LDS once at the end.  A wave's
// step is then KS times longer than a step of the shared kernel above for the same LDS (the per-step barrier + load
// issue phase was what limited these tiles: timeline stamps), and the load issue of one wave overlaps the
// MFMAs of the waves of the other co-resident workgroup.
template <int BM, int BNW, int NWN, int KS, bool BF = false>
__global__ __launch_bounds__(256, 2) void conv_cbt_f32_mfma_wp_kernel(ConvF32Args a) {
  static_assert(NWN * KS == 4, "four waves");
  constexpr int MT = BM / 32, NT = BNW / 32;
  constexpr int U = 2;
  constexpr int MBT = BM / 32;
  constexpr int BN = BNW * NWN;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;

  // XCD-aware tile order: the dispatcher deals workgroups round-robin over the 8 XCDs (private L2s); give every XCD a
  // contiguous range of the (m-tile major) tile list instead, so the workgroups that share weight fragments share an L2
  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd_remap) {
    const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
    if (nwg >= 16) {
      const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
      const unsigned L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
      by = L / gridDim.x;
      bx = L - by * gridDim.x;
    }
  }
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ks = wave / NWN, wn = wave % NWN;
  const int kh = lane >> 5, ln = lane & 31;
  const int g = by / a.mtiles_per_group, mt_idx = by % a.mtiles_per_group;
  const int co0 = g * a.cout_g + mt_idx * BM;
  const int k = a.k, s = a.stride, d = a.dil, xrow = a.xrow, ps = a.ps, pieces = a.pieces;
  const int nblk_step = ps >> 3;
  const int nqa_pad = BF ? nblk_step * a.tj * 4 : (ps * k + 3) & ~3;
  const int a_floats = MBT * nqa_pad * 64;
  const int halo = (k - 1) * d + 1;
  const int gap = max(halo - s, 0);  // extra columns between the staged segments of consecutive items
  const int ph = blockIdx.z;
  const int n_out = a.ph_nout[ph], pad_ = a.ph_pad[ph], out_off = a.ph_off[ph];
  if (n_out <= 0) return;  // a phase without outputs
  const float* wf_ = a.wf + (long long)ph * a.wf_phase_stride;
  const long long n_total = (long long)a.B * n_out;
  if ((long long)bx * BN >= n_total) return;  // tiles past a (shorter) phase
  const long long n0 = (long long)bx * BN + wn * BNW;  // first column of this wave
  const bool wave_live = n0 < n_total;
  const int b_first = wave_live ? (int)(n0 / n_out) : 0;
  const int to_first = wave_live ? (int)(n0 - (long long)b_first * n_out) : 0;
  const int m_valid = min(BM, a.cout_g - mt_idx * BM);
  const long long ch_stride = (long long)a.B * a.t_in;
  const float* xgrp = a.x + (long long)g * a.cin_g * ch_stride;
  float* wsm = smem + wave * (2 * a.stage);  // this wave's two slots

  int xbase[NT], col_b[NT], col_to[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = nt * 32 + ln;
    const long long n = n0 + c;
    if (n < n_total) {
      const int bb = (int)(n / n_out);
      col_b[nt] = bb;
      col_to[nt] = (int)(n - (long long)bb * n_out);
      xbase[nt] = a_floats + kh * (BF ? 8 : 1) * xrow + c * s + (bb - b_first) * gap;
    } else {
      col_b[nt] = -1;
      col_to[nt] = 0;
      xbase[nt] = a_floats + kh * (BF ? 8 : 1) * xrow;
    }
  }
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = mt * nqa_pad * 64 + (BF ? lane * 4 : lane);

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // source offsets of the staged columns of this wave (wave-private scratch: in-order LDS within a wave, no barrier)
  int so[F32_PMAX];
  {
    int* Stab = reinterpret_cast<int*>(wsm);
    for (int v = lane; v < pieces * 64; v += 64) Stab[v] = -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int c = 0, bb = b_first, to_lo = to_first;
    while (wave_live && c < BNW && bb < a.B) {
      const int cnt = min(n_out - to_lo, BNW - c);
      const int seglen = (cnt - 1) * s + halo;
      const int ti0 = to_lo * s - pad_;
      int* seg = Stab + c * s + (bb - b_first) * gap;
      for (int i = lane; i < seglen; i += 64) {
        const int ti = ti0 + i;
        seg[i] = (ti >= 0 && ti < a.t_in) ? bb * a.t_in + ti : -1;
      }
      c += cnt;
      ++bb;
      to_lo = 0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int pi = 0; pi < F32_PMAX; ++pi) so[pi] = pi < pieces ? Stab[pi * 64 + lane] : -1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }

  const int nsteps = ((a.pairs + ps - 1) / ps) * (BF ? a.nch : 1);
  const long long mb_stride = BF ? (long long)a.cblocks * k * 256 : (long long)a.pairs * k * 64;
  const float* wf_tile = wf_ + (long long)(g * a.mblocks + mt_idx * MBT) * mb_stride;
  const int mb_last = a.mblocks - 1 - mt_idx * MBT;
  const unsigned lane16 = lane * 16;
  auto issue = [&](int t, int slot) -> int {
    const int tc = BF ? t / a.nch : t;
    const int c0 = tc * 2 * ps;
    const int cbcur = min(2 * ps, a.cin_g - c0);
    const int pairs_cur = (cbcur + 1) >> 1;
    const int nblk_cur = (cbcur + 15) >> 4;
    const int j0 = BF ? (t - tc * a.nch) * a.tj : 0;
    const int nquads = BF ? nblk_cur * min(a.tj, k - j0) : (pairs_cur * k + 3) >> 2;
    const long long a_step = BF ? ((long long)tc * nblk_step * k + j0) * 256 : (long long)t * ps * k * 64;
    const int nrows = BF ? 16 * nblk_cur : 2 * pairs_cur;
    float* sa = wsm + slot * a.stage;
    float* sx = sa + a_floats;
    int issued = 0;
#pragma unroll
    for (int mbi = 0; mbi < MBT; ++mbi) {
      const char* src = reinterpret_cast<const char*>(wf_tile + min(mbi, mb_last) * mb_stride + a_step);
      float* dst = sa + mbi * nqa_pad * 64;
      for (int u = 0; u < nquads; ++u) {
        lds_direct_b128(reinterpret_cast<const float*>(src + (size_t)u * 1024 + lane16), dst + u * 256);
        ++issued;
      }
    }
    const float* xr = xgrp + (long long)c0 * ch_stride;
    float* dstrow = sx;
    for (int r = 0; r < nrows; ++r) {
      const bool zero_row = r >= cbcur;
#pragma unroll
      for (int pi = 0; pi < F32_PMAX; ++pi) {
        if (pi >= pieces) break;
        const float* srcp = (zero_row || so[pi] < 0) ? g_zero_line + lane : xr + so[pi];
        if (pi * 64 + lane < xrow) lds_direct_b32(srcp, dstrow + pi * 64);
        ++issued;
      }
      xr += ch_stride;
      dstrow += xrow;
    }
    return issued;
  };

  if (wave_live) {
    int slot = 0;
    if (ks < nsteps) issue(ks, 0);
    for (int t = ks; t < nsteps; t += KS) {
      // the next step of this wave goes to the other slot (its previous contents were consumed one iteration ago)
      const int n_next = t + KS < nsteps ? issue(t + KS, slot ^ 1) : 0;
      wait_vmcnt_le(n_next);  // step t has landed; step t + KS may still be in flight
      if (BF) {
        const int tc = t / a.nch, j0 = (t - tc * a.nch) * a.tj, tjc = min(a.tj, k - j0);
        const int nkb = ((min(2 * ps, a.cin_g - tc * 2 * ps) + 15) >> 4) * tjc;
        bf_consume<MT, NT>(wsm, abase, xbase, slot * a.stage, 0, nkb, tjc, j0, d, xrow, acc);
        slot ^= 1;
        continue;
      }
      const int cbcur = min(2 * ps, a.cin_g - t * 2 * ps);
      const int nq = ((cbcur + 1) >> 1) * k;
      int j = 0;
      int off_a = slot * a.stage;
      int off_x = slot * a.stage;
      float fa[U][MT], fb[U][NT], ga[U][MT], gb[U][NT];
      auto load_group = [&](float (&da)[U][MT], float (&db)[U][NT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) da[u][mt] = wsm[abase[mt] + off_a];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) db[u][nt] = wsm[xbase[nt] + off_x];
          ++j; off_a += 64; off_x += d;
          if (j == k) { j = 0; off_x += 2 * xrow - k * d; }
        }
      };
      load_group(fa, fb);
      const int nfull = nq / U, rem = nq - nfull * U;
      for (int gi = 0; gi < nfull; ++gi) {
        load_group(ga, gb);
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][mt], fb[u][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) fa[u][mt] = ga[u][mt];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) fb[u][nt] = gb[u][nt];
        }
#pragma unroll
        for (int i = 0; i < U * MT * NT; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < U - 1; ++u)
        if (u < rem) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
              acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][mt], fb[u][nt], acc[mt][nt], 0, 0, 0);
        }
      slot ^= 1;
    }
  }

  // ---- partial tiles of the K slices meet in LDS; wave ks == 0 of every column slice adds and stores ----
  {
    float* R = smem;  // [KS-1][NWN][MT][NT][16][64]
    lds_barrier();
    if (ks > 0) {
      float* dst = R + ((ks - 1) * NWN + wn) * (MT * NT * 16 * 64) + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) dst[((mt * NT + nt) * 16 + r) * 64] = acc[mt][nt][r];
    }
    lds_barrier();
    if (ks > 0) return;
#pragma unroll 1
    for (int kk = 0; kk < KS - 1; ++kk) {
      const float* src = R + (kk * NWN + wn) * (MT * NT * 16 * 64) + lane;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mt][nt][r] += src[((mt * NT + nt) * 16 + r) * 64];
    }
  }
  with_act(a.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (col_b[nt] < 0) continue;
      float* ycol = a.y + (long long)col_b[nt] * a.t_out_total + (long long)col_to[nt] * a.out_stride + out_off;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        constexpr int EB = 4;  // (8 would push the 128-register kernels of this family over their occupancy step)
        // bias and previous values in batches of EB from clamped rows: a read under a per-element condition costs a drained
        // round trip each (64 in a row per lane); 0 / -0 stand in for an absent operand
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += EB) {
          int mrow[EB];
          float* dst[EB];
          float bv[EB], prev[EB];
#pragma unroll
          for (int e = 0; e < EB; ++e) {
            const int r = r0 + e;
            mrow[e] = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            dst[e] = ycol + (long long)(co0 + min(mrow[e], m_valid - 1)) * a.B * a.t_out_total;
            bv[e] = 0.f;
            prev[e] = -0.f;
          }
          if (a.bias) {
#pragma unroll
            for (int e = 0; e < EB; ++e) bv[e] = a.bias[co0 + min(mrow[e], m_valid - 1)];
          }
          if (a.accumulate) {
#pragma unroll
            for (int e = 0; e < EB; ++e) prev[e] = *dst[e];
          }
#pragma unroll
          for (int e = 0; e < EB; ++e) {
            const float v = conv_act<ACT>(a.bias ? acc[mt][nt][r0 + e] + bv[e] : acc[mt][nt][r0 + e], a.act_param);
            if (mrow[e] < m_valid) *dst[e] = prev[e] + v;
          }
        }
      }
    }
  });
}

struct F32Tile { int bm, bn, ks; };
static const F32Tile kTiles[] = {{128, 128, 1}, {64, 128, 2}, {64, 64, 4}, {32, 256, 2}, {32, 128, 4}};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

static int pick_tile(const ConvF32Args& a, int groups) {
  const int forced = -1;
  if (forced >= 0 && forced < kNumTiles) return forced;
  const long long n_total = (long long)a.B * a.n_out;
  auto blocks = [&](int i) {
    return ((n_total + kTiles[i].bn - 1) / kTiles[i].bn) * ((a.cout_g + kTiles[i].bm - 1) / kTiles[i].bm) * groups;
  };
  static const long long want = 384;  // two workgroups per CU before a larger tile is worth its reuse
  if (a.cout_g > 64) {
    if (blocks(0) >= want) return 0;
    if (blocks(1) >= want) return 1;
    return 2;
  }
  if (a.cout_g > 32) return blocks(1) >= want ? 1 : 2;
  return blocks(3) >= want ? 3 : 4;
}

static long long wfrag_elems(int c_out, int c_in, int k, int groups) {
  const int cout_g = c_out / groups, cin_g = c_in / groups;
  const long long f32 = (long long)groups * ((cout_g + 31) / 32) * ((cin_g + 1) / 2) * k * 64 + 4 * 64;  // + slack: the loader reads whole quads
  const long long bf = (long long)groups * ((cout_g + 31) / 32) * ((cin_g + 15) / 16) * k * 256;  // bf16 fragments (32-bit words)
  return std::max(f32, bf);
}

struct F32Plan { int ti; size_t lds; dim3 grid; bool wp; };

// Tile, step depth and LDS budget for a shape; fills the tiling fields of `a`.  nullptr = runnable, else the reason.
static const char* plan_conv_f32(ConvF32Args& a, int groups, F32Plan& pl) {
  if (a.cin_g <= 0 || a.cout_g <= 0 || a.k <= 0 || a.stride <= 0 || a.dil <= 0 || a.n_out <= 0 || a.B <= 0) return "bad shape";
  if ((long long)a.B * a.t_in >= (1LL << 29)) return "B*t_in >= 2^29";
  a.mblocks = (a.cout_g + 31) / 32;
  a.pairs = (a.cin_g + 1) / 2;
  a.cblocks = (a.cin_g + 15) / 16;
  a.tj = a.k; a.nch = 1;
  const int halo = (a.k - 1) * a.dil + 1;
  auto xrow_of = [&](int bn) {
    const int nmin = a.n_out_min > 0 ? a.n_out_min : a.n_out;
    const int items_max = (int)std::min<long long>(a.B, (bn + nmin - 2) / nmin + 1);
    return ((bn - 1) * a.stride + (items_max - 1) * std::max(halo - a.stride, 0) + halo) | 1;
  };
  static const bool use_wp = 1 != 0;  // wave-private rings for the K-split tiles (A/B switch)
  int ti = pick_tile(a, groups);
  if (a.bf) {
    const int ft = -1;
    if (ft >= 0 && ft < kNumTiles) ti = ft;
  }
  // wave-private rings (K-split tiles): every wave stages only its own column slice, but a workgroup holds 8 slots:
  // taken where one step of one wave fits 1/8 of the LDS budget; long kernels (the 41-tap scale-discriminator layers)
  // keep the shared ring
  auto wp_fits = [&](int t) {
    if (!use_wp || kTiles[t].ks <= 1) return false;
    const int xr = xrow_of(kTiles[t].bn / (4 / kTiles[t].ks));
    if (a.bf) {  // one 16-channel block with all its taps per wave step
      const size_t stb = (size_t)((kTiles[t].bm / 32) * a.k * 256 + 16 * xr + 4) * sizeof(float);
      return xr <= 64 * F32_PMAX && 8 * stb <= 78 * 1024;
    }
    const size_t st1 = (size_t)(((kTiles[t].bm / 32) * ((a.k + 3) & ~3) * 64 + 2 * xr + 2 * xr + 3) & ~3) * sizeof(float);
    return xr <= 64 * F32_PMAX && 8 * st1 <= 78 * 1024;
  };
  const bool wp = wp_fits(ti);
  if (!wp)  // shared ring, strided layers: narrower tiles until the staged row fits the loader (tiles are ordered wide -> narrow)
    while (xrow_of(kTiles[ti].bn) > 64 * F32_PMAX && ti != 2 && ti != kNumTiles - 1) ++ti;
  const int bm = kTiles[ti].bm, bn = kTiles[ti].bn, ks = kTiles[ti].ks;
  a.mtiles_per_group = (a.cout_g + bm - 1) / bm;
  a.xrow = xrow_of(wp ? bn / (4 / ks) : bn);
  a.pieces = (a.xrow + 63) / 64;
  if (a.pieces > F32_PMAX) return "input span too large (very short rows with a long kernel)";
  if (a.bf && a.cin_g < 16) return "fewer than 16 channels per group";  // half-empty K blocks: the fp32 kernel is faster
  if (a.bf) {
    // bf16-operand steps: ps = 8 * (16-channel blocks per step); long kernels split their taps over ring steps (tj per step)
    auto stage_bf = [&](int ps, int tj) { return ((bm / 32) * (ps / 8) * tj * 256 + 2 * ps * a.xrow + 4) & ~3; };
    auto lds_bf = [&](int ps, int tj, int nst) { return (size_t)(wp ? 8 : nst) * stage_bf(ps, tj) * sizeof(float); };
    const size_t two_wg = 78 * 1024, one_wg = 160 * 1024;
    int ps = 8, tj = a.k, nst = 2;
    if (lds_bf(8, a.k, 2) <= two_wg) {
      const int ps_cap = std::min(64, ((a.pairs + 7) / 8) * 8);
      // deepen the step while it fits two workgroups per CU; K-split tiles want at least two K blocks per wave and step
      while (ps * 2 <= ps_cap && lds_bf(ps * 2, a.k, 2) <= two_wg && (wp ? (a.pairs + 2 * ps - 1) / (2 * ps) >= ks : (ps < 16 || (ps / 8) * a.k < 2 * ks)))
        ps *= 2;
      if (!wp && lds_bf(ps, a.k, 3) <= two_wg) nst = 3;
    } else if (!wp && 0) {  // tap-chunked steps re-stage the input window per chunk: measured slower than fp32
      const size_t budget = lds_bf(8, 1, 2) <= two_wg ? two_wg : one_wg;
      while (tj > 1 && lds_bf(8, tj, 2) > budget) --tj;
      if (lds_bf(8, tj, 2) > budget) return "LDS budget";
      const int nch = (a.k + tj - 1) / tj;
      tj = (a.k + nch - 1) / nch;  // balanced chunks
    } else {
      return "LDS budget";
    }
    {  // tuning overrides (experiments)
      const int fps = 0, fnst = 0;
      if (fps >= 8 && fps % 8 == 0 && tj == a.k) ps = std::min(fps, ((a.pairs + 7) / 8) * 8);
      if (!wp && (fnst == 2 || fnst == 3)) nst = fnst;
    }
    a.ps = ps; a.tj = tj; a.nch = (a.k + tj - 1) / tj; a.nst = nst;
    a.stage = stage_bf(ps, tj);
    size_t lds = lds_bf(ps, tj, nst);
    lds = std::max(lds, (size_t)(ks - 1) * bm * bn * sizeof(float));
    lds = std::max(lds, (size_t)a.pieces * 64 * sizeof(int));
    if (lds > one_wg) return "LDS budget";
    pl.wp = wp;
    const long long n_total = (long long)a.B * a.n_out;
    if ((n_total + bn - 1) / bn > 0x7fffffffLL || groups * a.mtiles_per_group > 65535 || groups * a.mblocks > 65535) return "grid limits";
    pl.ti = ti;
    pl.lds = lds;
    pl.grid = dim3((unsigned)((n_total + bn - 1) / bn), groups * a.mtiles_per_group, 1);
    return nullptr;
  }
  auto stage_floats = [&](int ps) {
    return ((bm / 32) * ((ps * a.k + 3) & ~3) * 64 + 2 * ps * a.xrow + 2 * a.xrow + 3) & ~3;  // + slack for the read-ahead past a step
  };
  // shared rings: nst slots per workgroup; wave-private rings: two slots per wave
  auto lds_bytes = [&](int ps, int nst) { return (size_t)(wp ? 8 : nst) * stage_floats(ps) * sizeof(float); };
  if (lds_bytes(1, 2) > 160 * 1024) return "LDS budget";
  auto deepest = [&](int nst) {
    int ps = 1;
    while (ps < 16 && ps < a.pairs && lds_bytes(ps * 2, nst) <= 78 * 1024) ps *= 2;
    return ps;
  };
  // three slots hide the load latency best; two slots buy twice the step depth (half the per-step barrier / issue
  // overhead) when three would leave less than ~24 K pairs of work per wave and step
  int nst = 3, ps = deepest(3);
  if (wp || lds_bytes(ps, 3) > 78 * 1024 || (ps * a.k < 24 * ks && deepest(2) > ps)) { nst = 2; ps = deepest(2); }
  const int forced_nst = 0;
  if (!wp && (forced_nst == 2 || forced_nst == 3)) { nst = forced_nst; ps = deepest(nst); }
  const int forced_ps = 0;
  if (forced_ps > 0 && lds_bytes(forced_ps, nst) <= 160 * 1024) ps = forced_ps;
  a.nst = nst;
  a.ps = ps;
  a.stage = stage_floats(ps);
  size_t lds = lds_bytes(ps, nst);
  lds = std::max(lds, (size_t)(ks - 1) * bm * bn * sizeof(float));
  lds = std::max(lds, (size_t)a.pieces * 64 * sizeof(int));
  if (lds > 160 * 1024) return "LDS budget";
  pl.wp = wp;
  const long long n_total = (long long)a.B * a.n_out;
  if ((n_total + bn - 1) / bn > 0x7fffffffLL || groups * a.mtiles_per_group > 65535 || groups * a.mblocks > 65535) return "grid limits";
  pl.ti = ti;
  pl.lds = lds;
  pl.grid = dim3((unsigned)((n_total + bn - 1) / bn), groups * a.mtiles_per_group, 1);
  return nullptr;
}

static int dispatch_conv_f32(ConvF32Args a, const F32Plan& pl, hipStream_t stream) {
  const int ti = pl.ti;
  const size_t lds = pl.lds;
  dim3 grid = pl.grid;
  grid.z = a.phases;
  const int bm = kTiles[ti].bm, bn = kTiles[ti].bn, ks = kTiles[ti].ks;
  a.ablate = 0;
  static const int xcd_remap = 1;
  a.xcd_remap = xcd_remap;
  a.tl = nullptr;
  const bool want_tl = 0 != 0;
  if (want_tl) {
    EVMI_HIP_CHECK(hipMalloc(&a.tl, 24 * 4 * 4 * sizeof(long long)));
    EVMI_HIP_CHECK(hipMemsetAsync(a.tl, 0, 24 * 4 * 4 * sizeof(long long), stream));
  }

  static thread_local size_t configured_dev[kMaxDevices][4 * kNumTiles] = {};
  size_t* configured = configured_dev[device_slot()];
#define EVMI_F32_LAUNCH1(BM, BN, WM, WN, KS, BFM, IDX)                                                             \
  {                                                                                                                \
    if (lds > configured[IDX]) {                                                                                   \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)conv_cbt_f32_mfma_kernel<BM, BN, WM, WN, KS, BFM>,           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
      configured[IDX] = lds;                                                                                       \
    }                                                                                                              \
    hipLaunchKernelGGL((conv_cbt_f32_mfma_kernel<BM, BN, WM, WN, KS, BFM>), grid, dim3(256), lds, stream, a);      \
  }
#define EVMI_F32_LAUNCH(BM, BN, WM, WN, KS, IDX)                                                                   \
  {                                                                                                                \
    if (a.bf) EVMI_F32_LAUNCH1(BM, BN, WM, WN, KS, true, IDX + 2 * kNumTiles)                                      \
    else EVMI_F32_LAUNCH1(BM, BN, WM, WN, KS, false, IDX)                                                          \
  }
#define EVMI_F32_LAUNCH_WP1(BM, BNW, NWN, KS, BFM, IDX)                                                            \
  {                                                                                                                \
    if (lds > configured[IDX]) {                                                                                   \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)conv_cbt_f32_mfma_wp_kernel<BM, BNW, NWN, KS, BFM>,          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
      configured[IDX] = lds;                                                                                       \
    }                                                                                                              \
    hipLaunchKernelGGL((conv_cbt_f32_mfma_wp_kernel<BM, BNW, NWN, KS, BFM>), grid, dim3(256), lds, stream, a);     \
  }
#define EVMI_F32_LAUNCH_WP(BM, BNW, NWN, KS, IDX)                                                                  \
  {                                                                                                                \
    if (a.bf) EVMI_F32_LAUNCH_WP1(BM, BNW, NWN, KS, true, IDX + 2 * kNumTiles)                                     \
    else EVMI_F32_LAUNCH_WP1(BM, BNW, NWN, KS, false, IDX)                                                         \
  }
  if (pl.wp) {
    switch (ti) {
      case 1: EVMI_F32_LAUNCH_WP(64, 64, 2, 2, 5) break;
      case 2: EVMI_F32_LAUNCH_WP(64, 64, 1, 4, 6) break;
      case 3: EVMI_F32_LAUNCH_WP(32, 128, 2, 2, 7) break;
      default: EVMI_F32_LAUNCH_WP(32, 128, 1, 4, 8) break;
    }
  } else {
    switch (ti) {
      case 0: EVMI_F32_LAUNCH(128, 128, 2, 2, 1, 0) break;
      case 1: EVMI_F32_LAUNCH(64, 128, 1, 2, 2, 1) break;
      case 2: EVMI_F32_LAUNCH(64, 64, 1, 1, 4, 2) break;
      case 3: EVMI_F32_LAUNCH(32, 256, 1, 2, 2, 3) break;
      default: EVMI_F32_LAUNCH(32, 128, 1, 1, 4, 4) break;
    }
  }
#undef EVMI_F32_LAUNCH_WP
#undef EVMI_F32_LAUNCH_WP1
#undef EVMI_F32_LAUNCH
#undef EVMI_F32_LAUNCH1
  EVMI_LAUNCH_CHECK("conv_cbt_f32_mfma");
  if (want_tl) {  // stamps of workgroup (0,0): per step and wave: arrive, past barrier, loads issued, MFMAs done (100 MHz ticks)
    long long h[24 * 4 * 4];
    EVMI_HIP_CHECK(hipStreamSynchronize(stream));
    EVMI_HIP_CHECK(hipMemcpy(h, a.tl, sizeof(h), hipMemcpyDeviceToHost));
    EVMI_HIP_CHECK(hipFree(a.tl));
    fprintf(stderr, "[f32 timeline] tile %dx%d ks %d ps %d k %d xrow %d grid %u x %u\n", bm, bn, ks, a.ps, a.k, a.xrow, grid.x, grid.y);
    const long long t0 = h[0];
    for (int t = 0; t < 24 && h[(t * 4) * 4]; ++t) {
      fprintf(stderr, "  step %2d:", t);
      for (int w = 0; w < 4; ++w) {
        const long long* e = h + (t * 4 + w) * 4;
        fprintf(stderr, "  w%d %6lld +%4lld +%4lld +%5lld", w, e[0] - t0, e[1] - e[0], e[2] - e[1], e[3] - e[2]);
      }
      fprintf(stderr, "\n");
    }
  }
  return EVMI_OK;
}

static ConvDirectArgs direct_args(int B, int c_in, int t_in, int c_out, int t_out_total, int n_out, int k, int stride, int pad,
                                  int dil) {
  ConvDirectArgs d = {};
  d.B = B; d.t_in = t_in; d.t_out_total = t_out_total; d.n_out = n_out; d.c_in = c_in; d.c_out = c_out; d.k = k;
  d.stride = stride; d.dil = dil; d.pad = pad; d.out_stride = 1;
  return d;
}

int launch_conv_cbt_f32_mfma(ConvF32Args a, const float* w, float* wfrag_ws, long long wfrag_ws_elems, int groups,
                             hipStream_t stream) {
  F32Plan pl;
  if (const char* why = plan_conv_f32(a, groups, pl)) {
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv_cbt_f32_mfma: ") + why);
  }
  const long long wf_total = wfrag_elems(a.cout_g * groups, a.cin_g * groups, a.k, groups);
  if (!wfrag_ws || wfrag_ws_elems < wf_total || (reinterpret_cast<uintptr_t>(wfrag_ws) & 15))
    return fail(EVMI_ERR_INVALID_ARG, "conv_cbt_f32_mfma: weight-fragment workspace missing, too small or unaligned");
  if (a.bf)
    hipLaunchKernelGGL(wfrag_bf16_kernel, dim3(a.cblocks, groups * a.mblocks), dim3(256), 0, stream, w,
                       reinterpret_cast<unsigned*>(wfrag_ws), a.cout_g, a.cin_g, a.k, a.mblocks, a.cblocks);
  else
    hipLaunchKernelGGL(wfrag_kernel, dim3(a.pairs, groups * a.mblocks), dim3(256), 0, stream, w, wfrag_ws, a.cout_g, a.cin_g,
                       a.k, a.mblocks, a.pairs);
  a.wf = wfrag_ws;
  a.phases = 1; a.wf_phase_stride = 0;
  a.ph_pad[0] = a.pad; a.ph_nout[0] = a.n_out; a.ph_off[0] = a.out_offset;
  return dispatch_conv_f32(a, pl, stream);
}

// Input gradient of y = conv1d(x, w, stride, pad, dil, groups): dx [c_in][B][t_in] from dy [c_out][B][t_out], all phases in
// one launch (grid.z).  Strided layers must not be dilated.
static const char* plan_dgrad(int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil, int groups,
                              ConvF32Args& a, F32Plan& pl, int bf = 0) {
  a.bf = bf;
  if (groups <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups) return "bad shape";
  if (stride > 1 && dil != 1) return "strided and dilated";
  if (stride > 8) return "stride above 8";
  const int cin_g = c_in / groups, cout_g = c_out / groups;
  if (cin_g <= SMALLCO_DIRECT || cout_g == 1) return "direct-kernel shape";  // GEMV / outer product: conv_cbt_direct.hip
  const int phases = std::min(stride, k), M = (k + stride - 1) / stride;
  a.B = B; a.t_in = t_out; a.t_out_total = t_in;
  a.cin_g = cout_g; a.cout_g = cin_g; a.k = M; a.stride = 1; a.dil = stride == 1 ? dil : 1;
  a.out_stride = stride; a.accumulate = 0; a.bias = nullptr; a.phases = phases;
  int n_max = 0, n_min = 0;
  for (int phi = 0; phi < phases; ++phi) {
    if (stride == 1) {
      a.ph_pad[0] = dil * (k - 1) - pad; a.ph_nout[0] = t_in; a.ph_off[0] = 0;
    } else {
      const int num = pad - phi;  // first q with stride*q + phi - pad >= 0
      const int q0 = num > 0 ? (num + stride - 1) / stride : 0;
      const int q_hi = (t_in - 1 + pad - phi) >= 0 ? (t_in - 1 + pad - phi) / stride : -1;
      a.ph_pad[phi] = (M - 1) - q0;
      a.ph_nout[phi] = std::max(0, q_hi - q0 + 1);
      a.ph_off[phi] = stride * q0 + phi - pad;
    }
    n_max = std::max(n_max, a.ph_nout[phi]);
    if (a.ph_nout[phi] > 0) n_min = n_min == 0 ? a.ph_nout[phi] : std::min(n_min, a.ph_nout[phi]);
  }
  if (n_max <= 0) return "no outputs";
  a.n_out = n_max; a.n_out_min = n_min; a.pad = 0; a.out_offset = 0;
  return plan_conv_f32(a, groups, pl);
}

int launch_conv_dgrad_f32_mfma(const float* dy, const float* w, float* dx, float* ws, long long ws_elems, int B, int c_in, int t_in,
                               int c_out, int t_out, int k, int stride, int pad, int dil, int groups, hipStream_t stream, int bf = 0) {
  ConvF32Args a = {};
  F32Plan pl;
  if (const char* why = plan_dgrad(B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups, a, pl, bf))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv_dgrad_f32_mfma: ") + why);
  const long long per_phase = wfrag_elems(c_in, c_out, a.k, groups);  // dy channels in, x channels out
  if (!ws || ws_elems < per_phase * a.phases || (reinterpret_cast<uintptr_t>(ws) & 15))
    return fail(EVMI_ERR_INVALID_ARG, "conv_dgrad_f32_mfma: workspace missing, too small or unaligned");
  if (a.pairs > 65535) return fail(EVMI_ERR_UNSUPPORTED, "conv_dgrad_f32_mfma: grid limits");
  if (bf)
    hipLaunchKernelGGL(wfrag_dgrad_bf16_kernel, dim3(a.cblocks, groups * a.mblocks, a.phases), dim3(256), 0, stream, w,
                       reinterpret_cast<unsigned*>(ws), c_out / groups, c_in / groups, k, stride, a.k, a.mblocks, a.cblocks, per_phase);
  else
    hipLaunchKernelGGL(wfrag_dgrad_kernel, dim3(a.pairs, groups * a.mblocks, a.phases), dim3(256), 0, stream, w, ws, c_out / groups,
                       c_in / groups, k, stride, a.k, a.mblocks, a.pairs, per_phase);
  a.x = dy; a.y = dx; a.wf = ws; a.wf_phase_stride = per_phase;
  return dispatch_conv_f32(a, pl, stream);
}

}  // namespace evmi

using namespace evmi;

extern "C" {

long long evmi_conv1d_cbt_f32_ws_elems(int B, int c_in, int c_out, int n_out, int k, int groups) {
  if (groups <= 0 || c_out <= 0 || c_in <= 0 || k <= 0 || B <= 0 || n_out <= 0 || c_in % groups || c_out % groups) return -1;
  int cc, nchunks;
  const long long direct = conv_direct_plan(direct_args(B, c_in, 0, c_out, n_out, n_out, k, 1, 0, 1), groups, cc, nchunks);
  if (direct > 0) return direct;  // scratch of the few-output-channel kernel (+1: never 0)
  return wfrag_elems(c_out, c_in, k, groups);
}

int evmi_conv1d_cbt_f32_supported(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int dil, int groups) {
  if (groups <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups) return 0;
  ConvF32Args a = {};
  a.B = B; a.t_in = t_in; a.n_out = n_out; a.cin_g = c_in / groups; a.cout_g = c_out / groups; a.k = k; a.stride = stride; a.dil = dil;
  int cc, nchunks;
  if (conv_direct_plan(direct_args(B, c_in, t_in, c_out, n_out, n_out, k, stride, 0, dil), groups, cc, nchunks) > 0) return 1;
  F32Plan pl;
  return plan_conv_f32(a, groups, pl) == nullptr ? 1 : 0;
}

long long evmi_conv1d_dgrad_cbt_f32_ws_elems(int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil,
                                             int groups) {
  ConvF32Args a = {};
  F32Plan pl;
  if (plan_dgrad(B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups, a, pl)) return 0;
  return wfrag_elems(c_in, c_out, a.k, groups) * a.phases;
}

int evmi_conv1d_dgrad_cbt_f32(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev, long long ws_elems, int B,
                              int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil, int groups,
                              void* stream) {
  if (!dy_dev || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_f32: null pointer");
  return launch_conv_dgrad_f32_mfma(dy_dev, w_dev, dx_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups,
                                    (hipStream_t)stream);
}

int evmi_conv1d_dgrad_cbt_bf16(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev, long long ws_elems, int B,
                               int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil, int groups,
                               void* stream) {
  if (!dy_dev || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16: null pointer");
  {  // shapes the bf16-operand staging does not fit run on the fp32 matrix cores
    ConvF32Args a = {};
    F32Plan pl;
    if (plan_dgrad(B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups, a, pl, 1))
      return launch_conv_dgrad_f32_mfma(dy_dev, w_dev, dx_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, t_out, k, stride, pad, dil,
                                        groups, (hipStream_t)stream, 0);
  }
  return launch_conv_dgrad_f32_mfma(dy_dev, w_dev, dx_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups,
                                    (hipStream_t)stream, 1);
}

static int conv1d_cbt_any(int bf, const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev,
                        float* wfrag_ws_dev, long long wfrag_ws_elems, int B, int c_in, int t_in, int c_out,
                        int t_out_total, int n_out, int k, int stride, int pad, int dil, int groups, int out_stride,
                        int out_offset, int accumulate, int act, float act_param, void* stream) {
  if (!x_dev || !w_dev || !y_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_f32: null pointer");
  if (groups <= 0 || c_in % groups || c_out % groups) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_f32: groups");
  {  // GEMV / outer-product shapes: direct kernels (conv_cbt_direct.hip)
    ConvDirectArgs d = direct_args(B, c_in, t_in, c_out, t_out_total, n_out, k, stride, pad, dil);
    d.x = x_dev; d.w = w_dev; d.bias = bias_dev; d.y = y_dev; d.out_stride = out_stride; d.out_offset = out_offset;
    d.accumulate = accumulate; d.act = act; d.act_param = act_param;
    int cc, nchunks;
    if (conv_direct_plan(d, groups, cc, nchunks) > 0)
      return launch_conv_direct(d, groups, wfrag_ws_dev, wfrag_ws_elems, (hipStream_t)stream);
  }
  ConvF32Args a = {};
  a.x = x_dev; a.wf = nullptr; a.bias = bias_dev; a.y = y_dev;
  a.B = B; a.t_in = t_in; a.t_out_total = t_out_total; a.n_out = n_out;
  a.cin_g = c_in / groups; a.cout_g = c_out / groups; a.k = k; a.stride = stride; a.dil = dil; a.pad = pad;
  a.out_stride = out_stride; a.out_offset = out_offset; a.accumulate = accumulate; a.mtiles_per_group = 1;
  a.act = act; a.act_param = act_param;
  if (act < 0 || act > 4 || (act && accumulate)) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_f32: activation (0..4, not with accumulate)");
  if (bf) {  // shapes the bf16-operand staging does not fit run on the fp32 matrix cores
    ConvF32Args probe = a;
    probe.bf = 1;
    F32Plan pl;
    if (plan_conv_f32(probe, groups, pl) == nullptr) a.bf = 1;
  }
  return launch_conv_cbt_f32_mfma(a, w_dev, wfrag_ws_dev, wfrag_ws_elems, groups, (hipStream_t)stream);
}

int evmi_conv1d_cbt_f32(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev,
                        float* wfrag_ws_dev, long long wfrag_ws_elems, int B, int c_in, int t_in, int c_out,
                        int t_out_total, int n_out, int k, int stride, int pad, int dil, int groups, int out_stride,
                        int out_offset, int accumulate, int act, float act_param, void* stream) {
  return conv1d_cbt_any(0, x_dev, w_dev, bias_dev, y_dev, wfrag_ws_dev, wfrag_ws_elems, B, c_in, t_in, c_out, t_out_total, n_out, k,
                        stride, pad, dil, groups, out_stride, out_offset, accumulate, act, act_param, stream);
}

int evmi_conv1d_cbt_bf16_rounds(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups) {
  if (groups <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups) return 0;
  if (evmi_conv1d_cbt_bf16pk_ws_elems(B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups) > 0) return 1;  // packed kernel
  {  // GEMV / outer-product shapes stay exact (direct kernels)
    ConvDirectArgs d = direct_args(B, c_in, t_in, c_out, n_out, n_out, k, stride, pad, dil);
    int cc, nchunks;
    if (conv_direct_plan(d, groups, cc, nchunks) > 0) return 0;
  }
  ConvF32Args a = {};
  a.B = B; a.t_in = t_in; a.t_out_total = n_out; a.n_out = n_out;
  a.cin_g = c_in / groups; a.cout_g = c_out / groups; a.k = k; a.stride = stride; a.dil = dil; a.pad = pad;
  a.out_stride = 1; a.mtiles_per_group = 1; a.bf = 1;
  F32Plan pl;
  return plan_conv_f32(a, groups, pl) == nullptr ? 1 : 0;  // the in-LDS rounding mode of the fp32 kernel takes it
}

int evmi_conv1d_cbt_bf16(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev,
                         float* wfrag_ws_dev, long long wfrag_ws_elems, int B, int c_in, int t_in, int c_out,
                         int t_out_total, int n_out, int k, int stride, int pad, int dil, int groups, int out_stride,
                         int out_offset, int accumulate, int act, float act_param, void* stream) {
  return conv1d_cbt_any(1, x_dev, w_dev, bias_dev, y_dev, wfrag_ws_dev, wfrag_ws_elems, B, c_in, t_in, c_out, t_out_total, n_out, k,
                        stride, pad, dil, groups, out_stride, out_offset, accumulate, act, act_param, stream);
}

}  // extern "C"
