// bf16-operand implicit-GEMM 1-D convolution for the channel-major training layout, on a PACKED copy of the input.
//
//   y[co][b][to*os + oo] (+)= act(bias[co] + sum_{ci in group} sum_{j<k} bf16(w[co][ci][j]) * bf16(x[ci][b][to*s + j*d - p]))
//
// The fp32 tensors stay [C][B][T] in HBM (fp32 master weights, fp32 activations, fp32 accumulation).  What changes against
// conv_cbt_f32_mfma.hip is how the operands reach v_mfma_f32_32x32x16_bf16, whose B operand wants 8 consecutive K values per
// lane for ONE column -- 8 channels of one time step, which are B*T floats apart in the channel-major layout:
//
//  * pack_x_kernel (one read of x, half a write): xp[g][octet][b][u] = 16-byte unit of the 8 channels 8*octet..+7 of group g
//    at padded position u (u = t + PL; zeros in the padding and in channels past the group), every item a segment of Tp
//    units.  A tile's input window -- several short items or a slice of a long one, taps, strides and padding included --
//    is then ONE contiguous range of units per octet row: all loads are full 1 KB global_load_lds_dwordx4, no masks, no
//    per-lane tables, and a B fragment is one ds_read_b128 at (column * stride + tap * dilation).
//  * K runs over "halves" h = (octet, tap) in that order; one MFMA K block = halves (2q, 2q+1): the lanes of the lower
//    half-wave take half 2q, the upper ones half 2q+1 (an odd tail is zero weights).  So any group width that is a multiple
//    of 8 channels fills the K blocks exactly (the grouped 41-tap layers of the scale discriminators have 8..64).
//  * wfrag_pk_kernel re-lays the weights per call as 1 KB A fragments [m-block][K block][lane][8 bf16].
//  * LDS ring of 2-3 slots per workgroup (weights + window rows of a step), LDS-direct loads one or two steps ahead, one
//    barrier per step; every wave owns a 64x64 / 32x64 / 32x32 accumulator tile.  Tiles are XCD-ordered (m-tile major).
//  * The input gradient of a strided layer runs its polyphase components in one launch (grid.z), as in the fp32 kernel.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "conv_cbt_direct.h"
#include "conv_pk_common.h"

namespace evmi {

struct ConvPkArgs {
  const uint4* xp;     // packed input [groups][octs][B][Tp] units (+ slack)
  const uint4* wf;     // fragments [phase][groups*mblocks][kblocks][64] units
  const int2* tab;     // per K block: window offsets (units) of its two halves, relative to the first octet row of its step
  const float* bias;   // [c_out] or nullptr
  float* y;            // [c_out][B][t_out_total]
  int B, Tp, t_out_total;
  int cout_g, k, stride, dil;
  int octs, kblocks;   // ceil(cin_g / 8), ceil(octs * k / 2)
  int kb_step;         // K blocks per ring step
  int rows_step;       // octet rows a step may span
  int xrow, pieces;    // LDS row stride in units (= 64 * pieces)
  int nst;             // ring slots
  int mblocks, mtiles_per_group;
  int out_stride, accumulate, act;
  float act_param;
  int phases;
  long long wf_phase_stride;  // units
  int ph_shift[8], ph_nout[8], ph_off[8];
  int xcd_remap, xcd_hb;  // xcd_hb: m-tiles per band of the XCD-ordered tile list (see the kernel)
  // split-K: grid.x = column tiles * ksplit; split sp takes the ring steps [sp * steps_per_split, ...) and stores its raw
  // accumulators to part[sp][phase][c_out][part_ld] (part_ld = B * longest phase); conv_pk_reduce_kernel adds them in split
  // order + epilogue
  int ksplit, steps_per_split, ntiles_n;
  float* part;
  long long part_stride, part_ld;
  // (Folding the reduce into the last-arriving workgroup of a tile -- partial store, __threadfence, atomic counter -- was built and
  // measured: 91 -> 160 us on the 1024 -> 1024 k = 5 layer, 83 -> 338 us with four splits.  A device-scope release on this
  // multi-XCD part writes the XCD's L2 back and invalidates it in EVERY workgroup; the kernel boundary in front of a reduce launch
  // does that once.  The separate reduce pass stays.)
  // fused epilogue tail, in this order: v = act(acc + bias); v *= (out_mask > 0 ? 1 : out_mask_slope); v += res
  //   out_mask: a tensor of y's shape -- the INPUT of the leaky ReLU in front of the convolution whose input gradient this
  //             launch computes (the activation backward without a separate pass);  res: a tensor of y's shape added to the
  //             result (the residual connection in a forward pass, the skip path's gradient in a backward pass)
  const float* out_mask;
  float out_mask_slope;
  const float* res;
  // ... and, with drop_p > 0, v is multiplied by (keep(drop_seed, element index in y) ? drop_fac : 0) in front of `+ res`:
  // y = res + s * dropout(act(conv + bias), p) with drop_fac = s / (1 - p) -- the residual add + dropout behind a Conformer
  // sub-layer's last dense layer in its epilogue (the mask stream of evmi_dropout_fused_f32: same seed, same element index)
  float drop_p, drop_fac;
  SeedArg drop_seed;
  // Flat packed output (the discriminator chains, csrc/disc_chain.hip): po.y != nullptr -- the tile is written as 16-byte units of
  // 8 bf16 channels into a packed tensor [channel octet][unit] (rows `plane` units apart), the layout the next layer's loads read.
  // Column n of phase ph is unit w = n * out_stride + ph_off[ph] of the COMPUTE geometry: items Tc units apart, of which the first
  // `valid` are outputs; unit (b, u) = (w / Tc, w % Tc) is stored at b * Ts + u (another item pitch: the consumer's), the other
  // columns are dropped (the gaps of a packed tensor stay zero: buffers are zeroed once, kernels write valid units only).
  //   value = act(acc + bias);  (+ fm_scale * sign(xf - xr): the feature-matching gradient, xf = mask, xr = fm);  * lrelu'(mask)
  // mask / fm: packed tensors of the output's shape (item pitch Tm, rows mplane apart).
  struct FlatOut {
    uint4* y;
    long long plane;
    int Tc, valid, Ts;
    const uint4* mask;
    const uint4* fm;
    long long mplane;
    int Tm;
    float mask_slope, fm_scale;
    // The wide middle tensor of a feed-forward block, kept packed in both directions (the FastSpeech2 step; flat_tail_silu):
    //   tail 1: y <- bf16(v) (the pre-activation, read again by the backward) and y2 <- bf16(dropout(silu(v), drop_p)) (the packed
    //           input of the block's second layer), v = act(acc + bias)
    //   tail 2: y <- bf16(dropout(v, drop_p) * silu'(pre)), pre = the packed pre-activation tail 1 stored, passed in `fm` (rows
    //           mplane apart, item pitch Tm; `mask` stays null): the first layer's packed output gradient, formed in the epilogue of
    //           the second layer's input gradient
    //   tail 3: y <- bf16(dropout(silu(v), drop_p)) alone (inference, drop_p = 0: the activated tensor is all the second layer reads)
    // Mask stream and arithmetic of PackArgs::fuse 1 / 2 (drop_seed; element index = channel * drop_ld + unit).
    int tail;
    uint4* y2;
    long long drop_ld;
    // what a pack pass would have zeroed, written by the same epilogue (no memset launches in front of the call): the units of columns
    // [valid, pad_end) of every row (a single item's pitch rounded up: conv_pk_common.h) and zero_n units at zero_p (the slack behind
    // the tensor).  Tails only.
    int pad_end;
    uint4* zero_p;
    int zero_n;
  } po;
};

// The feed-forward tails of a flat packed output for four channels of one unit: channels c .. c + 3 of column n, element index
// (c + e) * ld + n.  With ld even the two elements of a dropout pair (common.h) are columns n, n ^ 1 of one channel -- neighbouring
// lanes, same register.  LANES (ld even, all lanes live: the matrix kernel's epilogue): each lane hashes the two channels of its own
// parity and exchanges with its neighbour; otherwise one hash per element (odd ld; the split-K reduce pass).
template <int TAIL, bool LANES>
__device__ __forceinline__ void pk_flat_tail_silu(float& v0, float& v1, float& v2, float& v3, float& s0, float& s1, float& s2, float& s3,
                                                  uint2 pre, unsigned long long dseed, float p_drop, unsigned long long c, unsigned long long n,
                                                  unsigned long long ld) {
  // (hardware exp2 / reciprocal, a multiply for the 1 / (1 - p) scale: the results are rounded to bf16 on the spot -- their last bits
  // do not survive it -- and this tail is ~40 vector instructions per element in the epilogue of a matrix kernel)
  const float rk = 1.f / (1.f - p_drop);
  const float z[4] = {bf16_lo(pre.x), bf16_hi(pre.x), bf16_lo(pre.y), bf16_hi(pre.y)};
  float* vp[4] = {&v0, &v1, &v2, &v3};
  float* sp[4] = {&s0, &s1, &s2, &s3};
  float u[4];
  if (p_drop <= 0.f) {  // (wave-uniform: inference -- no mask, no hash)
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = 1.f;
  } else if (LANES) {
    const unsigned par = (unsigned)n & 1u, ldh = (unsigned)(ld >> 1);
    const unsigned j0 = (unsigned)c * ldh + (unsigned)(n >> 1);  // (below 2^32: ffn_tail_check)
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const unsigned mine = dropout_hash(dseed, j0 + (2u * c2 + par) * ldh, 0u);
      const unsigned other = (unsigned)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);  // quad_perm [1, 0, 3, 2]: lane ^ 1
      u[2 * c2] = dropout_u16(par ? other : mine, par);
      u[2 * c2 + 1] = dropout_u16(par ? mine : other, par);
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) u[e] = uniform01(dseed, (c + (unsigned long long)e) * ld + n);
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool keep = u[e] >= p_drop;
    if (TAIL == 1) {
      const float val = *vp[e] * __builtin_amdgcn_rcpf(1.f + __expf(-*vp[e]));
      *sp[e] = keep ? val * rk : 0.f;
    } else {
      const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-z[e]));
      const float d = keep ? *vp[e] * rk : 0.f;
      *vp[e] = d * (sg * (1.f + z[e] * (1.f - sg)));
    }
  }
}
template <int ACT>
__device__ __forceinline__ float pk_act(float v, float p) {
  if (ACT == 1) return v > 0.f ? v : v * p;
  if (ACT == 2) return v / (1.f + expf(-v));
  if (ACT == 3) return fmaxf(v, 0.f);
  if (ACT == 4) return tanhf(v);
  return v;
}
// old + ((v * fac) + add) with every operation rounded by itself: hipcc contracts a * b + c into one fma by default, and the
// un-fused sequence of kernels this tail replaces (scale, then add, then accumulate) rounds three times
__device__ __forceinline__ float pk_tail(float v, float fac, float add, float old) {
#pragma clang fp contract(off)
  const float p = v * fac;
  const float q = p + add;
  return old + q;
}

template <class F>
__device__ __forceinline__ void pk_with_act(int act, F&& body) {
  switch (act) {
    case 1: body(std::integral_constant<int, 1>{}); break;
    case 2: body(std::integral_constant<int, 2>{}); break;
    case 3: body(std::integral_constant<int, 3>{}); break;
    case 4: body(std::integral_constant<int, 4>{}); break;
    default: body(std::integral_constant<int, 0>{}); break;
  }
}

// A fragments.  mode 0 (forward): rows = output channels of w [c_out][cin_g][k], K channels = input channels, tap j.
// mode 1 (input gradient, phase phi of `stride`): rows = INPUT channels, K channels = output channels, M taps per phase
// (phases with fewer taps zero padded in front), tap m -> j = phi + stride * (m_phi - 1 - (m - lead)): see wfrag_dgrad_kernel.
// wf[ph][(g*MB + mb)][q][lane] (uint4): lane (mi = lane & 31, kh = lane >> 5), half h = 2q + kh = (octet, tap).
// grid (kblocks, groups*MB, phases), 256 threads = 64 lanes x 4 words
struct WfragArgs {
  const float* w;
  unsigned* wf;
  int rows_g, kch_g, kt, MB, octs, kblocks, mode, k_full, stride;
  long long phase_stride_words;
  int2* tab;
  int kb_step, xrow, dil;
  int gx, gy, gz;  // logical grid (kblocks, groups*MB, phases)
};
__device__ __forceinline__ void wfrag_pk_block(const WfragArgs& f, int q, int gmb, int phi) {
  const int kt = f.kt;
  if (gmb == 0 && phi == 0 && threadIdx.x == 0) {  // offsets of the K block's halves in the staged window of its ring step
    const int o_lo = (2 * (q / f.kb_step) * f.kb_step) / kt;
    const int h0 = 2 * q, h1 = h0 + 1 < f.octs * kt ? h0 + 1 : h0;  // odd tail: zero weights, any staged unit
    f.tab[q] = make_int2((h0 / kt - o_lo) * f.xrow + (h0 % kt) * f.dil, (h1 / kt - o_lo) * f.xrow + (h1 % kt) * f.dil);
  }
  const int g = gmb / f.MB, mb = gmb - g * f.MB;
  const int lane = threadIdx.x >> 2, wd = threadIdx.x & 3;
  const int mi = lane & 31, kh = lane >> 5;
  const int h = 2 * q + kh;
  const int o = h / kt, j = h - o * kt;
  const int row = mb * 32 + mi;
  float v[2] = {0.f, 0.f};
  if (o < f.octs && row < f.rows_g) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kc = o * 8 + 2 * wd + e;
      if (kc >= f.kch_g) continue;
      if (f.mode == 0) {
        v[e] = f.w[((long long)(g * f.rows_g + row) * f.kch_g + kc) * kt + j];
      } else {  // rows_g = cin_g (x channels), kch_g = cout_g (dy channels), w [c_out][cin_g][k_full]
        const int m_phi = (f.k_full - phi + f.stride - 1) / f.stride, lead = kt - m_phi;
        if (j >= lead) {
          const int jj = phi + f.stride * (m_phi - 1 - (j - lead));
          v[e] = f.w[((long long)(g * f.kch_g + kc) * f.rows_g + row) * f.k_full + jj];
        }
      }
    }
  }
  f.wf[phi * f.phase_stride_words + (((long long)gmb * f.kblocks + q) * 64 + lane) * 4 + wd] = pk_bf16x2(v[0], v[1]);
}
// Both preparation passes of a convolution call in ONE launch: blocks [0, n_pack) pack the input, the rest re-lay the weights
// (the two are independent; at 5-7 us of fixed cost per launch and ~350 convolution calls per GAN step the second launch was
// 2.5 ms of the step).
__global__ __launch_bounds__(256) void prep_pk_kernel(PackArgs p, WfragArgs f) {
  const unsigned n_pack = (unsigned)p.gx * p.gy * p.gz;
  if (blockIdx.x < n_pack) {
    pack_x_flat(p, blockIdx.x);
  } else {
    const unsigned b = blockIdx.x - n_pack;
    const unsigned q = b % f.gx, r = b / f.gx;
    wfrag_pk_block(f, (int)q, (int)(r % f.gy), (int)(r / f.gy));
  }
}

// ADIR: the weight fragments never pass through the LDS.  A fragment is already laid out per lane (wfrag: [K block][lane][8 bf16]),
// so a wave fetches the fragments of ITS rows for the next ring step straight into registers (two sets of PK_ADIR_KBS x MT
// fragments, alternating) while the current step computes.  Why: every tile shape and a K loop with a third of the vector
// instructions ran the 1024-channel layers at the same ~380 TFLOP/s (tools/pkflat_bench.py) -- what did not change between them
// was the number of 1 KB LDS-direct loads per workgroup and step (weights 2/3 of them), and those are issued at one per ~40-100
// cycles per CU while the LDS feeds the matrix cores.  With the weights off that path the LDS holds input windows
// only (a third of the bytes) and serves half the fragment reads.
constexpr int PK_ADIR_KBS = 5;
// TAILS: the instantiations that carry the feed-forward tails of a flat packed output (FlatOut::tail; their per-element hash and
// exponentials are ~10 k instructions of epilogue the other launches should neither fetch nor allocate registers for)
template <int BM, int BN, int WM, int WN, bool ADIR = false, bool TAILS = false>
__global__ __launch_bounds__(WM * WN * 64, 2) void conv_pk_kernel(ConvPkArgs a) {
  static_assert(WM * WN == 4 || WM * WN == 8, "four or eight waves");
  constexpr int NW = WM * WN;  // eight-wave tiles: the loads of a ring step are issued by twice the waves (an LDS-direct load costs its
                               // wave ~100 cycles of issue while the LDS feeds MFMAs), and a CU holds 16 waves instead of 8
  constexpr int MT = BM / (WM * 32), NT = BN / (WN * 32), MBT = BM / 32;
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int kh = lane >> 5, ln = lane & 31;

  unsigned bx = blockIdx.x, by = blockIdx.y;
  if (a.xcd_remap) {
    // Workgroups go to the eight XCDs (private L2s) round robin; every XCD takes a contiguous run of the tile list.  The list is
    // ordered in BANDS of xcd_hb m-tiles, column by column inside a band, so a run is a compact block of tiles -- xcd_hb m-tiles
    // by (columns / runs per band) n-tiles -- whose weight rows and window columns are each fetched into that L2 once and shared.
    // (xcd_hb = 1 is the plain m-tile-major list: a run is one m-tile across many columns, i.e. every XCD pulls ALL the input
    // windows through its L2 -- eight copies of the activations over the fabric, the traffic the 1024-channel layers were bound by.)
    const unsigned nwg = gridDim.x * gridDim.y, orig = blockIdx.x + gridDim.x * blockIdx.y;
    if (nwg >= 16) {
      const unsigned q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
      const unsigned L = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
      const unsigned hb = (unsigned)a.xcd_hb, band = L / (hb * gridDim.x), rem = L - band * hb * gridDim.x;
      const unsigned hcur = min(hb, gridDim.y - band * hb);
      bx = rem / hcur;
      by = band * hb + (rem - bx * hcur);
    }
  }
  const int g = by / a.mtiles_per_group, mt_idx = by % a.mtiles_per_group;
  const int co0 = g * a.cout_g + mt_idx * BM;
  const int k = a.k, s = a.stride, xrow = a.xrow, pieces = a.pieces, kbs = a.kb_step;
  const int ph = blockIdx.z;
  const int n_out = a.ph_nout[ph], shift = a.ph_shift[ph], out_off = a.ph_off[ph];
  if (n_out <= 0) return;
  const long long n_total = (long long)a.B * n_out;
  const int split = a.ksplit > 1 ? (int)(bx / a.ntiles_n) : 0;
  if (a.ksplit > 1) bx -= split * a.ntiles_n;
  const long long n0 = (long long)bx * BN;
  if (n0 >= n_total) return;
  const int b_first = (int)(n0 / n_out);
  const int to_first = (int)(n0 - (long long)b_first * n_out);
  const int m_valid = min(BM, a.cout_g - mt_idx * BM);
  const long long plane = (long long)a.B * a.Tp;  // units per octet row
  const uint4* xwin = a.xp + (long long)g * a.octs * plane + (long long)b_first * a.Tp + (long long)to_first * s + shift;
  const int a_units = ADIR ? 0 : MBT * kbs * 64;
  const int stage = a_units + a.rows_step * xrow;
  // Stride-2 / stride-4 layers read units (column * stride + tap): the 16 lanes of a ds_read_b128 group then share 8 / 4 of the
  // 16 sixteen-byte slots of the bank row (2- / 4-way conflicts: 21-32 % of the LDS cycles of the scale discriminators' layers).
  // Position p of a staged row holds unit p ^ ((p >> 4) & swz): consecutive 16-unit blocks are rotated against each other, a
  // group's units land on 16 distinct slots.  Applied on the source side of the LDS-direct loads and again on the reads (rows
  // are multiples of 64 units, so the row offset of a read does not disturb the block index).
  const int swz = s == 2 ? 1 : (s == 4 ? 3 : 0);

  int colu[NT], col_b[NT], col_to[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (wn * NT + nt) * 32 + ln;
    const long long n = n0 + c;
    if (n < n_total) {
      const int bb = (int)(n / n_out);
      col_b[nt] = bb;
      col_to[nt] = (int)(n - (long long)bb * n_out);
      colu[nt] = (bb - b_first) * a.Tp + (col_to[nt] - to_first) * s;  // window-relative unit of the column (tap 0)
    } else {
      col_b[nt] = -1;
      col_to[nt] = 0;
      colu[nt] = 0;  // staged data; the column is never stored
    }
  }
  int abase[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) abase[mt] = (wm * MT + mt) * kbs * 64 + lane;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nsteps_all = (a.kblocks + kbs - 1) / kbs;
  const int t_lo = a.ksplit > 1 ? split * a.steps_per_split : 0;
  const int nsteps = a.ksplit > 1 ? min(nsteps_all, t_lo + a.steps_per_split) : nsteps_all;  // this workgroup: steps [t_lo, nsteps)
  const uint4* wf_tile = a.wf + (long long)ph * a.wf_phase_stride + (long long)(g * a.mblocks + mt_idx * MBT) * a.kblocks * 64;
  const int mb_last = a.mblocks - 1 - mt_idx * MBT;  // m-blocks past the group re-read the last one (never stored)

  // ---- loader: the 1 KB units of step t (weight fragments, then window pieces) round-robin over the four waves ----
  auto issue = [&](int t, int slot) -> int {
    const int q0 = t * kbs;
    const int nq = min(kbs, a.kblocks - q0);
    uint4* sa = smem + slot * stage;
    uint4* sx = sa + a_units;
    int issued = 0;
    int u = wave;
    if (!ADIR) {
#pragma unroll
      for (int mbi = 0; mbi < MBT; ++mbi) {
        const uint4* src = wf_tile + ((long long)min(mbi, mb_last) * a.kblocks + q0) * 64 + lane;
        uint4* dst = sa + mbi * kbs * 64;
        for (; u < nq; u += NW) {
          pk_lds_direct(src + u * 64, dst + u * 64);
          ++issued;
        }
        u -= nq;
      }
    }
    const int o_lo = (2 * q0) / k;
    const int o_hi = min(a.octs - 1, (2 * (q0 + nq) - 1) / k);
    const int nunits = (o_hi - o_lo + 1) * pieces;
    for (; u < nunits; u += NW) {
      const int r = u / pieces, pi = u - r * pieces;
      const int pp = pi * 64 + lane;
      pk_lds_direct(xwin + (long long)(o_lo + r) * plane + (pp ^ ((pp >> 4) & swz)), sx + r * xrow + pi * 64);
      ++issued;
    }
    return issued;
  };

  if constexpr (ADIR) {
    // ---- weights in registers: steps alternate between two fragment sets and two LDS slots (window rows only) ----
    constexpr int KBS = PK_ADIR_KBS;
    typedef __attribute__((address_space(4))) const int cint_t;
    bf16x8 areg[2][KBS][MT];
    const uint4* arow[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) arow[mt] = wf_tile + (long long)min(wm * MT + mt, mb_last) * a.kblocks * 64 + lane;
    const int q_last = a.kblocks - 1;
    auto load_a = [&](auto par_c, int t) {  // the fragments of step t into set PAR (K blocks past the end re-read the last one)
      constexpr int PAR = decltype(par_c)::value;
      const int q0 = t * KBS;
#pragma unroll
      for (int qi = 0; qi < KBS; ++qi) {
        const int q = min(q0 + qi, q_last);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) areg[PAR][qi][mt] = __builtin_bit_cast(bf16x8, arow[mt][(long long)q * 64]);
      }
    };
    auto step = [&](auto par_c, int t) {
      constexpr int PAR = decltype(par_c)::value;
      const int q0 = t * KBS;
      const int nq = min(KBS, a.kblocks - q0);
      cint_t* tbw = (cint_t*)(a.tab + q0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // step t has landed: window rows in slot PAR, fragments in set PAR
      lds_barrier();
      if (t + 1 < nsteps) {
        issue(t + 1, PAR ^ 1);
        if constexpr (PAR == 0) load_a(std::integral_constant<int, 1>{}, t + 1);
        else load_a(std::integral_constant<int, 0>{}, t + 1);
      }
      const uint4* sm = smem + PAR * stage;
      auto kloop = [&](auto swz_c) {
        constexpr bool SWZ = decltype(swz_c)::value;
        bf16x8 fb[3][NT];  // blocks qi, qi + 1, qi + 2 in flight
        auto load_b = [&](int set, int qi) {
          qi = min(qi, nq - 1);
          const int lo = kh ? tbw[2 * qi + 1] : tbw[2 * qi];
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            const int p = colu[nt] + lo;
            fb[set][nt] = *reinterpret_cast<const bf16x8*>(sm + (SWZ ? (p ^ ((p >> 4) & swz)) : p));
          }
        };
        load_b(0, 0);
        load_b(1, 1);
#pragma unroll
        for (int qi = 0; qi < KBS; ++qi) {
          if (qi + 2 < KBS) load_b((qi + 2) % 3, qi + 2);
          if (qi < nq) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
              for (int nt = 0; nt < NT; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(areg[PAR][qi][mt], fb[qi % 3][nt], acc[mt][nt], 0, 0, 0);
          }
        }
      };
      if (swz) kloop(std::true_type{});
      else kloop(std::false_type{});
    };
    if (nsteps > t_lo) {
      issue(t_lo, 0);
      load_a(std::integral_constant<int, 0>{}, t_lo);
    }
    for (int t = t_lo; t < nsteps; t += 2) {
      step(std::integral_constant<int, 0>{}, t);
      if (t + 1 < nsteps) step(std::integral_constant<int, 1>{}, t + 1);
    }
  } else {
  int n_next = 0;
  const int nst = a.nst;
  if (nsteps > t_lo) issue(t_lo, 0);
  if (nst == 3 && nsteps > t_lo + 1) n_next = issue(t_lo + 1, 1);

  int slot = -1;
  for (int t = t_lo; t < nsteps; ++t) {
    slot = slot + 1 == nst ? 0 : slot + 1;
    const int slot_ahead = slot == 0 ? nst - 1 : slot - 1;
    const int q0 = t * kbs;
    const int nq = min(kbs, a.kblocks - q0);
    // wave-uniform table reads through the constant address space: scalar loads (s_load), which leave vmcnt -- the
    // counter the LDS-direct ring is tracked with -- alone.  The entries of the step's first four K blocks are fetched
    // BEFORE the wait for its operands (their latency hides behind it).
    typedef __attribute__((address_space(4))) const int cint_t;
    cint_t* tbw = (cint_t*)(a.tab + q0);
    auto tb_at = [&](int qi) { return make_int2(tbw[2 * qi], tbw[2 * qi + 1]); };
    // step t has landed (this wave's part); with three slots step t+1 may still be in flight (two slots -- what the planner picks --
    // wait for everything: the runtime-count form is a chain of ~8 branches per step)
    if (nst == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else wait_vmcnt_le(n_next);
    lds_barrier();          // ... everyone's part; the slot about to be refilled was last read in step t-1
    {
      const int issued = t + nst - 1 < nsteps ? issue(t + nst - 1, slot_ahead) : 0;
      n_next = nst == 3 ? issued : 0;
    }
    const uint4* sm = smem + slot * stage;
    // The K loop, four K blocks per trip on FOUR operand register sets used in rotation: the LDS reads of blocks q + 2, q + 3 are
    // in flight before the MFMAs of q, q + 1 issue, and no set is ever copied (the two-set form with a "next" pair moved 16 register
    // pairs per trip: with the address arithmetic, ~45 vector instructions beside 8 MFMAs).  The swizzle of the strided layers'
    // windows is compiled only into the loop that needs it.  Reads past the step's last block re-read that block (in bounds, unused).
    auto kloop = [&](auto swz_c) {
      constexpr bool SWZ = decltype(swz_c)::value;
      // four sets (reads two K blocks ahead) where the accumulators leave room, two (one block ahead) for the 64 x 128 wave tiles:
      // <128, 256> on four waves holds 128 accumulator registers, and four sets of its six fragments spilled 520 registers
      constexpr int NSETS = (MT * NT * 16 + 16 * (MT + NT) > 176) ? 2 : 4, D = NSETS / 2;
      bf16x8 fa[NSETS][MT], fb[NSETS][NT];
      auto load = [&](int set, int qi) {
        qi = min(qi, nq - 1);
        const int2 e = tb_at(qi);
        const int lo = kh ? e.y : e.x;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[set][mt] = *reinterpret_cast<const bf16x8*>(sm + abase[mt] + qi * 64);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const int p = colu[nt] + lo;
          fb[set][nt] = *reinterpret_cast<const bf16x8*>(sm + a_units + (SWZ ? (p ^ ((p >> 4) & swz)) : p));
        }
      };
      auto mma = [&](int set) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[set][mt], fb[set][nt], acc[mt][nt], 0, 0, 0);
      };
#pragma unroll
      for (int u = 0; u < D; ++u) load(u, u);
      for (int qi = 0; qi < nq; qi += NSETS) {
#pragma unroll
        for (int u = 0; u < D; ++u) load(D + u, qi + D + u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < D; ++u)
          if (u == 0 || qi + u < nq) mma(u);
        __builtin_amdgcn_sched_barrier(0);
        if (qi + D >= nq) break;
#pragma unroll
        for (int u = 0; u < D; ++u) load(u, qi + NSETS + u);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < D; ++u)
          if (u == 0 || qi + D + u < nq) mma(D + u);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (swz) kloop(std::true_type{});
    else kloop(std::false_type{});
  }
  }  // (!ADIR)

  // ---- epilogue: D layout: lane column = output position, registers = output channels ----
  if (a.ksplit > 1) {  // raw partial tile: rows = output channels, columns = the flat (item, position) index (coalesced)
    float* pp = a.part + ((long long)split * a.phases + ph) * a.part_stride;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (col_b[nt] < 0) continue;
      const long long n = n0 + (wn * NT + nt) * 32 + ln;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (m < m_valid) pp[(long long)(co0 + m) * a.part_ld + n] = acc[mt][nt][r];
        }
    }
    return;
  }
  if (a.po.y) {  // ---- flat packed output: bf16 units [octet][unit], straight from the accumulators ----
    if (TAILS && a.po.zero_n > 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
      for (int i = tid; i < a.po.zero_n; i += NW * 64) a.po.zero_p[i] = make_uint4(0u, 0u, 0u, 0u);
    pk_with_act(a.act, [&](auto act_c) {
      constexpr int ACT = decltype(act_c)::value;
      const int m_last = m_valid - 1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int n = (int)n0 + (wn * NT + nt) * 32 + ln;
        const int w = n * a.out_stride + out_off;  // (flat tensors stay far below 2^31 units)
        const int bb = w / a.po.Tc;
        const int u = w - bb * a.po.Tc;
        const bool ok = col_b[nt] >= 0 && w >= 0 && u < a.po.valid;
        const bool pad_col = TAILS && a.po.tail && !ok && n >= a.po.valid && n < a.po.pad_end;  // (tails: one row, w == n)
        const long long dst_u = ok ? (long long)bb * a.po.Ts + u : (pad_col ? (long long)n : 0);
        const long long msk_u = ok ? (long long)bb * a.po.Tm + u : 0;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const int mb = (wm * MT + mt) * 32;
          float v[16];
          {
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = 0.f;
            if (a.bias) {
#pragma unroll
              for (int r = 0; r < 16; ++r) bv[r] = a.bias[co0 + min(mb + (r & 3) + 8 * (r >> 2) + 4 * kh, m_last)];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = pk_act<ACT>(acc[mt][nt][r] + bv[r], a.act_param);
          }
          if (a.po.mask) {  // lane (n, kh) holds channels 8 i + 4 kh .. + 3 of octet i: the kh-th 8 bytes of that octet's unit
            uint2 mk[4], fr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const long long row = (co0 + min(mb + 8 * i, m_last & ~7)) >> 3;
              const uint2* src = reinterpret_cast<const uint2*>(a.po.mask + row * a.po.mplane + msk_u) + kh;
              mk[i] = *src;
              fr[i] = mk[i];
              if (a.po.fm) fr[i] = *(reinterpret_cast<const uint2*>(a.po.fm + row * a.po.mplane + msk_u) + kh);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) pk_flat_tail4(v + 4 * i, mk[i], fr[i], a.po.fm_scale, a.po.mask_slope);
          }
          float s2[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) s2[r] = 0.f;
          if (TAILS && a.po.tail) {  // (po.mask is not set with a tail: evmi_conv1d_cbt_bf16pk_ffn_up / ..._ffn_down)
            uint2 pre[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) pre[i] = make_uint2(0u, 0u);
            if (a.po.tail == 2) {
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                const long long row = (co0 + min(mb + 8 * i, m_last & ~7)) >> 3;
                pre[i] = *(reinterpret_cast<const uint2*>(a.po.fm + row * a.po.mplane + msk_u) + kh);
              }
            }
            const unsigned long long dseed = a.drop_seed.get(), ld = (unsigned long long)a.po.drop_ld;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int c = co0 + min(mb + 8 * i + 4 * kh, m_last & ~3);
              // (the column itself, not dst_u: a lane past the last column still hashes for its pair -- with tight items they are equal)
              // (an odd row length -- a single item of any length -- puts a pair's elements in lanes n, n +- 1 depending on the channel:
              // one hash per element there)
#define EVMI_PK_TAIL(T_, L_)                                                                                                                  \
  pk_flat_tail_silu<T_, L_>(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3], s2[4 * i], s2[4 * i + 1], s2[4 * i + 2], s2[4 * i + 3], pre[i], dseed, \
                            a.drop_p, (unsigned long long)c, (unsigned long long)n, ld)
              if (ld & 1ull) {
                if (a.po.tail != 2) EVMI_PK_TAIL(1, false);
                else EVMI_PK_TAIL(2, false);
              } else {
                if (a.po.tail != 2) EVMI_PK_TAIL(1, true);
                else EVMI_PK_TAIL(2, true);
              }
#undef EVMI_PK_TAIL
            }
          }
#pragma unroll
          for (int which = 0; which < 2; ++which) {
            if (which == 1 && !(TAILS && a.po.tail == 1)) break;
            const bool act_out = which || (TAILS && a.po.tail == 3);  // (tail 3: the activated tensor is the only output)
            uint4* dst = which ? a.po.y2 : a.po.y;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              u32x4 d;
#pragma unroll
              for (int q = 0; q < 4; ++q)
                d[q] = act_out ? pack_bf16x2(s2[8 * p + 2 * q], s2[8 * p + 2 * q + 1]) : pack_bf16x2(v[8 * p + 2 * q], v[8 * p + 2 * q + 1]);
              const u32x4 o = swap_quads_bf16(d);  // lane (n, kh): the 8 channels of octet 2 p + kh
              const int m_oct = mb + 8 * (2 * p + kh);
              if ((ok || pad_col) && m_oct < m_valid) {
                uint4 st;
                st.x = o[0]; st.y = o[1]; st.z = o[2]; st.w = o[3];
                if (pad_col) st = make_uint4(0u, 0u, 0u, 0u);
                dst[(long long)((co0 + m_oct) >> 3) * a.po.plane + dst_u] = st;
              }
            }
          }
        }
      }
    });
    return;
  }
  // Every global read of the tail (bias, mask, residual, previous value) is requested for a whole 16-register block before the
  // first is used, from clamped (always valid) addresses: a read under a per-element condition is compiled as a branch with a
  // full vmcnt(0) wait behind it -- one memory round trip per element, 64-256 in a row per lane.  Absent operands are replaced by
  // the neutral element (x * 1, x + -0) so the arithmetic is unconditional, and the products / sums are pinned to separate
  // roundings (no contraction): conv_pk_reduce_kernel computes the same tail for the split-K launches, bit for bit.
  pk_with_act(a.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
    const long long ch_stride = (long long)a.B * a.t_out_total;
    const int m_last = m_valid - 1;
    constexpr int EB = 8;  // registers per batch: 8 loads of each operand in flight, ~40 temporaries
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (col_b[nt] < 0) continue;
      const long long col_off = (long long)col_b[nt] * a.t_out_total + (long long)col_to[nt] * a.out_stride + out_off;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r0 = 0; r0 < 16; r0 += EB) {
          int mrow[EB];
          long long off[EB];
          float bv[EB], fac[EB], add[EB], old[EB];
#pragma unroll
          for (int e = 0; e < EB; ++e) {
            const int r = r0 + e;
            mrow[e] = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            off[e] = col_off + (long long)(co0 + min(mrow[e], m_last)) * ch_stride;
            bv[e] = 0.f;
            fac[e] = 1.f;
            add[e] = -0.f;
            old[e] = -0.f;
          }
          if (a.bias) {
#pragma unroll
            for (int e = 0; e < EB; ++e) bv[e] = a.bias[co0 + min(mrow[e], m_last)];
          }
          if (a.out_mask) {
#pragma unroll
            for (int e = 0; e < EB; ++e) fac[e] = a.out_mask[off[e]];
#pragma unroll
            for (int e = 0; e < EB; ++e) fac[e] = fac[e] > 0.f ? 1.f : a.out_mask_slope;
          }
          if (a.res) {
#pragma unroll
            for (int e = 0; e < EB; ++e) add[e] = a.res[off[e]];
          }
          if (a.drop_p > 0.f) {
            const unsigned long long dseed = a.drop_seed.get();
#pragma unroll
            for (int e = 0; e < EB; ++e) fac[e] *= uniform01(dseed, (unsigned long long)off[e]) >= a.drop_p ? a.drop_fac : 0.f;
          }
          if (a.accumulate) {
#pragma unroll
            for (int e = 0; e < EB; ++e) old[e] = a.y[off[e]];
          }
#pragma unroll
          for (int e = 0; e < EB; ++e) {
            float v = pk_act<ACT>(acc[mt][nt][r0 + e] + bv[e], a.act_param);
            v = pk_tail(v, fac[e], add[e], old[e]);
            if (mrow[e] < m_valid) a.y[off[e]] = v;
          }
        }
      }
    }
  });
}

// y[co][b][to*os + oo] (+)= act(bias + sum over the splits' partial tiles, in split order): one thread per (co, n), n fastest;
// grid.z = phase
__global__ __launch_bounds__(256) void conv_pk_reduce_kernel(ConvPkArgs a) {
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  const int co = blockIdx.y, ph = blockIdx.z;
  const int n_out = a.ph_nout[ph];
  if (n_out <= 0 || n >= (long long)a.B * n_out) return;
  float v = ordered_sum_strided(a.part + (long long)ph * a.part_stride + (long long)co * a.part_ld + n, (long long)a.phases * a.part_stride, a.ksplit);
  if (a.bias) v += a.bias[co];
  pk_with_act(a.act, [&](auto act_c) { v = pk_act<decltype(act_c)::value>(v, a.act_param); });
  const long long bb = n / n_out;
  const int to = (int)(n - bb * n_out);
  float* dst = a.y + ((long long)co * a.B + bb) * a.t_out_total + (long long)to * a.out_stride + a.ph_off[ph];
  float fac = a.out_mask ? (a.out_mask[dst - a.y] > 0.f ? 1.f : a.out_mask_slope) : 1.f;
  if (a.drop_p > 0.f) fac *= uniform01(a.drop_seed.get(), (unsigned long long)(dst - a.y)) >= a.drop_p ? a.drop_fac : 0.f;
  const float add = a.res ? a.res[dst - a.y] : -0.f;
  const float old = a.accumulate ? *dst : -0.f;
  *dst = pk_tail(v, fac, add, old);
}

// The same for a flat packed output: one thread per (channel octet, FOUR consecutive columns) -- 16-byte reads of the partial rows
// (part_ld is a multiple of 4), one 16-byte unit stored per column; grid (ceil(n / 1024), c_out / 8, phases)
__global__ __launch_bounds__(256) void conv_pk_reduce_flat_kernel(ConvPkArgs a) {
  const int n4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  const int oc = blockIdx.y, ph = blockIdx.z;
  const int n_out = a.ph_nout[ph];
  const int n_cols = a.B * n_out;  // (the chains' calls: B == 1; the FastSpeech2 feed-forward layers: B tight items, one row)
  if (a.po.tail && a.po.zero_n > 0 && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0)
    for (int i = threadIdx.x; i < a.po.zero_n; i += 256) a.po.zero_p[i] = make_uint4(0u, 0u, 0u, 0u);
  if (a.po.tail && n4 < a.po.pad_end && n4 + 3 >= n_cols && n_out > 0) {  // the zero units behind the row's last column
    for (int c = 0; c < 4; ++c) {
      const int n = n4 + c;
      if (n >= max(n_cols, a.po.valid) && n < a.po.pad_end) {
        a.po.y[(long long)oc * a.po.plane + n] = make_uint4(0u, 0u, 0u, 0u);
        if (a.po.tail == 1) a.po.y2[(long long)oc * a.po.plane + n] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  }
  if (n_out <= 0 || n4 >= n_cols) return;
  float v[4][8];
  const float* src = a.part + (long long)ph * a.part_stride + (long long)(oc * 8) * a.part_ld + n4;
  const long long sstride = (long long)a.phases * a.part_stride;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int sp = 0; sp < a.ksplit; ++sp) {  // split order: fixed summation order
      const float4 t = *reinterpret_cast<const float4*>(src + (long long)e * a.part_ld + sp * sstride);
      acc.x += t.x; acc.y += t.y; acc.z += t.z; acc.w += t.w;
    }
    v[0][e] = acc.x; v[1][e] = acc.y; v[2][e] = acc.z; v[3][e] = acc.w;
  }
  float bv[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bv[e] = a.bias ? a.bias[oc * 8 + e] : 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int n = n4 + c;
    if (n >= n_cols) break;
    const int w = n * a.out_stride + a.ph_off[ph];
    const int bb = w / a.po.Tc, u = w - bb * a.po.Tc;
    if (w < 0 || u >= a.po.valid) continue;
    float* vc = v[c];
    if (a.bias) {
#pragma unroll
      for (int e = 0; e < 8; ++e) vc[e] += bv[e];
    }
    pk_with_act(a.act, [&](auto act_c) {
#pragma unroll
      for (int e = 0; e < 8; ++e) vc[e] = pk_act<decltype(act_c)::value>(vc[e], a.act_param);
    });
    if (a.po.mask) {
      const long long mu = (long long)oc * a.po.mplane + (long long)bb * a.po.Tm + u;
      const uint4 mk = a.po.mask[mu];
      const uint4 fr = a.po.fm ? a.po.fm[mu] : mk;
      pk_flat_tail4(vc, make_uint2(mk.x, mk.y), make_uint2(fr.x, fr.y), a.po.fm_scale, a.po.mask_slope);
      pk_flat_tail4(vc + 4, make_uint2(mk.z, mk.w), make_uint2(fr.z, fr.w), a.po.fm_scale, a.po.mask_slope);
    }
    const long long du = (long long)bb * a.po.Ts + u;
    float s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (a.po.tail) {
      uint4 pre = make_uint4(0u, 0u, 0u, 0u);
      if (a.po.tail == 2) pre = a.po.fm[(long long)oc * a.po.mplane + (long long)bb * a.po.Tm + u];
      const unsigned long long dseed = a.drop_seed.get(), ld = (unsigned long long)a.po.drop_ld;
      const unsigned long long c0 = (unsigned long long)(oc * 8), nn = (unsigned long long)du;
      if (a.po.tail != 2) {
        pk_flat_tail_silu<1, false>(vc[0], vc[1], vc[2], vc[3], s2[0], s2[1], s2[2], s2[3], make_uint2(pre.x, pre.y), dseed, a.drop_p, c0, nn, ld);
        pk_flat_tail_silu<1, false>(vc[4], vc[5], vc[6], vc[7], s2[4], s2[5], s2[6], s2[7], make_uint2(pre.z, pre.w), dseed, a.drop_p, c0 + 4ull, nn, ld);
        if (a.po.tail == 3) {
#pragma unroll
          for (int e = 0; e < 8; ++e) vc[e] = s2[e];
        }
      } else {
        pk_flat_tail_silu<2, false>(vc[0], vc[1], vc[2], vc[3], s2[0], s2[1], s2[2], s2[3], make_uint2(pre.x, pre.y), dseed, a.drop_p, c0, nn, ld);
        pk_flat_tail_silu<2, false>(vc[4], vc[5], vc[6], vc[7], s2[4], s2[5], s2[6], s2[7], make_uint2(pre.z, pre.w), dseed, a.drop_p, c0 + 4ull, nn, ld);
      }
    }
    uint4 st;
    st.x = pack_bf16x2(vc[0], vc[1]); st.y = pack_bf16x2(vc[2], vc[3]); st.z = pack_bf16x2(vc[4], vc[5]); st.w = pack_bf16x2(vc[6], vc[7]);
    a.po.y[(long long)oc * a.po.plane + du] = st;
    if (a.po.tail == 1) {
      st.x = pack_bf16x2(s2[0], s2[1]); st.y = pack_bf16x2(s2[2], s2[3]); st.z = pack_bf16x2(s2[4], s2[5]); st.w = pack_bf16x2(s2[6], s2[7]);
      a.po.y2[(long long)oc * a.po.plane + du] = st;
    }
  }
}

// LayerNorm over the channels of every column, written STRAIGHT into the packed layout of a pointwise layer's input (tight items:
// [octet][B * T units]): workgroup = 64 columns x 4 channel slices as fs2_ops.hip's layernorm_cbt_kernel (same arithmetic, same
// order), but a thread's slice of C / 4 channels is whole octets, so it leaves as 16-byte units -- consecutive lanes, consecutive
// units.  The normalised tensor never exists in fp32 (its only reader is the dense layer behind it).
template <int CPT>  // channels per thread = C / 4 (a multiple of 8)
__global__ __launch_bounds__(256) void layernorm_pack_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, uint4* __restrict__ xp, int C, long long N,
                                                            long long plane, float eps, int slack_units, unsigned n_ln, WfragArgs f1,
                                                            WfragArgs f2) {
  // (plane >= N: units per octet row; the units behind column N - 1 of every row are zero -- n_ln = ceil(plane / 64) workgroups)
  // The workgroups behind those re-lay weights (prep_pk_kernel's second half): the fragments of the layer this LayerNorm feeds and,
  // for a feed-forward block, of the layer behind that one -- the convolutions then start without a preparation launch of their own.
  if (blockIdx.x >= n_ln) {
    unsigned b = blockIdx.x - n_ln;
    const unsigned n1 = (unsigned)f1.gx * f1.gy * f1.gz;
    const WfragArgs& f = b < n1 ? f1 : f2;
    if (b >= n1) b -= n1;
    const unsigned q = b % f.gx, r = b / f.gx;
    wfrag_pk_block(f, (int)q, (int)(r % f.gy), (int)(r / f.gy));
    return;
  }
  __shared__ float red[2][4][64];
  const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const long long n = (long long)blockIdx.x * 64 + lane;
  const bool live = n < N;
  if (blockIdx.x == 0) {  // the zero slack behind the packed tensor (what pack_x_block's last block writes)
    uint4* tail = xp + (long long)(C / 8) * plane;
    for (int i = threadIdx.x; i < slack_units; i += 256) tail[i] = make_uint4(0u, 0u, 0u, 0u);
  }
  const int c0 = slice * CPT;
  float v[CPT];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    v[i] = live ? x[(long long)(c0 + i) * N + n] : 0.f;
    s += v[i];
  }
  red[0][slice][lane] = s;
  __syncthreads();
  const float mean = (red[0][0][lane] + red[0][1][lane] + red[0][2][lane] + red[0][3][lane]) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    const float d = v[i] - mean;
    q = fmaf(d, d, q);
  }
  red[1][slice][lane] = q;
  __syncthreads();
  const float var = (red[1][0][lane] + red[1][1][lane] + red[1][2][lane] + red[1][3][lane]) / (float)C;
  const float rstd = 1.f / sqrtf(var + eps);
  if (!live) {
    if (n < plane) {
#pragma unroll
      for (int o = 0; o < CPT / 8; ++o) xp[(long long)(slice * CPT / 8 + o) * plane + n] = make_uint4(0u, 0u, 0u, 0u);
    }
    return;
  }
#pragma unroll
  for (int o = 0; o < CPT / 8; ++o) {
    float r[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + 8 * o + e;
      r[e] = (v[8 * o + e] - mean) * rstd * gamma[c] + beta[c];
    }
    uint4 out;
    out.x = pk_bf16x2(r[0], r[1]);
    out.y = pk_bf16x2(r[2], r[3]);
    out.z = pk_bf16x2(r[4], r[5]);
    out.w = pk_bf16x2(r[6], r[7]);
    xp[(long long)(c0 / 8 + o) * plane + n] = out;
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------
struct PkTile { int bm, bn; };
// (index 7, 8: eight-wave forms of 128 x 256 and 128 x 128 -- 64 x 64 / 64 x 32 per wave)
// (index 9: 128 x 128 with the weight fragments in registers -- conv_pk_kernel<..., ADIR>.  The same for the 32 x 128 tile of the
// narrow-group layers was built and measured SLOWER -- 110 -> 130 us, 66 -> 87 us on the scale discriminators' first grouped layers:
// its four waves all need the same 32 rows, so registers mean four fetches of every fragment where the LDS needs one)
static const PkTile kPkTiles[] = {{128, 128}, {64, 128}, {64, 64}, {32, 128}, {64, 256}, {32, 256}, {128, 256}, {128, 256}, {128, 128}, {128, 128}, {32, 512}};
constexpr int kNumPkTiles = sizeof(kPkTiles) / sizeof(kPkTiles[0]);

static int pk_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

struct PkPlan {
  int ti;
  size_t lds;
  dim3 grid;
  int PL;                  // left padding of the packed items (units)
  long long xp_units;      // packed input incl. slack
  long long wf_units;      // fragments, all phases
  int cin_g, t_in, groups;
  long long part_elems;    // split-K partial tiles (floats)
  int c_out;
};

static long long round_up_ll(long long v, long long m) { return (v + m - 1) / m * m; }

// Fills the geometry of `a` (phases must be set: ph_nout, ph_off and the per-phase padding in ph_shift) and the plan.
static const char* plan_pk(ConvPkArgs& a, int cin_g, int t_in, int groups, const int* ph_pad, PkPlan& pl) {
  if (cin_g <= 0 || a.cout_g <= 0 || a.k <= 0 || a.stride <= 0 || a.dil <= 0 || a.B <= 0 || t_in <= 0) return "bad shape";
  if (cin_g < 8) return "fewer than 8 channels per group";
  if (a.cout_g <= 4) return "direct-kernel shape";
  a.octs = (cin_g + 7) / 8;
  a.kblocks = (a.octs * a.k + 1) / 2;
  a.mblocks = (a.cout_g + 31) / 32;
  int PL = 0, n_max = 0, n_min = 0;
  for (int p = 0; p < a.phases; ++p) PL = std::max(PL, ph_pad[p]);
  long long ext = (long long)PL + t_in;
  for (int p = 0; p < a.phases; ++p) {
    a.ph_shift[p] = PL - ph_pad[p];
    if (a.ph_nout[p] <= 0) continue;
    ext = std::max<long long>(ext, (long long)a.ph_shift[p] + (long long)(a.ph_nout[p] - 1) * a.stride + (long long)(a.k - 1) * a.dil + 1);
    n_max = std::max(n_max, a.ph_nout[p]);
    n_min = n_min == 0 ? a.ph_nout[p] : std::min(n_min, a.ph_nout[p]);
  }
  if (n_max <= 0) return "no outputs";
  if (ext > (1 << 24)) return "row too long";
  // Items are Tp units apart in the packed rows.  A column tile that spans several (short) items reads, per 16-lane group, units
  // of more than one item: with Tp = n_out * stride (mod 16) the unit index keeps advancing by `stride` across the item boundary
  // modulo the 16 slots of a bank row, as inside one long row -- no two lanes of a group on one slot.  Costs at most 15 padding
  // units per item, so only items of 96 units or more are padded (with every item padded the 128 x 128 kernel's conflicts fell
  // from 31.8 % to 5.2 % of its LDS cycles, but the GAN step got 0.3 ms slower: the period discriminators' 11-34-unit items grew
  // by up to half).
  {
    static const int align_tp = 1;
    const long long want_mod = ((long long)n_max * a.stride) & 15;
    if (align_tp && ext >= 96) ext += ((want_mod - (ext & 15)) + 16) & 15;  // (short items: the padding would cost more than the conflicts)
  }
  // pointwise stride-1 layers are packed tight (ext == t_in here): the layout the weight gradient reads too (conv_pk_common.h); a
  // single item's row is rounded up to the weight gradient's K step (nothing follows it: the pitch is free)
  const bool pointwise = a.phases == 1 && PL == 0 && pk_shared_shape(a.k, a.stride, 0, a.dil, groups) && ext == t_in;
  if (pointwise && a.B == 1) ext = pk_shared_pitch(1, t_in);
  const bool shared = pointwise && pk_shared_items(a.B, t_in);
  a.Tp = (int)ext;
  const long long n_total = (long long)a.B * n_max;
  auto blocks = [&](int i) {
    return ((n_total + kPkTiles[i].bn - 1) / kPkTiles[i].bn) * ((a.cout_g + kPkTiles[i].bm - 1) / kPkTiles[i].bm) * groups;
  };
  static const long long want = 512;  // two workgroups per CU: one's epilogue / load waits overlap the other's MFMAs
  int ti;
  // wide layers on few columns (the 1024-channel discriminator layers: 1.6-2.8 k columns): 128 x 128 tiles -- one LDS read per
  // MFMA instead of the 64 x 64 tile's two -- fill the CUs only with the contraction split over workgroups
  static const int allow_split = pk_env_int("EVMI_PK_SPLITK", 1);
  a.ksplit = 1;
  // (flat one-item calls -- B == 1 with thousands of columns -- also split between 256 and 384 tiles: 4.2-4.7 k columns of a
  // 1024-channel layer otherwise fall to 64 x 128 tiles, 136 vs 92 us, tools/pkflat_bench.py)
  if (allow_split && a.cout_g > 64 && blocks(0) < (a.B == 1 ? 384 : 256) && a.kblocks >= 64) {
    static const int split_min_kb = 24;
    int ks;
    if (a.B == 1) {
      // flat calls: the split that needs the fewest rounds of the chip's 512 workgroup slots per unit of contraction, a reduce pass
      // priced at ~4 % of a round per split (4.2 k columns x 1024 channels: 264 tiles -- two splits are 528 workgroups, one past a
      // round, three are 792: two rounds of a third of the contraction each)
      double best = 1e30;
      ks = 1;
      for (int c = 2; c <= 8; ++c) {
        if (a.kblocks / c < split_min_kb) break;
        const double cost = (double)((blocks(0) * c + 511) / 512) / c + 0.04 * c;
        if (cost < best) { best = cost; ks = c; }
      }
    } else {
      static const int split_want = pk_env_int("EVMI_PK_SPLIT_WANT", 384);
      ks = (int)std::min<long long>(8, (split_want + blocks(0) - 1) / blocks(0));
      while (ks > 1 && a.kblocks / ks < split_min_kb) --ks;
    }
    a.ksplit = ks;
  }
  // candidate tiles in order of preference; the next one is tried while the staged window does not fit
  int cand[12], nc = 0;
  // 256-column tiles for narrow layers on many columns (the generator's last stages: 65-131 k columns, 32-64 channels; half the
  // prologues / epilogues per column): measured 26.6 vs 25.3 ms per GAN step (EVMI_PK_WIDE=1 vs 0) -- kept as a switch, off
  static const int wide = pk_env_int("EVMI_PK_WIDE", 0);
  // weights in registers for the wide tile wherever it would be picked (EVMI_PK_ADIR=0: the LDS form, A/B)
  static const int use_adir = pk_env_int("EVMI_PK_ADIR", 1);
  if (a.ksplit > 1) { if (use_adir) cand[nc++] = 9; cand[nc++] = 0; cand[nc++] = 1; cand[nc++] = 2; cand[nc++] = 3; }
  else if (a.cout_g > 64) {
    // Measured at the FastSpeech2 decoder's shapes (32 x 814 columns; tools/debug/pk_tile_bench.py, pack + convolution):
    //  * 128 x 256 tiles for long contractions on many columns (the postnet's 512 -> 512, k = 5: 150 vs 180 us): the staged window
    //    and the weight fragments are each read by half as many workgroups;
    //  * 128 x 128 already from 1.5 workgroups per CU (256-row layers: 1024 -> 256 81 vs 97 us, 256 -> 256 34 vs 38 us on 64 x 128).
    static const long long want0 = 384, want6 = 384;
    static const int kb6 = 128;
    if (a.cout_g >= 128 && a.kblocks >= kb6 && blocks(6) >= want6) cand[nc++] = 6;
    //  * short contractions on many columns (the FastSpeech2 decoder's 256-channel dense layers: 16 K blocks, 26 k columns): the
    //    eight-wave 128 x 128 tile -- 71 vs 82 us (256 -> 1024), 56 vs 62 us (256 -> 768), pack included (tools/pk_tile_sweep.py);
    //    slower everywhere else (longer contractions, short items, grouped layers), as is the eight-wave 128 x 256 tile
    if (a.kblocks <= 16 && a.k == 1 && blocks(8) >= 2 * want0) cand[nc++] = 8;
    if (blocks(0) >= want0) { if (use_adir) cand[nc++] = 9; cand[nc++] = 0; }
    if (blocks(1) >= want || nc == 0) cand[nc++] = blocks(1) >= want ? 1 : 2;
    cand[nc++] = 2; cand[nc++] = 3;
  } else if (a.cout_g > 32) {
    if (wide && blocks(4) >= want) cand[nc++] = 4;
    cand[nc++] = blocks(1) >= want ? 1 : 2; cand[nc++] = 2; cand[nc++] = 3;
  } else {
    // flat one-item calls of narrow groups (the scale discriminators' grouped 41-tap layers: 8-32 rows, 65-131 k columns): the weight
    // fragments are four fifths of a step's loads there and every column tile re-loads them -- 256-column tiles halve that:
    // forward 110 -> 87 / input gradient 143 -> 106 us (128 -> 128, g 4), 180 -> 129 us (128 -> 256, g 16); stride-4 windows get too
    // long for it (40 -> 55 us), those keep 128 columns (tools/pkflat_bench.py, EVMI_PK_WIDE=1 against 0)
    // ... and 512-column tiles (index 10) the 8 / 16-row groups: 128 -> 256, g 16 forward 62 -> 52, input gradient 129 -> 111 us, the
    // input gradient of 256 -> 512, g 16 87 -> 81 us; 32-row groups lose on them (86 -> 124 us)
    if (a.B == 1 && a.stride <= 2 && a.cout_g <= 16 && blocks(10) >= want) cand[nc++] = 10;
    if ((wide || (a.B == 1 && a.stride <= 2)) && blocks(5) >= want) cand[nc++] = 5;
    cand[nc++] = 3;
  }
  const int forced = pk_env_int("EVMI_PK_TILE", -1);
  if (forced >= 0 && forced < kNumPkTiles) { cand[0] = forced; nc = 1; }
  ti = cand[0];
  const size_t two_wg = 78 * 1024, one_wg = 160 * 1024;
  const int ti_first = ti;
  for (int ci = 0;; ++ci) {  // the next candidate while the staged window does not fit
    if (ci >= nc) return "LDS budget";
    ti = cand[ci];
    const bool last = ci == nc - 1;
    const int bm = kPkTiles[ti].bm, bn = kPkTiles[ti].bn;
    const int items_max = (int)std::min<long long>(a.B, (bn + n_min - 2) / n_min + 1);
    // tile column c of item bb sits at unit c*s + (bb - b_first) * (Tp - n_out*s) of the window (+ tap * dilation)
    const long long gap = std::max<long long>(0, (long long)a.Tp - (long long)n_min * a.stride);
    const long long win = (long long)(bn - 1) * a.stride + (items_max - 1) * gap + (long long)(a.k - 1) * a.dil + 1;
    if (win > 64 * 24) { if (last) return "input window too long"; continue; }
    a.pieces = (int)((win + 63) / 64);
    a.xrow = a.pieces * 64;
    const bool adir = ti == 9;
    auto rows_of = [&](int kbs) { return std::min(a.octs, (2 * kbs + a.k - 2) / a.k + 1); };
    auto lds_of = [&](int kbs, int nst) { return (size_t)nst * ((adir ? 0 : (bm / 32) * kbs * 64) + rows_of(kbs) * a.xrow) * 16; };
    if (lds_of(1, 2) > one_wg) { if (last) return "LDS budget"; continue; }
    // deepest step (K blocks) that leaves two workgroups per CU; three slots when they fit at that depth
    int kbs = 1, nst = 2;
    const size_t budget = lds_of(1, 2) <= two_wg ? two_wg : one_wg;
    const int kbs_cap = std::min(a.kblocks, 32);
    while (kbs < kbs_cap && lds_of(kbs + 1, 2) <= budget) ++kbs;
    // (three slots at 2/3 of the depth measured slower on every layer: the per-step cost -- barrier, scalar bookkeeping, the
    // un-overlapped first fragment reads -- outweighs the extra step of load latency hidden)
    if (0 && lds_of(std::max(1, kbs * 2 / 3), 3) <= budget && kbs >= 3) { nst = 3; kbs = std::max(1, kbs * 2 / 3); }
    if (adir) {  // fixed step depth (the fragment registers are indexed at compile time), two slots
      if (a.kblocks < PK_ADIR_KBS || lds_of(PK_ADIR_KBS, 2) > two_wg) { if (last) return "LDS budget"; continue; }
      kbs = PK_ADIR_KBS;
      nst = 2;
    }
    while (kbs > 1 && lds_of(kbs, nst) > one_wg) --kbs;
    if (lds_of(kbs, nst) > one_wg) return "LDS budget";
    a.kb_step = kbs;
    a.rows_step = rows_of(kbs);
    a.nst = nst;
    a.mtiles_per_group = (a.cout_g + bm - 1) / bm;
    pl.lds = lds_of(kbs, nst);
    if ((n_total + bn - 1) / bn > 0x7fffffffLL || groups * a.mtiles_per_group > 65535 || groups * a.mblocks > 65535) return "grid limits";
    a.ntiles_n = (int)((n_total + bn - 1) / bn);
    if (a.ksplit > 1 && ti != ti_first && !(ti == 0 && ti_first == 9)) a.ksplit = 1;  // (the wide tile did not fit: no split)
    if (a.ksplit > 1) {
      const int nsteps_all = (a.kblocks + kbs - 1) / kbs;
      a.ksplit = std::min(a.ksplit, nsteps_all);
      a.steps_per_split = (nsteps_all + a.ksplit - 1) / a.ksplit;
      a.ksplit = (nsteps_all + a.steps_per_split - 1) / a.steps_per_split;  // no empty splits
    }
    pl.grid = dim3((unsigned)(a.ntiles_n * a.ksplit), groups * a.mtiles_per_group, a.phases);
    {  // band height of the XCD order: the run of an XCD (1/8 of the list) as square as the tile grid allows -- fewest unique
       // weight rows + window columns per L2.  Rows of different groups share nothing: bands stay inside a group's m-tiles.
      const long long Y = a.mtiles_per_group, X = (long long)a.ntiles_n * a.ksplit, N = a.ntiles_n;
      double best = 1e300;
      a.xcd_hb = 1;
      static const int fixed_hb = pk_env_int("EVMI_PK_XCD_HB", 0);
      for (int xm = 8; xm >= 1; xm >>= 1) {
        const long long hb = (Y + xm - 1) / xm;
        if (groups > 1 && hb > 1 && (Y % hb)) continue;  // (a band must not straddle two groups)
        const double run = std::max(1.0, (double)X * Y * groups / 8.0 / hb);  // columns per run
        const double c = std::max(1.0, run / N);                              // K slices (splits) a run spans
        const double cols = std::min<double>(run, N);
        const double cost = (hb * bm + cols * bn) * c / a.ksplit;
        if (cost < best) { best = cost; a.xcd_hb = (int)hb; }
      }
      if (fixed_hb > 0) a.xcd_hb = (int)std::min<long long>(fixed_hb, Y);
    }
    break;
  }
  pl.c_out = a.cout_g * groups;
  a.part_ld = (n_total + 3) / 4 * 4;  // B * longest phase (rows 16-byte aligned: the flat reduce reads them four columns at a time)
  a.part_stride = (long long)pl.c_out * a.part_ld;
  pl.part_elems = a.ksplit > 1 ? a.part_stride * a.ksplit * a.phases : 0;
  pl.ti = ti;
  pl.PL = PL;
  pl.cin_g = cin_g; pl.t_in = t_in; pl.groups = groups;
  // slack: the last window piece of the last tile reads up to 63 units past its window, rows of the last octet included
  pl.xp_units = (long long)groups * a.octs * a.B * a.Tp + std::max<long long>((long long)a.xrow + 64, shared ? PK_SHARED_SLACK : 0);
  a.wf_phase_stride = (long long)groups * a.mblocks * a.kblocks * 64;
  pl.wf_units = a.wf_phase_stride * a.phases + ((long long)a.kblocks * 8 + 15) / 16;  // + the K-block offset table (int2 each)
  if (pl.xp_units >= (1LL << 31)) return "packed input too large";

  return nullptr;
}

struct PkInputFusion {  // what the pack applies to the input on its way in (see PackArgs)
  float pre_slope = 1.f;
  const float* mask = nullptr;
  float mask_slope = 1.f;
  int fuse = 0;  // 1: dropout(silu(x)); 2: dropout(x) * silu'(aux); 3: fuse_scale * dropout(x)
  float fuse_scale = 1.f;
  const float* aux = nullptr;
  float p_drop = 0.f;
  SeedArg seed = SeedArg{0ull, nullptr};
};

static int launch_pk_tile(ConvPkArgs& a, const PkPlan& pl, hipStream_t stream) {
  static const int xcd_remap = 1;
  a.xcd_remap = xcd_remap;
  const size_t lds = pl.lds;
  static thread_local size_t configured_dev[kMaxDevices][2 * kNumPkTiles] = {};
  size_t* configured = configured_dev[device_slot()];
#define EVMI_PK_LAUNCH_AS(BM, BN, WM, WN, ADIR, TAILS, IDX)                                                              \
  {                                                                                                                      \
    if (lds > configured[IDX]) {                                                                                         \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)conv_pk_kernel<BM, BN, WM, WN, ADIR, TAILS>,                       \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                        \
      configured[IDX] = lds;                                                                                             \
    }                                                                                                                    \
    hipLaunchKernelGGL((conv_pk_kernel<BM, BN, WM, WN, ADIR, TAILS>), pl.grid, dim3(WM * WN * 64), lds, stream, a);      \
  }
#define EVMI_PK_LAUNCH(BM, BN, WM, WN, ADIR, IDX)                                                                        \
  {                                                                                                                      \
    if (tails) EVMI_PK_LAUNCH_AS(BM, BN, WM, WN, ADIR, true, kNumPkTiles + IDX)                                          \
    else EVMI_PK_LAUNCH_AS(BM, BN, WM, WN, ADIR, false, IDX)                                                             \
  }
  const bool tails = a.po.y && a.po.tail;
  switch (pl.ti) {
    case 0: EVMI_PK_LAUNCH(128, 128, 2, 2, false, 0) break;
    case 1: EVMI_PK_LAUNCH(64, 128, 1, 4, false, 1) break;
    case 2: EVMI_PK_LAUNCH(64, 64, 2, 2, false, 2) break;
    case 4: EVMI_PK_LAUNCH(64, 256, 1, 4, false, 4) break;
    case 5: EVMI_PK_LAUNCH(32, 256, 1, 4, false, 5) break;
    case 6: EVMI_PK_LAUNCH(128, 256, 2, 2, false, 6) break;
    case 7: EVMI_PK_LAUNCH(128, 256, 2, 4, false, 7) break;
    case 8: EVMI_PK_LAUNCH(128, 128, 2, 4, false, 8) break;
    case 10: EVMI_PK_LAUNCH(32, 512, 1, 4, false, 10) break;
    case 9: EVMI_PK_LAUNCH(128, 128, 2, 2, true, 9) break;
    default: EVMI_PK_LAUNCH(32, 128, 1, 4, false, 3) break;
  }
#undef EVMI_PK_LAUNCH_AS
#undef EVMI_PK_LAUNCH
  EVMI_LAUNCH_CHECK("conv_cbt_bf16_pk");
  return EVMI_OK;
}

static WfragArgs make_wfrag_args(const ConvPkArgs& a, const PkPlan& pl, const float* w, uint4* wf, int wmode, int rows_g, int kch_g, int k_full,
                                 int stride_full) {
  WfragArgs fa;
  fa.w = w; fa.wf = reinterpret_cast<unsigned*>(wf); fa.rows_g = rows_g; fa.kch_g = kch_g; fa.kt = a.k; fa.MB = a.mblocks; fa.octs = a.octs;
  fa.kblocks = a.kblocks; fa.mode = wmode; fa.k_full = k_full; fa.stride = stride_full; fa.phase_stride_words = a.wf_phase_stride * 4;
  fa.tab = reinterpret_cast<int2*>(wf + a.wf_phase_stride * a.phases); fa.kb_step = a.kb_step; fa.xrow = a.xrow; fa.dil = a.dil;
  fa.gx = a.kblocks; fa.gy = pl.groups * a.mblocks; fa.gz = a.phases;
  return fa;
}

// stage 0: pack + weight fragments + convolution in one call; 1: the preparation alone -- pack AND fragments, one launch; the packed
// input is left at the head of ws; 2: the convolution alone, on what stage 1 left in the same ws (same shape, hence the same plan: no
// launch in front of it); 3: fragments + convolution on a packed input that something else put at the head of ws (a producer's
// epilogue, LayerNorm's pack)
static int launch_pk(ConvPkArgs a, const PkPlan& pl, const float* x, const float* w, float* ws, long long ws_elems, int wmode,
                     int rows_g, int kch_g, int k_full, int stride_full, hipStream_t stream, PkInputFusion in = PkInputFusion(),
                     int stage = 0) {
  const long long need = (pl.xp_units + pl.wf_units) * 4 + pl.part_elems;
  if (!ws || ws_elems < need || (reinterpret_cast<uintptr_t>(ws) & 15))
    return fail(EVMI_ERR_INVALID_ARG, "conv_cbt_bf16_pk: workspace missing, too small or unaligned");
  uint4* xp = reinterpret_cast<uint4*>(ws);
  uint4* wf = xp + pl.xp_units;
  a.part = reinterpret_cast<float*>(wf + pl.wf_units);
  PackArgs pa = make_pack_args(x, xp, pl.cin_g, a.octs, a.B, pl.t_in, a.Tp, pl.PL,
                               (int)(pl.xp_units - (long long)pl.groups * a.octs * a.B * a.Tp), pl.groups);
  pa.pre_slope = in.pre_slope; pa.mask = in.mask; pa.mask_slope = in.mask_slope;
  pa.fuse = in.fuse; pa.fuse_scale = in.fuse_scale; pa.aux = in.aux; pa.p_drop = in.p_drop; pa.seed = in.seed;
  WfragArgs fa = make_wfrag_args(a, pl, w, wf, wmode, rows_g, kch_g, k_full, stride_full);
  if (stage == 2) fa.gx = fa.gy = fa.gz = 0;
  if (stage == 2 || stage == 3) pa.gx = pa.gy = pa.gz = 0;
  const long long n_prep = (long long)pa.gx * pa.gy * pa.gz + (long long)fa.gx * fa.gy * fa.gz;
  if (n_prep > 0x7fffffffLL) return fail(EVMI_ERR_UNSUPPORTED, "conv_cbt_bf16_pk: grid limits (preparation pass)");
  if (n_prep > 0) hipLaunchKernelGGL(prep_pk_kernel, dim3((unsigned)n_prep), dim3(256), 0, stream, pa, fa);
  if (stage == 1) {
    EVMI_LAUNCH_CHECK("conv_cbt_bf16_pk (pack)");
    return EVMI_OK;
  }
  a.tab = fa.tab;
  a.xp = xp;
  a.wf = wf;
  if (int rc = launch_pk_tile(a, pl, stream)) return rc;
  if (a.ksplit > 1 && a.po.y) {
    const long long cols = std::max<long long>(a.part_ld, a.po.tail ? a.po.pad_end : 0);
    hipLaunchKernelGGL(conv_pk_reduce_flat_kernel, dim3((unsigned)((cols + 1023) / 1024), pl.c_out / 8, a.phases), dim3(256), 0, stream, a);
    EVMI_LAUNCH_CHECK("conv_pk_reduce_flat");
  } else if (a.ksplit > 1) {
    hipLaunchKernelGGL(conv_pk_reduce_kernel, dim3((unsigned)((a.part_ld + 255) / 256), pl.c_out, a.phases), dim3(256), 0, stream, a);
    EVMI_LAUNCH_CHECK("conv_pk_reduce");
  }
  return EVMI_OK;
}

static const char* plan_fwd_pk(ConvPkArgs& a, PkPlan& pl, int B, int c_in, int t_in, int c_out, int t_out_total, int n_out, int k, int stride,
                               int pad, int dil, int groups, int out_stride, int out_offset) {
  if (groups <= 0 || c_in % groups || c_out % groups) return "groups";
  if (pad < 0) return "negative padding";
  a.B = B; a.t_out_total = t_out_total; a.cout_g = c_out / groups; a.k = k; a.stride = stride; a.dil = dil;
  a.out_stride = out_stride; a.phases = 1;
  a.ph_nout[0] = n_out; a.ph_off[0] = out_offset;
  const int ph_pad[1] = {pad};
  return plan_pk(a, c_in / groups, t_in, groups, ph_pad, pl);
}

// dx [c_in][B][t_in] from dy [c_out][B][t_out]: min(stride, k) polyphase stride-1 convolutions of dy in one launch
static const char* plan_dgrad_pk(ConvPkArgs& a, PkPlan& pl, int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad,
                                 int dil, int groups) {
  if (groups <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups) return "bad shape";
  if (stride > 1 && dil != 1) return "strided and dilated";
  if (stride > 8) return "stride above 8";
  const int phases = std::min(stride, k), M = (k + stride - 1) / stride;
  a.B = B; a.t_out_total = t_in; a.cout_g = c_in / groups; a.k = M; a.stride = 1; a.dil = stride == 1 ? dil : 1;
  a.out_stride = stride; a.phases = phases; a.accumulate = 0; a.bias = nullptr; a.act = 0;
  int ph_pad[8];
  for (int phi = 0; phi < phases; ++phi) {
    if (stride == 1) {
      ph_pad[0] = dil * (k - 1) - pad; a.ph_nout[0] = t_in; a.ph_off[0] = 0;
    } else {
      const int num = pad - phi;  // first q with stride*q + phi - pad >= 0
      const int q0 = num > 0 ? (num + stride - 1) / stride : 0;
      const int q_hi = (t_in - 1 + pad - phi) >= 0 ? (t_in - 1 + pad - phi) / stride : -1;
      ph_pad[phi] = (M - 1) - q0;
      a.ph_nout[phi] = std::max(0, q_hi - q0 + 1);
      a.ph_off[phi] = stride * q0 + phi - pad;
    }
    if (ph_pad[phi] < 0) return "negative phase padding";
  }
  return plan_pk(a, c_out / groups, t_out, groups, ph_pad, pl);
}


// ---- flat packed tensors in, flat packed tensors out (the discriminator chains) ---------------------------------------------------
// The input is ONE long row: n_items items T units apart laid end to end, every item's gap zero; the kernel runs as a one-item
// convolution over it (B = 1: no item arithmetic in its window loads, the unit index advances by `stride` per column everywhere).
struct PkFlatShape {
  int mode;  // 0: forward; 1: input gradient (polyphase)
  int n_items, T, c_in, c_out, k, stride, pad, dil, groups;
};
// Weight fragments of a flat call depend on the layer alone (shape, direction, weights) -- not on the item count or the tile the
// planner picks: one fragment buffer serves every call of a layer in a phase of the step (discriminator step pair, generator step
// real / generated).  The K-block offset table and the split counters are per call and static: written once (evmi_conv_pkflat_tab).
struct FragGeom {
  int rows_g, kch_g, kt, MB, octs, kblocks, phases, groups;
  long long phase_stride_units;
};
static FragGeom frag_geom(int mode, int c_in, int c_out, int k, int stride, int groups) {
  FragGeom f;
  const int cin_g = c_in / groups, cout_g = c_out / groups;
  f.groups = groups;
  f.rows_g = mode == 0 ? cout_g : cin_g;
  f.kch_g = mode == 0 ? cin_g : cout_g;
  f.kt = mode == 0 ? k : (stride == 1 ? k : (k + stride - 1) / stride);
  f.phases = mode == 0 ? 1 : std::min(stride, k);
  f.MB = (f.rows_g + 31) / 32;
  f.octs = (f.kch_g + 7) / 8;
  f.kblocks = (f.octs * f.kt + 1) / 2;
  f.phase_stride_units = (long long)groups * f.MB * f.kblocks * 64;
  return f;
}
struct FragJob {
  const float* w;
  unsigned* wf;
  int mode, rows_g, kch_g, kt, MB, octs, kblocks, k_full, stride, phases;
  long long phase_stride_words;
};
constexpr int PKFLAT_MAX_JOBS = 32;
struct FragBatch {
  FragJob job[PKFLAT_MAX_JOBS];
  int start[PKFLAT_MAX_JOBS + 1];
  int n;
};
// One workgroup per (group, 32-row block, channel octet): its 256 (row, channel) pairs own k contiguous floats each -- the block is a
// set of contiguous runs (forward: 8 x k floats per row; input gradient: 32 x k floats per channel).  Every thread copies its run into
// the LDS first (each cache line requested once: the loop over (phase, tap) that picked single floats out of the runs re-fetched
// the lines per tap -- 5.2 GB per GAN step for 1.7 GB of weights and fragments, profiles/r04v_train_pmc_summary.json), then the half-fragments
// of every (phase, tap) are assembled from there.
__global__ __launch_bounds__(256) void wfrag_flat_kernel(FragBatch b) {
  extern __shared__ float runs[];  // [256][k_full + 1]
  int jn = 0;
  while (jn + 1 < b.n && (int)blockIdx.x >= b.start[jn + 1]) ++jn;
  const FragJob& f = b.job[jn];
  const int bid = blockIdx.x - b.start[jn];
  const int gmb = bid / f.octs, o = bid - gmb * f.octs;
  const int g = gmb / f.MB, mb = gmb - g * f.MB;
  const int t = threadIdx.x;
  const int r = f.mode == 0 ? t >> 3 : t & 31, c = f.mode == 0 ? t & 7 : t >> 5;
  const int row = mb * 32 + r, kc = o * 8 + c;
  const bool valid = row < f.rows_g && kc < f.kch_g;
  const int kt = f.kt, kf = f.k_full, ld = kf + 1;
  const float* src = f.mode == 0 ? f.w + ((long long)(g * f.rows_g + row) * f.kch_g + kc) * kf
                                 : f.w + ((long long)(g * f.kch_g + kc) * f.rows_g + row) * kf;
  float* mine = runs + t * ld;
  {
    // eight floats of the run requested before the first is stored (from clamped -- always valid -- addresses, selected afterwards):
    // one load, its wait and a store per trip was k dependent round trips per thread, 41 on the grouped layers (197 us per launch
    // for 140 MB: 0.7 TB/s)
    const float* safe = valid ? src : f.w;
    for (int j0 = 0; j0 < kf; j0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = safe[min(j0 + u, kf - 1)];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (j0 + u < kf) mine[j0 + u] = valid ? v[u] : 0.f;
    }
  }
  // The runs are re-read across threads: thread (row r, tap slot jq) assembles the eight channels of its row at taps jq, jq + 8, ...
  // into one 16-byte unit per (phase, tap) and stores it whole.  (One channel per thread, the pair's other half through ds_bpermute
  // and a 4-byte store per tap, was a dependent LDS round trip per tap and a quarter-filled store: 171 us for 47 MB.)
  __syncthreads();
  const int wr = t >> 3, jq = t & 7;
  const bool row_ok = mb * 32 + wr < f.rows_g;
  for (int phi = 0; phi < f.phases; ++phi) {
    const int m_phi = f.mode == 0 ? kt : (kf - phi + f.stride - 1) / f.stride, lead = kt - m_phi;
    uint4* dst = reinterpret_cast<uint4*>(f.wf + phi * f.phase_stride_words + (long long)gmb * f.kblocks * 256);
    for (int j = jq; j < kt; j += 8) {
      const int idx = f.mode == 0 ? j : phi + f.stride * (m_phi - 1 - (j - lead));
      float v[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) {
        const int owner = f.mode == 0 ? wr * 8 + cc : cc * 32 + wr;  // the thread that staged (row wr, channel cc)
        v[cc] = (j >= lead && row_ok) ? runs[owner * ld + idx] : 0.f;  // (a channel past kch_g was staged as zeros)
      }
      const int h = o * kt + j;
      uint4 u;
      u.x = pk_bf16x2(v[0], v[1]); u.y = pk_bf16x2(v[2], v[3]); u.z = pk_bf16x2(v[4], v[5]); u.w = pk_bf16x2(v[6], v[7]);
      dst[(h >> 1) * 64 + (h & 1) * 32 + wr] = u;
    }
    if (o == f.octs - 1 && ((f.octs * kt) & 1) && jq == 0)  // odd tail: the upper half of the last K block is zero weights
      dst[(f.kblocks - 1) * 64 + 32 + wr] = make_uint4(0u, 0u, 0u, 0u);
  }
}

// per K block: window offsets (units) of its two halves (as wfrag_pk_block writes them)
__global__ void flat_tab_kernel(int2* tab, int kblocks, int kb_step, int kt, int octs, int xrow, int dil) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= kblocks) return;
  const int o_lo = (2 * (q / kb_step) * kb_step) / kt;
  const int h0 = 2 * q, h1 = h0 + 1 < octs * kt ? h0 + 1 : h0;
  tab[q] = make_int2((h0 / kt - o_lo) * xrow + (h0 % kt) * dil, (h1 / kt - o_lo) * xrow + (h1 % kt) * dil);
}

static const char* plan_flat(const PkFlatShape& sh, ConvPkArgs& a, PkPlan& pl) {
  if (sh.n_items <= 0 || sh.T <= 0) return "bad flat shape";
  if ((sh.c_in / std::max(1, sh.groups)) % 8 || (sh.c_out / std::max(1, sh.groups)) % 8) return "channels per group must be multiples of 8";
  const long long len = (long long)sh.n_items * sh.T;
  if (len * std::max(1, sh.stride) >= (1LL << 30)) return "flat row too long";
  const char* why;
  if (sh.mode == 0) {
    if (sh.T % sh.stride) return "item pitch must be a multiple of the stride";
    const int n_cols = (int)(len / sh.stride);
    why = plan_fwd_pk(a, pl, 1, sh.c_in, (int)len, sh.c_out, n_cols, n_cols, sh.k, sh.stride, sh.pad, sh.dil, sh.groups, 1, 0);
  } else {
    why = plan_dgrad_pk(a, pl, 1, sh.c_in, (int)(len * sh.stride), sh.c_out, (int)len, sh.k, sh.stride, sh.pad, sh.dil, sh.groups);
  }
  if (why) return why;
  const FragGeom f = frag_geom(sh.mode, sh.c_in, sh.c_out, sh.k, sh.stride, sh.groups);
  if (f.kt != a.k || f.MB != a.mblocks || f.octs != a.octs || f.kblocks != a.kblocks || f.phases != a.phases || f.phase_stride_units != a.wf_phase_stride)
    return "fragment geometry (internal)";
  return nullptr;
}
// workspace of a call: [offset table][split partial tiles]
struct FlatWs {
  long long tab_units, total_floats;
};
static FlatWs flat_ws(const ConvPkArgs& a, const PkPlan& pl) {
  FlatWs w;
  w.tab_units = ((long long)a.kblocks * 8 + 15) / 16;
  w.total_floats = w.tab_units * 4 + pl.part_elems;
  return w;
}

static int launch_flat(const PkFlatShape& sh, const void* in_dev, long long in_plane, const void* wf_dev, const float* bias, float* ws,
                       long long ws_elems, ConvPkArgs::FlatOut po, int act, float act_param, hipStream_t stream) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_flat(sh, a, pl)) return fail(EVMI_ERR_UNSUPPORTED, std::string("conv_pkflat: ") + why);
  const FlatWs fw = flat_ws(a, pl);
  if (!ws || ws_elems < fw.total_floats || (reinterpret_cast<uintptr_t>(ws) & 15)) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat: workspace missing, too small or unaligned");
  if (!in_dev || !wf_dev || !po.y || (reinterpret_cast<uintptr_t>(wf_dev) & 15)) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat: null / unaligned pointer");
  a.tab = reinterpret_cast<const int2*>(ws);
  a.wf = reinterpret_cast<const uint4*>(wf_dev);
  a.part = ws + fw.tab_units * 4;
  a.Tp = (int)in_plane;                                            // B == 1: the octet rows are `Tp` units apart
  a.xp = reinterpret_cast<const uint4*>(in_dev) - pl.PL;           // the front guard of the tensor is the left padding
  a.bias = bias; a.act = act; a.act_param = act_param; a.accumulate = 0;
  a.po = po;
  if (int rc = launch_pk_tile(a, pl, stream)) return rc;
  if (a.ksplit > 1) {
    int n_max = 0;
    for (int p = 0; p < a.phases; ++p) n_max = std::max(n_max, a.ph_nout[p]);
    hipLaunchKernelGGL(conv_pk_reduce_flat_kernel, dim3((unsigned)((n_max + 1023) / 1024), pl.c_out / 8, a.phases), dim3(256), 0, stream, a);
    EVMI_LAUNCH_CHECK("conv_pk_reduce_flat");
  }
  return EVMI_OK;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

/* Floats of workspace the packed bf16 convolution needs (packed input + weight fragments); 0 = shape not taken by it. */
long long evmi_conv1d_cbt_bf16pk_ws_elems(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil,
                                          int groups) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (plan_fwd_pk(a, pl, B, c_in, t_in, c_out, n_out, n_out, k, stride, pad, dil, groups, 1, 0)) return 0;
  return (pl.xp_units + pl.wf_units) * 4 + pl.part_elems;
}

int evmi_conv1d_cbt_bf16pk_plan(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (plan_fwd_pk(a, pl, B, c_in, t_in, c_out, n_out, n_out, k, stride, pad, dil, groups, 1, 0)) return -1;
  return pl.ti + 16 * (a.ksplit > 1 ? a.ksplit : 0);
}

int evmi_conv1d_dgrad_cbt_bf16pk_plan(int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil,
                                      int groups) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (plan_dgrad_pk(a, pl, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups)) return -1;
  return pl.ti + 16 * (a.ksplit > 1 ? a.ksplit : 0);
}

int evmi_conv1d_cbt_bf16pk(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev,
                           long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out_total, int n_out, int k, int stride,
                           int pad, int dil, int groups, int out_stride, int out_offset, int accumulate, int act, float act_param,
                           void* stream) {
  if (!x_dev || !w_dev || !y_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk: null pointer");
  if (act < 0 || act > 4 || (act && accumulate)) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk: activation (0..4, not with accumulate)");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_fwd_pk(a, pl, B, c_in, t_in, c_out, t_out_total, n_out, k, stride, pad, dil, groups, out_stride, out_offset))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_cbt_bf16pk: ") + why);
  a.bias = bias_dev; a.y = y_dev; a.accumulate = accumulate; a.act = act; a.act_param = act_param;
  return launch_pk(a, pl, x_dev, w_dev, ws_dev, ws_elems, 0, c_out / groups, c_in / groups, k, stride, (hipStream_t)stream);
}

/* LayerNorm in front of a pointwise layer, written as that layer's packed input (the layers of evmi_conv1d_bf16pk_shares_packed: tight
 * items, B * t a multiple of 64; c_in 128 or 256), and the layer itself on an input that is already packed in the head of ws:
 *   evmi_layernorm_pack_bf16pk(x, gamma, beta, ws, ...)   ws head <- packed bf16 LayerNorm(x) (+ the zero slack the kernels read past it)
 *   evmi_conv1d_cbt_bf16pk_prepacked(w, bias, y, ws, ...) y = act(conv(that) + bias); ws as evmi_conv1d_cbt_bf16pk's, same geometry
 * The normalised tensor is never stored in fp32; the head of ws is what the layer's weight gradient reads again (..._wgrad_..._prepacked). */
static int layernorm_pack_impl(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* ws_dev, long long ws_elems, int B, int c_in, int t_in,
                               int c_out, float eps, const float* w_dev, const float* w2_dev, float* ws2_dev, long long ws2_elems, int c_out2,
                               void* stream) {
  if (!x_dev || !gamma_dev || !beta_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "layernorm_pack_bf16pk: null pointer");
  if (c_in != 128 && c_in != 256) return fail(EVMI_ERR_UNSUPPORTED, "layernorm_pack_bf16pk: 128 or 256 channels");
  if (!pk_shared_items(B, t_in)) return fail(EVMI_ERR_UNSUPPORTED, "layernorm_pack_bf16pk: B * t must be a multiple of 64 (or one item)");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_fwd_pk(a, pl, B, c_in, t_in, c_out, t_in, t_in, 1, 1, 0, 1, 1, 1, 0))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("layernorm_pack_bf16pk: ") + why);
  if (a.Tp != pk_shared_pitch(B, t_in) || pl.PL != 0) return fail(EVMI_ERR_UNSUPPORTED, "layernorm_pack_bf16pk: the layer's items are not packed tight");
  if (ws_elems < (pl.xp_units + pl.wf_units) * 4 + pl.part_elems || (reinterpret_cast<uintptr_t>(ws_dev) & 15))
    return fail(EVMI_ERR_INVALID_ARG, "layernorm_pack_bf16pk: workspace too small or unaligned");
  const long long N = (long long)B * t_in, plane = (long long)B * a.Tp;
  const int slack = (int)(pl.xp_units - (long long)a.octs * plane);
  uint4* xp = reinterpret_cast<uint4*>(ws_dev);
  // the fragment jobs riding in the same launch (empty grids when the weights are not given)
  WfragArgs f1 = {}, f2 = {};
  if (w_dev) f1 = make_wfrag_args(a, pl, w_dev, xp + pl.xp_units, 0, c_out, c_in, 1, 1);
  if (w2_dev) {
    ConvPkArgs a2 = {};
    PkPlan pl2;
    if (const char* why = plan_fwd_pk(a2, pl2, B, c_out, t_in, c_out2, t_in, t_in, 1, 1, 0, 1, 1, 1, 0))
      return fail(EVMI_ERR_UNSUPPORTED, std::string("layernorm_pack_bf16pk (second layer): ") + why);
    if (!w_dev || !ws2_dev || ws2_elems < (pl2.xp_units + pl2.wf_units) * 4 + pl2.part_elems || (reinterpret_cast<uintptr_t>(ws2_dev) & 15))
      return fail(EVMI_ERR_INVALID_ARG, "layernorm_pack_bf16pk: the second layer's workspace is missing, too small or unaligned");
    f2 = make_wfrag_args(a2, pl2, w2_dev, reinterpret_cast<uint4*>(ws2_dev) + pl2.xp_units, 0, c_out2, c_out, 1, 1);
  }
  const unsigned n_ln = (unsigned)((plane + 63) / 64);
  const long long n_blk = (long long)n_ln + (long long)f1.gx * f1.gy * f1.gz + (long long)f2.gx * f2.gy * f2.gz;
  if (n_blk > 0x7fffffffLL) return fail(EVMI_ERR_UNSUPPORTED, "layernorm_pack_bf16pk: grid limits");
  const dim3 grid((unsigned)n_blk);
  if (c_in == 256) hipLaunchKernelGGL(layernorm_pack_kernel<64>, grid, dim3(256), 0, (hipStream_t)stream, x_dev, gamma_dev, beta_dev, xp, c_in, N, plane, eps, slack, n_ln, f1, f2);
  else hipLaunchKernelGGL(layernorm_pack_kernel<32>, grid, dim3(256), 0, (hipStream_t)stream, x_dev, gamma_dev, beta_dev, xp, c_in, N, plane, eps, slack, n_ln, f1, f2);
  EVMI_LAUNCH_CHECK("layernorm_pack");
  return EVMI_OK;
}

int evmi_layernorm_pack_bf16pk(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* ws_dev, long long ws_elems, int B,
                               int c_in, int t_in, int c_out, float eps, void* stream) {
  return layernorm_pack_impl(x_dev, gamma_dev, beta_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, eps, nullptr, nullptr, nullptr, 0, 0, stream);
}

/* ... with the weight fragments of the layer(s) behind the LayerNorm prepared by the SAME launch: w_dev [c_out][c_in] into ws_dev (the
 * layer then runs with fragments_ready = 1 / in_mode 3) and, for a feed-forward block, w2_dev [c_out2][c_out] into ws2_dev, the second
 * layer's workspace (NULL: none). */
int evmi_layernorm_pack_bf16pk_w(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* ws_dev, long long ws_elems, int B,
                                 int c_in, int t_in, int c_out, float eps, const float* w_dev, const float* w2_dev, float* ws2_dev,
                                 long long ws2_elems, int c_out2, void* stream) {
  if (!w_dev) return fail(EVMI_ERR_INVALID_ARG, "layernorm_pack_bf16pk_w: null pointer");
  return layernorm_pack_impl(x_dev, gamma_dev, beta_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, eps, w_dev, w2_dev, ws2_dev, ws2_elems, c_out2, stream);
}

int evmi_conv1d_cbt_bf16pk_prepacked(const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev, long long ws_elems, int B,
                                     int c_in, int t_in, int c_out, int act, float act_param, int fragments_ready, void* stream) {
  if (!w_dev || !y_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_prepacked: null pointer");
  if (act < 0 || act > 4) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_prepacked: activation");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_fwd_pk(a, pl, B, c_in, t_in, c_out, t_in, t_in, 1, 1, 0, 1, 1, 1, 0))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_cbt_bf16pk_prepacked: ") + why);
  a.bias = bias_dev; a.y = y_dev; a.accumulate = 0; a.act = act; a.act_param = act_param;
  // (stage 3: the pack is an empty grid; the fp32 input pointer is never read.  stage 2: the fragments are there too)
  return launch_pk(a, pl, reinterpret_cast<const float*>(ws_dev), w_dev, ws_dev, ws_elems, 0, c_out, c_in, 1, 1, (hipStream_t)stream, PkInputFusion(),
                   fragments_ready ? 2 : 3);
}

/* The same with the fusions of a residual block's forward: the input passes through leaky_relu(., pre_slope) while it is packed
 * (pre_slope 1 = none) and `residual` [c_out][B][t_out_total] (may be NULL) is added behind the activation. */
int evmi_conv1d_cbt_bf16pk_fused(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev,
                                 long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out_total, int n_out, int k, int stride,
                                 int pad, int dil, int groups, int act, float act_param, float pre_slope, const float* residual_dev,
                                 void* stream) {
  if (!x_dev || !w_dev || !y_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_fused: null pointer");
  if (act < 0 || act > 4) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_fused: activation");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_fwd_pk(a, pl, B, c_in, t_in, c_out, t_out_total, n_out, k, stride, pad, dil, groups, 1, 0))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_cbt_bf16pk_fused: ") + why);
  a.bias = bias_dev; a.y = y_dev; a.accumulate = 0; a.act = act; a.act_param = act_param; a.res = residual_dev;
  PkInputFusion in;
  in.pre_slope = pre_slope;
  return launch_pk(a, pl, x_dev, w_dev, ws_dev, ws_elems, 0, c_out / groups, c_in / groups, k, stride, (hipStream_t)stream, in);
}

long long evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil,
                                                int groups) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (plan_dgrad_pk(a, pl, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups)) return 0;
  return (pl.xp_units + pl.wf_units) * 4 + pl.part_elems;
}

int evmi_conv1d_dgrad_cbt_bf16pk(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev, long long ws_elems, int B,
                                 int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil, int groups,
                                 void* stream) {
  if (!dy_dev || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk: null pointer");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_dgrad_pk(a, pl, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_dgrad_cbt_bf16pk: ") + why);
  a.y = dx_dev;
  return launch_pk(a, pl, dy_dev, w_dev, ws_dev, ws_elems, 1, c_in / groups, c_out / groups, k, stride, (hipStream_t)stream);
}

/* The same call in two steps, so that a caller can put something between them: stage 1 packs dy into the head of ws, prepares the weight
 * fragments behind it (one launch) and returns; stage 2 (same arguments, same ws, untouched in between) runs the convolution on them.
 * Stage 3: fragments + convolution on a packed dy that something else left at the head of ws (evmi_conv1d_dgrad_cbt_bf16pk_ffn_down).  What goes
 * between them in training: the fork of the weight-gradient stream -- the weight gradient of a pointwise layer reads that packed dy
 * (evmi_conv1d_wgrad_cbt_bf16pk_prepacked) and can then run BESIDE the input gradient instead of behind it. */
int evmi_conv1d_dgrad_cbt_bf16pk_staged(int stage, const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev, long long ws_elems,
                                        int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil, int groups,
                                        void* stream) {
  if (stage < 1 || stage > 3) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged: stage 1, 2 or 3");
  if (!dy_dev || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged: null pointer");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_dgrad_pk(a, pl, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_dgrad_cbt_bf16pk_staged: ") + why);
  a.y = dx_dev;
  return launch_pk(a, pl, dy_dev, w_dev, ws_dev, ws_elems, 1, c_in / groups, c_out / groups, k, stride, (hipStream_t)stream, PkInputFusion(), stage);
}

/* The two halves of a feed-forward block's middle -- dense2(dropout(silu(a), p)) and its backward -- with the activation and the mask
 * applied while the tensors are packed (PackArgs::fuse), so that neither dropout(silu(a)) nor dropout(ds) * silu'(a) exists in fp32:
 *   _silu_dropout:         y = conv(dropout(silu(x), p)) + bias                    (x = a; stride 1, no dilation, one group)
 *   _staged_silu_dropout:  evmi_conv1d_dgrad_cbt_bf16pk_staged on dy = dropout(ds, p) * silu'(pre) (stage 1 packs THAT; stage 2 as usual)
 * Mask stream and arithmetic of evmi_dropout_fused_f32 modes 2 / 3 (seed_value + *seed_base_dev, element index = index in the tensor). */
int evmi_conv1d_cbt_bf16pk_silu_dropout(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev,
                                        long long ws_elems, int B, int c_in, int t_in, int c_out, int k, int pad, float p,
                                        unsigned long long seed_value, const unsigned long long* seed_base_dev, void* stream) {
  if (!x_dev || !w_dev || !y_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_silu_dropout: null pointer");
  if (p < 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_silu_dropout: p outside [0, 1)");
  const int t_out = t_in + 2 * pad - (k - 1);
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_fwd_pk(a, pl, B, c_in, t_in, c_out, t_out, t_out, k, 1, pad, 1, 1, 1, 0))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_cbt_bf16pk_silu_dropout: ") + why);
  a.bias = bias_dev; a.y = y_dev; a.accumulate = 0; a.act = 0; a.act_param = 0.f;
  PkInputFusion in;
  in.fuse = 1; in.p_drop = p; in.seed = SeedArg{seed_value, seed_base_dev};
  return launch_pk(a, pl, x_dev, w_dev, ws_dev, ws_elems, 0, c_out, c_in, k, 1, (hipStream_t)stream, in);
}

int evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout(int stage, const float* ds_dev, const float* pre_dev, float p, unsigned long long seed_value,
                                                     const unsigned long long* seed_base_dev, const float* w_dev, float* dx_dev, float* ws_dev,
                                                     long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out, int k, int stride,
                                                     int pad, int dil, int groups, void* stream) {
  if (stage != 1 && stage != 2) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged_silu_dropout: stage 1 or 2");
  if (!ds_dev || !pre_dev || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged_silu_dropout: null pointer");
  if (p < 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged_silu_dropout: p outside [0, 1)");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_dgrad_pk(a, pl, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_dgrad_cbt_bf16pk_staged_silu_dropout: ") + why);
  a.y = dx_dev;
  PkInputFusion in;
  in.fuse = 2; in.aux = pre_dev; in.p_drop = p; in.seed = SeedArg{seed_value, seed_base_dev};
  return launch_pk(a, pl, ds_dev, w_dev, ws_dev, ws_elems, 1, c_in / groups, c_out / groups, k, stride, (hipStream_t)stream, in, stage);
}

/* A Conformer sub-layer's LAST pointwise layer with the residual add and the dropout behind it in the epilogue, and the matching pack
 * of the backward (the FastSpeech2 step's residual_dropout sites: feed-forward blocks, attention out_proj, convolution module):
 *   _resdrop:        y = residual + out_scale * dropout(conv(in) + bias, out_p)       (k = 1, stride 1, one group)
 *                    in_mode 0: in = x;  1: in = dropout(silu(x), in_p) (the feed-forward middle, as _silu_dropout);  2: the packed
 *                    input already sits at the head of ws (as _prepacked; x is not read)
 *   _staged_dropout: evmi_conv1d_dgrad_cbt_bf16pk_staged on dz = scale * dropout(dy, p): stage 1 packs THAT (what the layer's input
 *                    gradient, weight gradient and bias gradient read); stage 2 as usual
 * Mask streams and arithmetic of evmi_dropout_fused_f32 modes 1 / 2 / 4 (seed + *seed_base_dev; element index = index in the tensor). */
int evmi_conv1d_cbt_bf16pk_resdrop(int in_mode, const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                                   float* y_dev, float* ws_dev, long long ws_elems, int B, int c_in, int t_in, int c_out, float in_p,
                                   unsigned long long in_seed, float out_p, unsigned long long out_seed, float out_scale,
                                   const unsigned long long* seed_base_dev, void* stream) {
  if (!w_dev || !y_dev || !residual_dev || (in_mode < 2 && !x_dev)) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_resdrop: null pointer");
  if (in_mode < 0 || in_mode > 3) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_resdrop: in_mode 0 .. 3");
  if (in_p < 0.f || in_p >= 1.f || out_p < 0.f || out_p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_resdrop: p outside [0, 1)");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_fwd_pk(a, pl, B, c_in, t_in, c_out, t_in, t_in, 1, 1, 0, 1, 1, 1, 0))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_cbt_bf16pk_resdrop: ") + why);
  a.bias = bias_dev; a.y = y_dev; a.accumulate = 0; a.act = 0; a.act_param = 0.f; a.res = residual_dev;
  a.drop_p = out_p; a.drop_fac = out_scale / (1.f - out_p); a.drop_seed = SeedArg{out_seed, seed_base_dev};
  if (out_p == 0.f) {  // no mask: the scale alone goes through the same factor (a.drop_p > 0 gates the hash)
    if (out_scale != 1.f) return fail(EVMI_ERR_UNSUPPORTED, "conv1d_cbt_bf16pk_resdrop: a scale needs out_p > 0");
  }
  PkInputFusion in;
  if (in_mode == 1) {
    in.fuse = 1; in.p_drop = in_p; in.seed = SeedArg{in_seed, seed_base_dev};
  }
  if (in_mode >= 2)  // (3: the weight fragments are in ws too -- evmi_layernorm_pack_bf16pk_w)
    return launch_pk(a, pl, reinterpret_cast<const float*>(ws_dev), w_dev, ws_dev, ws_elems, 0, c_out, c_in, 1, 1, (hipStream_t)stream, PkInputFusion(),
                     in_mode == 3 ? 2 : 3);
  return launch_pk(a, pl, x_dev, w_dev, ws_dev, ws_elems, 0, c_out, c_in, 1, 1, (hipStream_t)stream, in);
}

int evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout(int stage, const float* dy_dev, float p, unsigned long long seed_value,
                                                const unsigned long long* seed_base_dev, float scale, const float* w_dev, float* dx_dev,
                                                float* ws_dev, long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out, int k,
                                                int stride, int pad, int dil, int groups, void* stream) {
  if (stage != 1 && stage != 2) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged_dropout: stage 1 or 2");
  if (!dy_dev || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged_dropout: null pointer");
  if (p <= 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_staged_dropout: p outside (0, 1)");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_dgrad_pk(a, pl, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_dgrad_cbt_bf16pk_staged_dropout: ") + why);
  a.y = dx_dev;
  PkInputFusion in;
  in.fuse = 3; in.fuse_scale = scale; in.p_drop = p; in.seed = SeedArg{seed_value, seed_base_dev};
  return launch_pk(a, pl, dy_dev, w_dev, ws_dev, ws_elems, 1, c_in / groups, c_out / groups, k, stride, (hipStream_t)stream, in, stage);
}

/* A feed-forward block's wide middle tensor kept PACKED in both directions (pointwise layers c_in -> c_mid -> c_out on tight items,
 * the geometry of evmi_conv1d_bf16pk_shares_packed): neither the pre-activation a, nor dropout(silu(a)), nor the gradient
 * dropout(ds) * silu'(a) is ever stored in fp32, and no pack pass reads them back.
 *   _ffn_up:   the first layer on the packed input at the head of ws (as _prepacked), its result a = conv + bias leaving the epilogue twice:
 *                a_pk [c_mid / 8][B * t] 16-byte units  <- bf16(a)                          (the backward's silu' reads it)
 *                head of next_ws                        <- bf16(dropout(silu(a), p))        (next_ws = the SECOND layer's workspace, whose
 *                head is that layer's packed input: run it with evmi_conv1d_cbt_bf16pk_prepacked / _resdrop in_mode 2)
 *   _ffn_down_dgrad:  the second layer's input gradient on the packed dz at the head of ws (stage 1 of ..._dgrad_..._staged[_dropout] put
 *              it there), its result ds leaving the epilogue as
 *                head of next_ws <- bf16(dropout(ds, p) * silu'(a_pk))   (next_ws = the workspace of the FIRST layer's input gradient,
 *                whose head is that layer's packed dy: run evmi_conv1d_dgrad_cbt_bf16pk_staged stage 2 on it; the weight gradient and the
 *                bias gradient read the same units)
 * Same mask stream and arithmetic as evmi_conv1d_cbt_bf16pk_silu_dropout / ..._staged_silu_dropout (seed_value + *seed_base_dev, element
 * index = index in the fp32 tensor that is no longer stored); the one difference: silu' is taken at bf16(a) instead of a. */
static int ffn_tail_check(const ConvPkArgs& a2, const PkPlan& pl2, int B, int t, int c_mid, float* next_ws, long long next_ws_elems, const char* who) {
  if (a2.Tp != pk_shared_pitch(B, t) || pl2.PL != 0 || !pk_shared_items(B, t))
    return fail(EVMI_ERR_UNSUPPORTED, std::string(who) + ": the consumer's items are not packed tight");
  if ((long long)B * t >= (1LL << 31) || c_mid % 8 || (long long)c_mid * B * t >= (1LL << 33))
    return fail(EVMI_ERR_UNSUPPORTED, std::string(who) + ": shape (channels a multiple of 8, fewer than 2^33 elements)");
  if (!next_ws || next_ws_elems < (pl2.xp_units + pl2.wf_units) * 4 + pl2.part_elems || (reinterpret_cast<uintptr_t>(next_ws) & 15))
    return fail(EVMI_ERR_INVALID_ARG, std::string(who) + ": the consumer's workspace is missing, too small or unaligned");
  return EVMI_OK;
}

int evmi_conv1d_cbt_bf16pk_ffn_up(const float* w_dev, const float* bias_dev, float* ws_dev, long long ws_elems, void* a_pk_dev,
                                  float* next_ws_dev, long long next_ws_elems, int B, int c_in, int t, int c_mid, int c_out, float p,
                                  unsigned long long seed_value, const unsigned long long* seed_base_dev, int fragments_ready, void* stream) {
  if (!w_dev || !ws_dev || (reinterpret_cast<uintptr_t>(a_pk_dev) & 15)) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_ffn_up: null / unaligned pointer");
  if (p < 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "conv1d_cbt_bf16pk_ffn_up: p outside [0, 1)");
  ConvPkArgs a = {}, a2 = {};
  PkPlan pl, pl2;
  if (const char* why = plan_fwd_pk(a, pl, B, c_in, t, c_mid, t, t, 1, 1, 0, 1, 1, 1, 0))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_cbt_bf16pk_ffn_up: ") + why);
  if (const char* why = plan_fwd_pk(a2, pl2, B, c_mid, t, c_out, t, t, 1, 1, 0, 1, 1, 1, 0))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_cbt_bf16pk_ffn_up (second layer): ") + why);
  if (int rc = ffn_tail_check(a2, pl2, B, t, c_mid, next_ws_dev, next_ws_elems, "conv1d_cbt_bf16pk_ffn_up")) return rc;
  const long long N = (long long)B * t, plane = (long long)B * a2.Tp;
  uint4* nxt = reinterpret_cast<uint4*>(next_ws_dev);
  // the zeros the pack pass that no longer runs wrote -- the slack behind the second layer's packed input and, in a single item's rows,
  // the units behind the last column -- come from the epilogue too (FlatOut::pad_end / zero_p)
  a.po.pad_end = (int)plane; a.po.zero_p = nxt + (long long)a2.octs * plane; a.po.zero_n = (int)(pl2.xp_units - (long long)a2.octs * plane);
  a.bias = bias_dev; a.y = nullptr; a.accumulate = 0; a.act = 0; a.act_param = 0.f;
  a.po.y = reinterpret_cast<uint4*>(a_pk_dev); a.po.y2 = nxt; a.po.plane = plane; a.po.Tc = a.po.valid = a.po.Ts = (int)N;
  a.po.tail = 1; a.po.drop_ld = N; a.po.Tm = (int)N; a.po.mplane = plane;
  if (!a_pk_dev) {  // (inference: no backward will ask for the pre-activation)
    a.po.y = nxt; a.po.y2 = nullptr; a.po.tail = 3;
  }
  a.drop_p = p; a.drop_seed = SeedArg{seed_value, seed_base_dev};
  return launch_pk(a, pl, reinterpret_cast<const float*>(ws_dev), w_dev, ws_dev, ws_elems, 0, c_mid, c_in, 1, 1, (hipStream_t)stream, PkInputFusion(),
                   fragments_ready ? 2 : 3);
}

int evmi_conv1d_dgrad_cbt_bf16pk_ffn_down(const float* w_dev, float* ws_dev, long long ws_elems, const void* a_pk_dev, float* next_ws_dev,
                                          long long next_ws_elems, int B, int c_in, int t, int c_mid, int c_out, float p,
                                          unsigned long long seed_value, const unsigned long long* seed_base_dev, void* stream) {
  if (!w_dev || !ws_dev || !a_pk_dev || (reinterpret_cast<uintptr_t>(a_pk_dev) & 15)) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_ffn_down: null / unaligned pointer");
  if (p < 0.f || p >= 1.f) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_ffn_down: p outside [0, 1)");
  ConvPkArgs a = {}, a1 = {};
  PkPlan pl, pl1;
  // the second layer (c_mid -> c_out): its input gradient has c_mid rows; the first layer's (c_in -> c_mid) input gradient reads them packed
  if (const char* why = plan_dgrad_pk(a, pl, B, c_mid, t, c_out, t, 1, 1, 0, 1, 1))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_dgrad_cbt_bf16pk_ffn_down: ") + why);
  if (const char* why = plan_dgrad_pk(a1, pl1, B, c_in, t, c_mid, t, 1, 1, 0, 1, 1))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_dgrad_cbt_bf16pk_ffn_down (first layer): ") + why);
  if (int rc = ffn_tail_check(a1, pl1, B, t, c_mid, next_ws_dev, next_ws_elems, "conv1d_dgrad_cbt_bf16pk_ffn_down")) return rc;
  const long long N = (long long)B * t, plane = (long long)B * a1.Tp;
  uint4* nxt = reinterpret_cast<uint4*>(next_ws_dev);
  a.po.pad_end = (int)plane; a.po.zero_p = nxt + (long long)a1.octs * plane; a.po.zero_n = (int)(pl1.xp_units - (long long)a1.octs * plane);
  a.y = nullptr;
  a.po.y = nxt; a.po.plane = plane; a.po.Tc = a.po.valid = a.po.Ts = (int)N;
  a.po.tail = 2; a.po.drop_ld = N; a.po.fm = reinterpret_cast<const uint4*>(a_pk_dev); a.po.Tm = (int)N; a.po.mplane = plane;
  a.drop_p = p; a.drop_seed = SeedArg{seed_value, seed_base_dev};
  return launch_pk(a, pl, reinterpret_cast<const float*>(ws_dev), w_dev, ws_dev, ws_elems, 1, c_mid, c_out, 1, 1, (hipStream_t)stream, PkInputFusion(), 2);
}

/* Input gradient with the fusions of a backward pass: dy is multiplied by (dy_mask > 0 ? 1 : dy_mask_slope) while it is packed
 * (dy_mask = the OUTPUT of the leaky ReLU behind the convolution: its backward; NULL = none); the result is multiplied by
 * (dx_mask > 0 ? 1 : dx_mask_slope) (dx_mask = the INPUT of the leaky ReLU in front of the convolution; NULL = none) and
 * `residual` [c_in][B][t_in] is added (the skip path's gradient; NULL = none).  pre_slope applies leaky_relu to the packed input
 * itself (a transposed convolution run as an input-gradient kernel: ConvTranspose(leaky_relu(x))). */
int evmi_conv1d_dgrad_cbt_bf16pk_fused(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev, long long ws_elems, int B,
                                       int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil, int groups,
                                       float pre_slope, const float* dy_mask_dev, float dy_mask_slope, const float* dx_mask_dev,
                                       float dx_mask_slope, const float* residual_dev, void* stream) {
  if (!dy_dev || !w_dev || !dx_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_dgrad_cbt_bf16pk_fused: null pointer");
  ConvPkArgs a = {};
  PkPlan pl;
  if (const char* why = plan_dgrad_pk(a, pl, B, c_in, t_in, c_out, t_out, k, stride, pad, dil, groups))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_dgrad_cbt_bf16pk_fused: ") + why);
  if ((dx_mask_dev || residual_dev) && k < stride)
    return fail(EVMI_ERR_UNSUPPORTED, "conv1d_dgrad_cbt_bf16pk_fused: positions no phase writes would miss the residual");
  a.y = dx_dev; a.out_mask = dx_mask_dev; a.out_mask_slope = dx_mask_slope; a.res = residual_dev;
  PkInputFusion in;
  in.pre_slope = pre_slope; in.mask = dy_mask_dev; in.mask_slope = dy_mask_slope;
  return launch_pk(a, pl, dy_dev, w_dev, ws_dev, ws_elems, 1, c_in / groups, c_out / groups, k, stride, (hipStream_t)stream, in);
}


/* ---- flat packed convolutions (include/evmi.h: "discriminator chains") ---- */
static PkFlatShape flat_shape(int mode, int n_items, int T, int c_in, int c_out, int k, int stride, int pad, int dil, int groups) {
  PkFlatShape sh;
  sh.mode = mode; sh.n_items = n_items; sh.T = T; sh.c_in = c_in; sh.c_out = c_out; sh.k = k; sh.stride = stride; sh.pad = pad; sh.dil = dil;
  sh.groups = groups;
  return sh;
}

long long evmi_conv_pkflat_ws_elems(int mode, int n_items, int T, int c_in, int c_out, int k, int stride, int pad, int dil, int groups) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (groups <= 0 || stride <= 0 || plan_flat(flat_shape(mode, n_items, T, c_in, c_out, k, stride, pad, dil, groups), a, pl)) return 0;
  return flat_ws(a, pl).total_floats;
}

int evmi_conv_pkflat_plan(int mode, int n_items, int T, int c_in, int c_out, int k, int stride, int pad, int dil, int groups) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (groups <= 0 || stride <= 0 || plan_flat(flat_shape(mode, n_items, T, c_in, c_out, k, stride, pad, dil, groups), a, pl)) return -1;
  return pl.ti + 16 * (a.ksplit > 1 ? a.ksplit : 0);
}

int evmi_conv_pkflat_tab(int mode, int n_items, int T, int c_in, int c_out, int k, int stride, int pad, int dil, int groups, float* ws_dev,
                         long long ws_elems, void* stream) {
  ConvPkArgs a = {};
  PkPlan pl;
  if (groups <= 0 || stride <= 0) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_tab: bad shape");
  if (const char* why = plan_flat(flat_shape(mode, n_items, T, c_in, c_out, k, stride, pad, dil, groups), a, pl))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv_pkflat_tab: ") + why);
  const FlatWs fw = flat_ws(a, pl);
  if (!ws_dev || ws_elems < fw.total_floats || (reinterpret_cast<uintptr_t>(ws_dev) & 15)) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_tab: workspace missing, too small or unaligned");
  hipLaunchKernelGGL(flat_tab_kernel, dim3((a.kblocks + 255) / 256), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<int2*>(ws_dev), a.kblocks, a.kb_step, a.k,
                     a.octs, a.xrow, a.dil);
  EVMI_LAUNCH_CHECK("conv_pkflat_tab");
  return EVMI_OK;
}

long long evmi_conv_pkflat_frag_elems(int mode, int c_in, int c_out, int k, int stride, int groups) {
  if (groups <= 0 || stride <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups || k <= 0) return 0;
  const FragGeom f = frag_geom(mode, c_in, c_out, k, stride, groups);
  return f.phase_stride_units * f.phases * 4;
}

int evmi_conv_pkflat_fragments(int n_jobs, const evmi_pkflat_job* jobs, void* stream) {
  if (n_jobs < 0 || (n_jobs && !jobs)) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_fragments: bad job list");
  for (int j0 = 0; j0 < n_jobs; j0 += PKFLAT_MAX_JOBS) {
    FragBatch b;
    b.n = std::min(PKFLAT_MAX_JOBS, n_jobs - j0);
    b.start[0] = 0;
    for (int j = 0; j < b.n; ++j) {
      const evmi_pkflat_job& jb = jobs[j0 + j];
      if (jb.groups <= 0 || jb.stride <= 0 || jb.c_in <= 0 || jb.c_out <= 0 || jb.c_in % jb.groups || jb.c_out % jb.groups || jb.k <= 0 || jb.mode < 0 || jb.mode > 1)
        return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_fragments: bad shape");
      const FragGeom f = frag_geom(jb.mode, jb.c_in, jb.c_out, jb.k, jb.stride, jb.groups);
      if (!jb.w || !jb.wf || jb.wf_elems < f.phase_stride_units * f.phases * 4 || (reinterpret_cast<uintptr_t>(jb.wf) & 15))
        return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_fragments: weights / fragment buffer missing, too small or unaligned");
      FragJob& q = b.job[j];
      q.w = jb.w; q.wf = reinterpret_cast<unsigned*>(jb.wf); q.mode = jb.mode; q.rows_g = f.rows_g; q.kch_g = f.kch_g; q.kt = f.kt; q.MB = f.MB;
      q.octs = f.octs; q.kblocks = f.kblocks; q.k_full = jb.k; q.stride = jb.stride; q.phases = f.phases; q.phase_stride_words = f.phase_stride_units * 4;
      const long long nb = (long long)f.groups * f.MB * f.octs;
      if (b.start[j] + nb > 0x7fffffffLL) return fail(EVMI_ERR_UNSUPPORTED, "conv_pkflat_fragments: grid limits");
      b.start[j + 1] = b.start[j] + (int)nb;
    }
    int k_max = 1;
    for (int j = 0; j < b.n; ++j) k_max = std::max(k_max, b.job[j].k_full);
    if (k_max > 48) return fail(EVMI_ERR_UNSUPPORTED, "conv_pkflat_fragments: kernels longer than 48 taps");
    if (b.start[b.n] > 0)
      hipLaunchKernelGGL(wfrag_flat_kernel, dim3((unsigned)b.start[b.n]), dim3(256), (size_t)256 * (k_max + 1) * sizeof(float), (hipStream_t)stream, b);
    EVMI_LAUNCH_CHECK("conv_pkflat_fragments");
  }
  return EVMI_OK;
}

int evmi_conv_pkflat_fwd(const void* x_pk, long long x_plane, const void* wf_dev, const float* bias_dev, void* y_pk, long long y_plane,
                         float* ws_dev, long long ws_elems, int n_items, int T_x, int c_in, int c_out, int k, int stride, int pad,
                         int dil, int groups, int valid, int T_store, int act, float act_param, void* stream) {
  if (act < 0 || act > 4) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_fwd: activation");
  if (groups <= 0 || stride <= 0 || T_x % stride) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_fwd: the item pitch must be a multiple of the stride");
  ConvPkArgs::FlatOut po = {};
  po.y = reinterpret_cast<uint4*>(y_pk); po.plane = y_plane; po.Tc = T_x / stride; po.valid = valid; po.Ts = T_store;
  return launch_flat(flat_shape(0, n_items, T_x, c_in, c_out, k, stride, pad, dil, groups), x_pk, x_plane, wf_dev, bias_dev, ws_dev, ws_elems,
                     po, act, act_param, (hipStream_t)stream);
}

int evmi_conv_pkflat_dgrad(const void* dy_pk, long long dy_plane, const void* wf_dev, void* dx_pk, long long dx_plane, float* ws_dev,
                           long long ws_elems, int n_items, int T_dy, int c_in, int c_out, int k, int stride, int pad, int dil,
                           int groups, int valid, int T_store, const void* mask_pk, const void* fm_pk, long long mask_plane, int T_mask,
                           float mask_slope, float fm_scale, void* stream) {
  if (groups <= 0 || stride <= 0) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_dgrad: bad shape");
  if (fm_pk && !mask_pk) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_dgrad: the feature-matching reference needs the mask tensor (the generated side's activation)");
  if (k < stride) return fail(EVMI_ERR_UNSUPPORTED, "conv_pkflat_dgrad: kernel shorter than the stride");
  ConvPkArgs::FlatOut po = {};
  po.y = reinterpret_cast<uint4*>(dx_pk); po.plane = dx_plane; po.Tc = T_dy * stride; po.valid = valid; po.Ts = T_store;
  po.mask = reinterpret_cast<const uint4*>(mask_pk); po.fm = reinterpret_cast<const uint4*>(fm_pk); po.mplane = mask_plane; po.Tm = T_mask;
  po.mask_slope = mask_slope; po.fm_scale = fm_scale;
  return launch_flat(flat_shape(1, n_items, T_dy, c_in, c_out, k, stride, pad, dil, groups), dy_pk, dy_plane, wf_dev, nullptr, ws_dev, ws_elems,
                     po, 0, 0.f, (hipStream_t)stream);
}

}  // extern "C"
