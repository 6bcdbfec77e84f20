// Weight gradient of the 1-D convolution with bf16 operands (fp32 accumulation), on PACKED copies of x and dy:
//
//   dw[co][ci][j] (+)= sum_{b, to} bf16(dy[co][b][to]) * bf16(x[ci][b][to*s + j*d - p])
//
// GEMM view: M = output channels, N = input channels (one tap j per accumulator tile), K = the flattened (item, position)
// index.  Both operands come from the packed layout of conv_pk_common.h -- 16-byte units of 8 channels at one position -- in
// which the K index runs ACROSS units: exactly the case of gfx950's transposing LDS read.  ds_read_b64_tr_b16 hands every lane
// 4 consecutive positions of ONE channel out of a [position][channel] image, so the LDS image of both operands is a plain
// copy of the packed rows (1 KB LDS-direct loads, no masks), and the stride / dilation of the convolution is just the
// per-lane address of the x read (row stride of the transposed block = s units, tap offset = j*d units).
//
//  * dy is packed with every item padded to Tq positions (a multiple of 16, zeros behind the n_out valid ones) and x with items
//    of exactly Tq*s positions (PL = pad zeros in front): then position f = b*Tq + to of dy pairs with unit f*s + j*d of x for
//    EVERY item -- the K loop runs over one flat index, K steps span short items freely, and the zero tail of dy cancels
//    whatever the x window holds there.
//  * Workgroup = 64 output channels x 64 input channels x TG taps (accumulators: TG tiles of 32x32 per wave, 2x2 waves);
//    the x window of a K step is staged once and serves all TG taps.  Long kernels split their taps over grid.x.
//  * Split-K over grid.z with partial tiles added in a fixed order by wgrad_pk_reduce_kernel (deterministic).
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "conv_pk_common.h"

namespace evmi {

struct WgradPkArgs {
  const uint4* dyp;   // [groups][octs_y][B*Tq] units (+ slack)
  const uint4* xp;    // [groups][octs_x][B*Tq*s] units (+ slack)
  float* out;         // dw [c_out][cin_g][k] (splits == 1) or partial tiles [splits][c_out][cin_g][k]
  int cout_g, cin_g, k, stride, dil;
  int octs_y, octs_x;
  long long plane_y, plane_x;  // units per octet row
  int ksteps;          // K steps of KS positions in total
  int steps_per_split;
  int tg, ntg;         // taps per workgroup, tap groups
  int tiles_ci, tiles_co;
  int xrow, xpieces;   // staged x row: units (64 * pieces)
  int nst;
  int accumulate;      // splits == 1 only: dw += instead of =
  int partial, c_out;  // 1: out is the partial-tile buffer [split][tap][c_out][cin_g]
  long long split_stride;  // floats between partial copies
  // TM instantiation (time-major bf16 tensors [row][C], rows = the flat (item, padded position) index, zero rows around every
  // item): the 16-byte unit (octet o, row f) is gathered from row f of the tensor by the LDS-direct load (one global address
  // per lane) -- no packed copy.  x rows are shifted by x_row_off = -pad (+ tap offsets); both tensors are readable (finite)
  // for a guard of rows in front of row 0 and behind the last row.
  const __bf16* dy_tm;
  const __bf16* x_tm;
  int cy_row, cx_row;      // elements per row (total channels) of dy / x
  long long x_row_off;
  // flat packed operands (the discriminator chains): the x unit of (position f, tap j) is f * s + j * d + x_unit_off
  long long x_unit_off;
  // block-diagonal mode for narrow groups: the workgroup's "group" is a super-group of several convolution groups (cout_g / cin_g
  // above are the super-group's widths); only the blocks (co / bd_cout == ci / bd_cin) are stored, as dw[co][ci % bd_cin][j]
  int bd_cout, bd_cin;
  // tiles whose input-channel extent is one 32-column MFMA tile (cin_g <= 32: the wn = 1 waves would idle): the two waves of a row share
  // the x rows and split the workgroup's TAPS instead (1); block-diagonal super-groups of two 32 x 32 groups: wave (wm, wn) takes the
  // diagonal block wm and half of the taps (2)
  int tap_split;
  // XCD-aware order of the workgroups (set by the launchers): the hardware hands consecutive workgroups to the eight XCDs in turn, so
  // in launch order the (tile_ci x tile_co) workgroups of ONE split -- which read the same K range of dy and x -- sit on eight different
  // L2s and every operand byte crosses the fabric 4-16 times (256 -> 1024 pointwise layer: 420 MB requested for 66 MB of operands).
  // Remapped, XCD c runs the contiguous range [c * n / 8, (c + 1) * n / 8) of the logical (x fastest, split slowest) order: the
  // workgroups resident on an XCD at one time are the tiles of one or a few splits and share their operands in that XCD's L2.
  // Which workgroup computes which tile changes, what a tile computes (K order, split order of the reduce) does not: same bits.
  int xcd;
};

constexpr int WG_KS = 64;  // positions per K step (4 MFMA K blocks)
// LDS row stride of the staged dy rows: 64 units + 4 (64 bytes), so that the two octet rows a 16-lane group of the transposing
// read touches (4 consecutive positions = 64 bytes each) fall on different banks (a stride of 1 KB puts them on the same ones:
// SQ_LDS_BANK_CONFLICT was 75 % of the LDS cycles); the staged x rows get the same 4 units of padding
constexpr int WG_YROW = WG_KS + 4;
// ... and of the staged x rows.  A 16-lane group of the transposing read touches two octet rows x four positions `stride` units apart:
// positions sit 4 * stride banks apart, so with strides 2 / 4 a row offset of 16 banks (4 units) lands the second row ON the first
// one's positions (SQ_LDS_BANK_CONFLICT 40.8 % of the LDS cycles of the flat weight gradients, profiles/r04v_train_pmc_summary.json);
// one unit (4 banks) interleaves them.  Strides 1 and 3 keep the 4 units (positions 4 / 12 banks apart: one unit would collide).
static inline int wg_xrow_pad(int stride) { return (stride == 2 || stride == 4) ? 1 : 4; }

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ s16x4 lds_read_tr(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
__device__ __forceinline__ bf16x8 join_tr(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

// WIDE: 128 output channels x 128 input channels per workgroup on EIGHT waves (2 x 4: a wave owns 64 x 32 x TG taps = 2 TG accumulator
// tiles) instead of 64 x 64 on four.  The kernel is bound by the rate its 1 KB LDS-direct pieces arrive (a 64 x 64 K step of the
// 1024 -> 1024, k = 5 layer stages 24 KB for 20 MFMAs per wave: 320 MB per launch at ~4.3 TB/s, MFMA-busy ~5 %); the wide tile stages
// 48 KB for four times the products -- half the bytes per product.  Same K order per accumulator: the sums differ from the narrow
// tile's only through the split count.  TGMAX <= 5 (160 accumulator registers), no tap split, no block-diagonal mode.
template <int TGMAX, bool TM = false, bool WIDE = false>
__global__ __launch_bounds__(WIDE ? 512 : 256, 2) void wgrad_pk_kernel(WgradPkArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint4 smem[];
  constexpr int NW = WIDE ? 8 : 4;       // waves
  constexpr int MB = WIDE ? 2 : 1;       // 32-channel output blocks per wave
  constexpr int RT = WIDE ? 16 : 8;      // octet rows of a tile side (128 / 64 channels)
  constexpr int TE = WIDE ? 128 : 64;    // channels of a tile side
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WIDE ? wave >> 2 : wave >> 1, wn = WIDE ? wave & 3 : wave & 1;
  const int kh = lane >> 5;
  const int k = a.k, s = a.stride, d = a.dil, tg = a.tg;

  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (a.xcd) {
    const unsigned gx = gridDim.x, gy = gridDim.y, n = gx * gy * gridDim.z;
    const unsigned lin = bx + gx * (by + gy * bz);
    const unsigned c = lin & 7, q = n >> 3, r = n & 7;           // XCD c runs q (+ 1 for c < r) workgroups
    const unsigned lg = c * q + min(c, r) + (lin >> 3);          // its slot-th one is logical id lg
    bx = lg % gx;
    by = (lg / gx) % gy;
    bz = lg / (gx * gy);
  }
  const int tile_ci = bx % a.tiles_ci, tgi = bx / a.tiles_ci;
  const int g = by / a.tiles_co, tile_co = by % a.tiles_co;
  const int j_lo = tgi * tg;
  const int tgw = min(tg, k - j_lo);  // taps of this group
  int j0 = 0, tgc = tgw;              // ... of this wave: [j0, j0 + tgc) of them
  if (!WIDE && a.tap_split) {
    const int half = (tgw + 1) >> 1;
    j0 = wn * half;
    tgc = max(0, min(half, tgw - j0));
  }
  const int xb = WIDE ? wn : (a.tap_split == 2 ? wm : (a.tap_split ? 0 : wn));  // 32-channel block of the staged x rows this wave multiplies
  const int split = bz;
  const int t_lo = split * a.steps_per_split, t_hi = min(a.ksteps, t_lo + a.steps_per_split);
  if (t_lo >= t_hi) return;  // (the reduce pass only reads the splits that exist)

  // octet rows of this tile, clamped to the group (rows past it are staged from the last one and never stored)
  const int oy0 = tile_co * RT, ox0 = tile_ci * RT;
  const uint4* dy_g = a.dyp + (long long)g * a.octs_y * a.plane_y;
  const uint4* x_g = a.xp + (long long)g * a.octs_x * a.plane_x;
  const int xrow = a.xrow, xpieces = a.xpieces;
  const int y_units = RT * WG_YROW;
  const int stage = y_units + RT * xrow;

  // ---- per-lane addresses of the transposing reads (bytes within a stage) ----
  // 16-lane group G: kh = G >> 1 (K half of the MFMA operand), hf = G & 1 (which 16 of the 32 channels); lane i of the group
  // supplies the address of 4 channels (i & 3) of position row (i >> 2) and receives channel i, 4 positions.
  const int i16 = lane & 15, hf = (lane >> 4) & 1;
  const int oct_in_blk = 2 * hf + ((i16 & 3) >> 1);
  const int a_base = (((wm * MB * 4 + oct_in_blk) * WG_YROW) + 8 * kh + (i16 >> 2)) * 16 + (i16 & 1) * 8;  // (+ mb * 4 rows per further block)
  const int b_base = (y_units + (xb * 4 + oct_in_blk) * xrow + (8 * kh + (i16 >> 2)) * s + j0 * d) * 16 + (i16 & 1) * 8;

  f32x16 acc[MB][TGMAX];
#pragma unroll
  for (int mb = 0; mb < MB; ++mb)
#pragma unroll
    for (int j = 0; j < TGMAX; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mb][j][r] = 0.f;

  auto issue = [&](int t, int slot) -> int {
    uint4* sy = smem + slot * stage;
    uint4* sx = sy + y_units;
    const long long f0 = (long long)t * WG_KS;
    int issued = 0;
    // (octet rows past the group are NOT staged: what the transposing reads find there multiplies into accumulator rows / columns
    // that are never stored -- staging the last valid row again in their place was half of the loads of the 32-channel groups)
    const int ny = min(RT, a.octs_y - oy0), nx = min(RT, a.octs_x - ox0);
    int u = wave;
    for (; u < ny; u += NW) {  // dy: up to RT octet rows x 64 units = one 1 KB piece each
      const int o = oy0 + u;
      if (TM) pk_lds_direct(reinterpret_cast<const uint4*>(a.dy_tm + (f0 + lane) * a.cy_row + (g * a.cout_g + o * 8)), sy + u * WG_YROW);
      else pk_lds_direct(dy_g + (long long)o * a.plane_y + f0 + lane, sy + u * WG_YROW);
      ++issued;
    }
    u = wave;
    const long long x0 = f0 * s + (long long)j_lo * d + (TM ? 0 : a.x_unit_off);
    for (; u < nx * xpieces; u += NW) {
      const int r = u / xpieces, pi = u - r * xpieces;
      const int o = ox0 + r;
      if (TM) pk_lds_direct(reinterpret_cast<const uint4*>(a.x_tm + (x0 + a.x_row_off + pi * 64 + lane) * a.cx_row + (g * a.cin_g + o * 8)),
                            sx + r * xrow + pi * 64);
      else pk_lds_direct(x_g + (long long)o * a.plane_x + x0 + pi * 64 + lane, sx + r * xrow + pi * 64);
      ++issued;
    }
    return issued;
  };

  const int nst = a.nst;
  int n_next = 0;
  issue(t_lo, 0);
  if (nst == 3 && t_lo + 1 < t_hi) n_next = issue(t_lo + 1, 1);
  int slot = -1;
  for (int t = t_lo; t < t_hi; ++t) {
    slot = slot + 1 == nst ? 0 : slot + 1;
    const int slot_ahead = slot == 0 ? nst - 1 : slot - 1;
    wait_vmcnt_le(n_next);
    lds_barrier();
    {
      const int issued = t + nst - 1 < t_hi ? issue(t + nst - 1, slot_ahead) : 0;
      n_next = nst == 3 ? issued : 0;
    }
    const char* sm = reinterpret_cast<const char*>(smem + slot * stage);
#pragma unroll
    for (int kb = 0; kb < WG_KS / 16; ++kb) {
      const char* pa = sm + a_base + kb * 256;
      bf16x8 fa[MB];
#pragma unroll
      for (int mb = 0; mb < MB; ++mb) {
        const char* pm = pa + mb * (4 * WG_YROW * 16);
        fa[mb] = join_tr(lds_read_tr(pm), lds_read_tr(pm + 64));
      }
      const char* pb = sm + b_base + kb * 16 * s * 16;
      bf16x8 fb[TGMAX];
#pragma unroll
      for (int j = 0; j < TGMAX; ++j)
        if (j < tgc) {
          const char* q = pb + j * d * 16;
          fb[j] = join_tr(lds_read_tr(q), lds_read_tr(q + 4 * s * 16));
        }
#pragma unroll
      for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int j = 0; j < TGMAX; ++j)
          if (j < tgc) acc[mb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mb], fb[j], acc[mb][j], 0, 0, 0);
    }
  }

  // ---- store: lane column = input channel, registers = output channels ----
  const int ci_sg = tile_ci * TE + xb * 32 + (lane & 31);
  if (ci_sg >= a.cin_g) return;
  const int cig = a.bd_cin ? ci_sg / a.bd_cin : 0;           // convolution group inside the super-group (block-diagonal mode)
  const int ci = a.bd_cin ? ci_sg - cig * a.bd_cin : ci_sg;  // input channel inside its convolution group
  const int cin_st = a.bd_cin ? a.bd_cin : a.cin_g;
#pragma unroll
  for (int mb = 0; mb < MB; ++mb) {
    const int m_blk = tile_co * TE + (wm * MB + mb) * 32;  // first output channel of this accumulator block
    if (a.partial) {  // partial tiles [split][tap][c_out][cin_g]: consecutive lanes consecutive addresses
      float* outp = a.out + (long long)split * a.split_stride;
      const long long tap_stride = (long long)a.c_out * cin_st;
#pragma unroll
      for (int j = 0; j < TGMAX; ++j) {
        if (j >= tgc) break;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m_blk + (r & 3) + 8 * (r >> 2) + 4 * kh;
          if (m >= a.cout_g || (a.bd_cout && m / a.bd_cout != cig)) continue;
          outp[(j_lo + j0 + j) * tap_stride + (long long)(g * a.cout_g + m) * cin_st + ci] = acc[mb][j][r];
        }
      }
      continue;
    }
#pragma unroll
    for (int j = 0; j < TGMAX; ++j) {
      if (j >= tgc) break;
      // the previous values of the 16 rows are requested together, from clamped rows (a read per element under the `accumulate`
      // condition is one drained round trip each); -0 is the neutral start of a plain store
      long long off[16];
      float prev[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = min(m_blk + (r & 3) + 8 * (r >> 2) + 4 * kh, a.cout_g - 1);
        off[r] = ((long long)(g * a.cout_g + m) * cin_st + ci) * k + j_lo + j0 + j;
        prev[r] = -0.f;
      }
      if (a.accumulate) {
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[r] = a.out[off[r]];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m_blk + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (m < a.cout_g && !(a.bd_cout && m / a.bd_cout != cig)) a.out[off[r]] = prev[r] + acc[mb][j][r];
      }
    }
  }
}

// dw[co][ci][j] (+)= sum over the partial copies [split][j][co][ci], in split order (fixed summation order: bitwise
// reproducible).  One workgroup per 256 consecutive (co, ci) pairs: the k tap planes are read coalesced (summed over the splits)
// into LDS and written back as the 256 * k contiguous floats of dw they are.
__global__ __launch_bounds__(256) void wgrad_pk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long long rows_ci, int k,
                                                              int splits, long long split_stride, int accumulate) {
  extern __shared__ float tile[];  // [k][256]
  const long long base = (long long)blockIdx.x * 256;
  const long long i = base + threadIdx.x;
  if (i < rows_ci)
    for (int j = 0; j < k; ++j) {
      tile[j * 256 + threadIdx.x] = ordered_sum_strided(part + j * rows_ci + i, split_stride, splits);
    }
  __syncthreads();
  const int n_here = (int)min<long long>(256, rows_ci - base);
  float* dst = dw + base * k;
  for (int e = threadIdx.x; e < n_here * k; e += 256) {
    const int p = e / k, j = e - p * k;
    const float v = tile[j * 256 + p];
    dst[e] = accumulate ? dst[e] + v : v;
  }
}

// the same for small weight tensors reduced over many splits: one thread per (j, co, ci) (k times the workgroups; strided write)
__global__ __launch_bounds__(256) void wgrad_pk_reduce_planes_kernel(const float* __restrict__ part, float* __restrict__ dw, long long rows_ci,
                                                                     int k, int splits, long long split_stride, int accumulate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;  // co * cin_g + ci
  const int j = blockIdx.y;
  if (i >= rows_ci) return;
  const float acc = ordered_sum_strided(part + j * rows_ci + i, split_stride, splits);
  float* dst = dw + i * k + j;
  *dst = accumulate ? *dst + acc : acc;
}

struct WgradPkPlan {
  int Tq, octs_y, octs_x, splits;
  long long dy_units, x_units, part_elems;
  size_t lds;
  dim3 grid;
  int tgmax;
  int wide;  // 1: the 128 x 128 tile on eight waves (wgrad_pk_kernel<5, TM, true>)
};

static int wg_env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}

// The 128 x 128 tile (wgrad_pk_kernel<.., true>) where both channel extents reach it, no wave splits taps or diagonal blocks, and the
// shape still gives the chip a workgroup per CU: fills the tile geometry of `a` / `pl` (taps per workgroup <= 5) and returns true.
// EVMI_WG_WIDE=0: never (A/B).
static bool plan_wgrad_wide(WgradPkArgs& a, WgradPkPlan& pl, int k, int stride, int dil, int xrow_pad, long long n_groups) {
  static const int on = wg_env_int("EVMI_WG_WIDE", 1);
  pl.wide = 0;
  if (!on || a.tap_split || a.bd_cin || a.cin_g < 128 || a.cout_g < 128) return false;
  const int tgcap = 5;
  const int ntg = (k + tgcap - 1) / tgcap, tg = (k + ntg - 1) / ntg;
  const long long xwin = (long long)(WG_KS - 1) * stride + (long long)(tg - 1) * dil + 1;
  if (xwin > 64 * 12) return false;
  const int xpieces = (int)((xwin + 63) / 64), xrow = xpieces * 64 + xrow_pad;
  const size_t stage_bytes = (size_t)(16 * WG_YROW + 16 * xrow) * 16;
  if (2 * stage_bytes > 160 * 1024) return false;
  const int tiles_ci = (a.cin_g + 127) / 128, tiles_co = (a.cout_g + 127) / 128;
  const long long tiles = (long long)tiles_ci * ntg * tiles_co * n_groups;
  static const long long want = wg_env_int("EVMI_WG_WIDE_WANT", 256);  // one eight-wave workgroup per CU
  int splits = (int)std::min<long long>(std::max<long long>(1, (want + tiles - 1) / tiles), std::max(1, a.ksteps / 8));
  const int steps = (a.ksteps + splits - 1) / splits;
  splits = (a.ksteps + steps - 1) / steps;
  if (tiles * splits < 192) return false;  // too few workgroups for the chip: the 64 x 64 tile (four times the tiles) fills it
  a.ntg = ntg; a.tg = tg;
  a.tiles_ci = tiles_ci; a.tiles_co = tiles_co;
  a.xpieces = xpieces; a.xrow = xrow;
  a.nst = 3 * stage_bytes <= 160 * 1024 ? 3 : 2;
  a.steps_per_split = steps;
  pl.lds = a.nst * stage_bytes;
  pl.tgmax = 5;
  pl.splits = splits;
  pl.wide = 1;
  return true;
}

static const char* plan_wgrad_pk(WgradPkArgs& a, WgradPkPlan& pl, int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad,
                                 int dil, int groups) {
  if (groups <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups || k <= 0 || stride <= 0 || dil <= 0 || B <= 0 || n_out <= 0 ||
      t_in <= 0 || pad < 0)
    return "bad shape";
  const int cin_g = c_in / groups, cout_g = c_out / groups;
  // half-empty 64 x 64 tiles pay most with many taps sharing the staged window (128->128 k41 g4 0.38 -> 0.26 ms); the 32 x 32
  // layers of the generator's last stage (k 3 / 7 / 11: 0.09 ms here) still beat the fp32 unfold + GEMM route (0.15 ms)
  static const int min_prod = 1024;
  if (cin_g < 32 || cout_g < 32 || (cin_g * cout_g < min_prod && k < 16))
    return "narrow groups (the fp32 implicit-GEMM kernel takes them)";
  if (stride > 8) return "stride above 8";
  a.cout_g = cout_g; a.cin_g = cin_g; a.k = k; a.stride = stride; a.dil = dil;
  pl.octs_y = a.octs_y = (cout_g + 7) / 8;
  pl.octs_x = a.octs_x = (cin_g + 7) / 8;
  // item length: a multiple of 16 positions, and Tq*s units of x must hold the padded item and every tap of a valid output
  long long need = std::max<long long>((long long)pad + t_in, (long long)(n_out - 1) * stride + (long long)(k - 1) * dil + 1);
  long long Tq = ((long long)n_out + 15) / 16 * 16;
  // ... and the rows of the packed dy must end on a K-step boundary (the step past the end would read the next row)
  while (Tq * stride < need || ((long long)B * Tq) % WG_KS) Tq += 16;
  // pointwise stride-1 layers whose rows end on a K step anyway: tight items, the layout the convolution kernels pack (conv_pk_common.h)
  static_assert(WG_KS == 64, "pk_shared_items states the K step");
  if (pk_shared_shape(k, stride, pad, dil, groups) && pk_shared_items(B, n_out) && t_in == n_out) Tq = pk_shared_pitch(B, n_out);
  if (Tq > (1 << 22)) return "row too long";
  pl.Tq = (int)Tq;
  a.plane_y = (long long)B * Tq;
  a.plane_x = (long long)B * Tq * stride;
  a.ksteps = (int)((a.plane_y + WG_KS - 1) / WG_KS);
  a.tap_split = 0; a.bd_cin = a.bd_cout = 0;
  int splits;
  if (plan_wgrad_wide(a, pl, k, stride, dil, wg_xrow_pad(stride), groups)) {
    splits = pl.splits;
  } else {
    // taps per workgroup
    const int tgcap = 8;
    a.ntg = (k + tgcap - 1) / tgcap;
    a.tg = (k + a.ntg - 1) / a.ntg;
    pl.tgmax = a.tg <= 4 ? 4 : 8;
    a.tiles_ci = (cin_g + 63) / 64;
    a.tiles_co = (cout_g + 63) / 64;
    const long long xwin = (long long)(WG_KS - 1) * stride + (long long)(a.tg - 1) * dil + 1;
    if (xwin > 64 * 12) return "input window too long";
    a.xpieces = (int)((xwin + 63) / 64);
    a.xrow = a.xpieces * 64 + wg_xrow_pad(stride);
    const size_t stage_bytes = (size_t)(8 * WG_YROW + 8 * a.xrow) * 16;
    a.nst = 3 * stage_bytes <= 78 * 1024 ? 3 : 2;
    const int fn = wg_env_int("EVMI_WG_NST", 0);
    if (fn == 2 || fn == 3) a.nst = fn;
    pl.lds = a.nst * stage_bytes;
    if (pl.lds > 160 * 1024) return "LDS budget";
    const long long tiles = (long long)a.tiles_ci * a.ntg * a.tiles_co * groups;
    static const long long want = wg_env_int("EVMI_WG_WANT", 512);  // (A/B: workgroups a weight gradient is split up to)
    static const int min_steps = 8;  // K steps per workgroup that pay for its prologue and tile store
    splits = (int)std::min<long long>(std::max<long long>(1, (want + tiles - 1) / tiles), std::max(1, a.ksteps / min_steps));
    const int fs = wg_env_int("EVMI_WG_SPLITS", 0);
    if (fs > 0) splits = std::min(fs, a.ksteps);
    a.steps_per_split = (a.ksteps + splits - 1) / splits;
    splits = (a.ksteps + a.steps_per_split - 1) / a.steps_per_split;  // no empty splits
    pl.splits = splits;
  }
  if ((long long)a.tiles_ci * a.ntg > 0x7fffffffLL || (long long)groups * a.tiles_co > 65535 || splits > 65535) return "grid limits";
  pl.grid = dim3(a.tiles_ci * a.ntg, groups * a.tiles_co, splits);
  // slack: the last K step reads up to its full window past the end of the last row (the wide tile's 16 octet rows: rows past the
  // group are not staged, see the kernel)
  pl.dy_units = (long long)groups * a.octs_y * a.plane_y + WG_KS + 64;
  pl.x_units = (long long)groups * a.octs_x * a.plane_x + (long long)WG_KS * stride + a.xrow + 64;
  a.split_stride = (long long)c_out * cin_g * k;
  pl.part_elems = splits > 1 ? a.split_stride * splits : 0;

  if (pl.dy_units >= (1LL << 31) || pl.x_units >= (1LL << 31)) return "packed operands too large";
  return nullptr;
}

// the packed-operand instantiations (one attribute cache for every caller: the attribute belongs to the kernel, not to the call site)
static int wg_xcd_order() {
  static const int on = wg_env_int("EVMI_WG_XCD", 1);
  return on;
}

static int launch_wgrad_packed(const WgradPkArgs& a_in, const WgradPkPlan& pl, hipStream_t s) {
  WgradPkArgs a = a_in;
  a.xcd = wg_xcd_order();
  static thread_local size_t configured_dev[kMaxDevices][2] = {};
  size_t* configured = configured_dev[device_slot()];
  const size_t lds = pl.lds;
  if (pl.wide) {
    static thread_local size_t wide_lds[kMaxDevices] = {};
    size_t& cfg = wide_lds[device_slot()];
    if (lds > cfg) {
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_pk_kernel<5, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      cfg = lds;
    }
    hipLaunchKernelGGL((wgrad_pk_kernel<5, false, true>), pl.grid, dim3(512), lds, s, a);
  } else if (pl.tgmax == 4) {
    if (lds > configured[0]) {
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_pk_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured[0] = lds;
    }
    hipLaunchKernelGGL((wgrad_pk_kernel<4>), pl.grid, dim3(256), lds, s, a);
  } else {
    if (lds > configured[1]) {
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_pk_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured[1] = lds;
    }
    hipLaunchKernelGGL((wgrad_pk_kernel<8>), pl.grid, dim3(256), lds, s, a);
  }
  EVMI_LAUNCH_CHECK("wgrad_pk_kernel");
  return EVMI_OK;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

/* Floats of workspace (packed dy, packed x, split-K partial tiles); 0 = shape not taken here. */
long long evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil,
                                                int groups) {
  WgradPkArgs a = {};
  WgradPkPlan pl;
  if (plan_wgrad_pk(a, pl, B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups)) return 0;
  return (pl.dy_units + pl.x_units) * 4 + pl.part_elems;
}

int evmi_conv1d_wgrad_cbt_bf16pk_plan(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil,
                                      int groups) {
  WgradPkArgs a = {};
  WgradPkPlan pl;
  if (plan_wgrad_pk(a, pl, B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups)) return -1;
  return pl.tgmax + 16 * (pl.splits > 1 ? pl.splits : 0);
}

static int wgrad_pk_impl(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev, long long ws_elems, int B, int c_in, int t_in,
                         int c_out, int n_out, int k, int stride, int pad, int dil, int groups, int accumulate, float x_pre_slope,
                         const float* dy_mask_dev, float dy_mask_slope, void* stream, const void* x_packed_dev = nullptr,
                         const void* dy_packed_dev = nullptr);

int evmi_conv1d_wgrad_cbt_bf16pk(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev, long long ws_elems, int B, int c_in,
                                 int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups, int accumulate,
                                 void* stream) {
  return wgrad_pk_impl(x_dev, dy_dev, dw_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups, accumulate, 1.f,
                       nullptr, 1.f, stream);
}

/* Pointwise stride-1 layers (k = 1, no padding, one group): operands that are ALREADY packed -- x_packed_dev: the head of the workspace
 * the forward evmi_conv1d_cbt_bf16pk* call of this layer was given (its packed input), dy_packed_dev: the head of the workspace of the
 * layer's input-gradient call (its packed dy); either may be NULL (then the fp32 tensor is packed here as usual); B * t must be a
 * multiple of 64 (evmi_conv1d_bf16pk_shares_packed).  The caller keeps
 * those workspaces untouched until this call has run. */
int evmi_conv1d_wgrad_cbt_bf16pk_prepacked(const float* x_dev, const void* x_packed_dev, const float* dy_dev, const void* dy_packed_dev,
                                           float* dw_dev, float* ws_dev, long long ws_elems, int B, int c_in, int t_in, int c_out, int n_out,
                                           int k, int stride, int pad, int dil, int groups, int accumulate, void* stream) {
  return wgrad_pk_impl(x_dev, dy_dev, dw_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups, accumulate, 1.f,
                       nullptr, 1.f, stream, x_packed_dev, dy_packed_dev);
}
/* 1 when a layer of this shape packs its operands in the format the call above reads (evmi_conv1d_cbt_bf16pk / _dgrad_ and the weight
 * gradient agree on it), else 0. */
int evmi_conv1d_bf16pk_shares_packed(int B, int t, int k, int stride, int pad, int dil, int groups) {
  return pk_shared_shape(k, stride, pad, dil, groups) && pk_shared_items(B, t) ? 1 : 0;
}

/* The same with the operands transformed while they are packed: x -> leaky_relu(x, x_pre_slope) (the activation in front of the
 * convolution, never materialised) and dy -> dy * (dy_mask > 0 ? 1 : dy_mask_slope) (the backward of the leaky ReLU behind it). */
int evmi_conv1d_wgrad_cbt_bf16pk_fused(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev, long long ws_elems, int B,
                                       int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups,
                                       int accumulate, float x_pre_slope, const float* dy_mask_dev, float dy_mask_slope, void* stream) {
  return wgrad_pk_impl(x_dev, dy_dev, dw_dev, ws_dev, ws_elems, B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups, accumulate,
                       x_pre_slope, dy_mask_dev, dy_mask_slope, stream);
}

// ---- time-major operands (train_tm.hip's residual stacks): no pack, the kernel gathers its units from [row][C] tensors ----------
static const char* plan_wgrad_tm(WgradPkArgs& a, WgradPkPlan& pl, long long rows, int c_in, int c_out, int k, int dil) {
  if (c_in < 32 || c_out < 32 || (c_in % 8) || (c_out % 8) || k <= 0 || dil <= 0 || rows <= 0) return "bad shape (channels: multiples of 8, at least 32)";
  a.cout_g = c_out; a.cin_g = c_in; a.k = k; a.stride = 1; a.dil = dil;
  pl.octs_y = a.octs_y = c_out / 8;
  pl.octs_x = a.octs_x = c_in / 8;
  a.ksteps = (int)((rows + WG_KS - 1) / WG_KS);
  int splits;
  // (the 128 x 128 tile loses on the generator's time-major layers: c128 / k11 54 -> 58 us, c256 unchanged -- tools/bench_wgrad_bf16.py)
  pl.wide = 0;
  if (false) {
  } else {
    const int tgcap = 8;
    a.ntg = (k + tgcap - 1) / tgcap;
    a.tg = (k + a.ntg - 1) / a.ntg;
    pl.tgmax = a.tg <= 4 ? 4 : 8;
    a.tiles_ci = (c_in + 63) / 64;
    a.tiles_co = (c_out + 63) / 64;
    const long long xwin = (long long)(WG_KS - 1) + (long long)(a.tg - 1) * dil + 1;
    if (xwin > 64 * 12) return "input window too long";
    a.xpieces = (int)((xwin + 63) / 64);
    a.xrow = a.xpieces * 64 + 4;
    const size_t stage_bytes = (size_t)(8 * WG_YROW + 8 * a.xrow) * 16;
    a.nst = 3 * stage_bytes <= 78 * 1024 ? 3 : 2;
    pl.lds = a.nst * stage_bytes;
    if (pl.lds > 160 * 1024) return "LDS budget";
    const long long tiles = (long long)a.tiles_ci * a.ntg * a.tiles_co;
    const long long want = 512;
    splits = (int)std::min<long long>(std::max<long long>(1, (want + tiles - 1) / tiles), std::max(1, a.ksteps / 8));
    a.steps_per_split = (a.ksteps + splits - 1) / splits;
    splits = (a.ksteps + a.steps_per_split - 1) / a.steps_per_split;
    pl.splits = splits;
  }
  if (splits > 65535) return "grid limits";
  pl.grid = dim3(a.tiles_ci * a.ntg, a.tiles_co, splits);
  a.split_stride = (long long)c_out * c_in * k;
  pl.part_elems = splits > 1 ? a.split_stride * splits : 0;
  return nullptr;
}

long long evmi_conv1d_wgrad_tm_bf16_ws_elems(long long rows, int c_in, int c_out, int k, int dil) {
  WgradPkArgs a = {};
  WgradPkPlan pl;
  if (plan_wgrad_tm(a, pl, rows, c_in, c_out, k, dil)) return -1;
  return std::max<long long>(pl.part_elems, 4);
}

int evmi_conv1d_wgrad_tm_bf16(const void* x_tm, const void* dy_tm, float* dw_dev, float* ws_dev, long long ws_elems, long long rows, int c_in,
                              int c_out, int k, int pad, int dil, int accumulate, void* stream) {
  if (!x_tm || !dy_tm || !dw_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_tm_bf16: null pointer");
  if ((reinterpret_cast<uintptr_t>(x_tm) | reinterpret_cast<uintptr_t>(dy_tm)) & 15) return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_tm_bf16: operands must be 16-byte aligned");
  WgradPkArgs a = {};
  WgradPkPlan pl;
  if (const char* why = plan_wgrad_tm(a, pl, rows, c_in, c_out, k, dil)) return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_wgrad_tm_bf16: ") + why);
  if (pl.part_elems > 0 && (!ws_dev || ws_elems < pl.part_elems)) return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_tm_bf16: workspace missing or too small");
  hipStream_t s = (hipStream_t)stream;
  a.dy_tm = reinterpret_cast<const __bf16*>(dy_tm);
  a.x_tm = reinterpret_cast<const __bf16*>(x_tm);
  a.cy_row = c_out; a.cx_row = c_in; a.x_row_off = -(long long)pad;
  a.out = pl.splits > 1 ? ws_dev : dw_dev;
  a.accumulate = pl.splits > 1 ? 0 : accumulate;
  a.partial = pl.splits > 1;
  a.c_out = c_out;
  a.xcd = wg_xcd_order();
  static thread_local size_t configured_dev[kMaxDevices][2] = {};
  size_t* configured = configured_dev[device_slot()];
  const size_t lds = pl.lds;
  if (pl.wide) {
    static thread_local size_t wide_lds[kMaxDevices] = {};
    size_t& cfg = wide_lds[device_slot()];
    if (lds > cfg) {
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_pk_kernel<5, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      cfg = lds;
    }
    hipLaunchKernelGGL((wgrad_pk_kernel<5, true, true>), pl.grid, dim3(512), lds, s, a);
  } else if (pl.tgmax == 4) {
    if (lds > configured[0]) {
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_pk_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured[0] = lds;
    }
    hipLaunchKernelGGL((wgrad_pk_kernel<4, true>), pl.grid, dim3(256), lds, s, a);
  } else {
    if (lds > configured[1]) {
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)wgrad_pk_kernel<8, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      configured[1] = lds;
    }
    hipLaunchKernelGGL((wgrad_pk_kernel<8, true>), pl.grid, dim3(256), lds, s, a);
  }
  EVMI_LAUNCH_CHECK("wgrad_tm_kernel");
  if (pl.splits > 1) {
    const long long rows_ci = (long long)c_out * c_in;
    if (rows_ci >= 131072)
      hipLaunchKernelGGL(wgrad_pk_reduce_kernel, dim3((unsigned)((rows_ci + 255) / 256)), dim3(256), (size_t)k * 256 * sizeof(float), s, ws_dev,
                         dw_dev, rows_ci, k, pl.splits, a.split_stride, accumulate);
    else
      hipLaunchKernelGGL(wgrad_pk_reduce_planes_kernel, dim3((unsigned)((rows_ci + 255) / 256), k), dim3(256), 0, s, ws_dev, dw_dev, rows_ci, k,
                         pl.splits, a.split_stride, accumulate);
    EVMI_LAUNCH_CHECK("wgrad_tm_reduce");
  }
  return EVMI_OK;
}

// ---- flat packed operands (the discriminator chains): dy one long row of n_items * T_dy units per channel octet, x one of
// n_items * T_dy * stride units; unit f of dy pairs with unit f * stride + j * dil - pad of x.  No pack, no per-item padding rule:
// the zero gaps of dy cancel whatever x holds beside them.  Groups narrower than 32 channels run block-diagonally. ------------
static const char* plan_wgrad_flat(WgradPkArgs& a, WgradPkPlan& pl, int& sgroups, long long n_pos, int c_in, int c_out, int k, int stride, int dil,
                                   int groups) {
  if (groups <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups || k <= 0 || stride <= 0 || dil <= 0 || n_pos <= 0) return "bad shape";
  int cin_g = c_in / groups, cout_g = c_out / groups;
  if (cin_g % 8 || cout_g % 8) return "channels per group must be multiples of 8";
  if (stride > 8) return "stride above 8";
  a.bd_cout = a.bd_cin = 0;
  sgroups = groups;
  if (groups > 1 && (cin_g < 64 && cout_g < 64)) {  // super-groups of m convolution groups: 64 output channels where the group count allows
    int m = std::max(1, 64 / cout_g);
    while (m > 1 && groups % m) m >>= 1;
    if (m > 1) {
      a.bd_cout = cout_g; a.bd_cin = cin_g;
      cin_g *= m; cout_g *= m; sgroups = groups / m;
    }
  }
  a.cout_g = cout_g; a.cin_g = cin_g; a.k = k; a.stride = stride; a.dil = dil;
  pl.octs_y = a.octs_y = cout_g / 8;
  pl.octs_x = a.octs_x = cin_g / 8;
  a.ksteps = (int)((n_pos + WG_KS - 1) / WG_KS);
  // one 32-column tile of input channels (or two 32 x 32 groups on the diagonal): the waves of a row split the taps -- up to 16 per
  // workgroup, 8 per wave, half the workgroups staging the same windows (EVMI_WG_TAPSPLIT=0: off)
  static const int tap_split_on = wg_env_int("EVMI_WG_TAPSPLIT", 1);
  a.tap_split = !tap_split_on ? 0 : (cin_g <= 32 ? 1 : (a.bd_cin == 32 && a.bd_cout == 32 && cin_g == 64 && cout_g == 64 ? 2 : 0));
  int splits;
  if (plan_wgrad_wide(a, pl, k, stride, dil, wg_xrow_pad(stride), sgroups)) {
    splits = pl.splits;
  } else {
    const int tgcap = a.tap_split ? 16 : 8;
    a.ntg = (k + tgcap - 1) / tgcap;
    a.tg = (k + a.ntg - 1) / a.ntg;
    const int tg_wave = a.tap_split ? (a.tg + 1) / 2 : a.tg;
    pl.tgmax = tg_wave <= 4 ? 4 : 8;
    a.tiles_ci = (cin_g + 63) / 64;
    a.tiles_co = (cout_g + 63) / 64;
    const long long xwin = (long long)(WG_KS - 1) * stride + (long long)(a.tg - 1) * dil + 1;
    if (xwin > 64 * 12) return "input window too long";
    a.xpieces = (int)((xwin + 63) / 64);
    a.xrow = a.xpieces * 64 + wg_xrow_pad(stride);
    const size_t stage_bytes = (size_t)(8 * WG_YROW + 8 * a.xrow) * 16;
    a.nst = 3 * stage_bytes <= 78 * 1024 ? 3 : 2;
    pl.lds = a.nst * stage_bytes;
    if (pl.lds > 160 * 1024) return "LDS budget";
    const long long tiles = (long long)a.tiles_ci * a.ntg * a.tiles_co * sgroups;
    const long long want = 512;
    splits = (int)std::min<long long>(std::max<long long>(1, (want + tiles - 1) / tiles), std::max(1, a.ksteps / 8));
    a.steps_per_split = (a.ksteps + splits - 1) / splits;
    splits = (a.ksteps + a.steps_per_split - 1) / a.steps_per_split;
    pl.splits = splits;
  }
  if ((long long)a.tiles_ci * a.ntg > 0x7fffffffLL || (long long)sgroups * a.tiles_co > 65535 || splits > 65535) return "grid limits";
  pl.grid = dim3(a.tiles_ci * a.ntg, sgroups * a.tiles_co, splits);
  a.split_stride = (long long)c_out * (c_in / groups) * k;
  pl.part_elems = splits > 1 ? a.split_stride * splits : 0;
  return nullptr;
}

long long evmi_conv_pkflat_wgrad_ws_elems(int n_items, int T_dy, int c_in, int c_out, int k, int stride, int dil, int groups) {
  WgradPkArgs a = {};
  WgradPkPlan pl;
  int sg;
  if (plan_wgrad_flat(a, pl, sg, (long long)n_items * T_dy, c_in, c_out, k, stride, dil, groups)) return -1;
  return std::max<long long>(pl.part_elems, 4);
}

int evmi_conv_pkflat_wgrad(const void* x_pk, long long x_plane, const void* dy_pk, long long dy_plane, float* dw_dev, float* ws_dev,
                           long long ws_elems, int n_items, int T_dy, int c_in, int c_out, int k, int stride, int pad, int dil, int groups,
                           int accumulate, void* stream) {
  if (!x_pk || !dy_pk || !dw_dev) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_wgrad: null pointer");
  WgradPkArgs a = {};
  WgradPkPlan pl;
  int sgroups;
  if (const char* why = plan_wgrad_flat(a, pl, sgroups, (long long)n_items * T_dy, c_in, c_out, k, stride, dil, groups))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv_pkflat_wgrad: ") + why);
  if (pl.part_elems > 0 && (!ws_dev || ws_elems < pl.part_elems)) return fail(EVMI_ERR_INVALID_ARG, "conv_pkflat_wgrad: workspace missing or too small");
  hipStream_t s = (hipStream_t)stream;
  a.dyp = reinterpret_cast<const uint4*>(dy_pk);
  a.xp = reinterpret_cast<const uint4*>(x_pk);
  a.plane_y = dy_plane; a.plane_x = x_plane;
  a.x_unit_off = -(long long)pad;
  a.out = pl.splits > 1 ? ws_dev : dw_dev;
  a.accumulate = pl.splits > 1 ? 0 : accumulate;
  a.partial = pl.splits > 1;
  a.c_out = c_out;
  if (int rc = launch_wgrad_packed(a, pl, s)) return rc;
  if (pl.splits > 1) {
    const long long rows_ci = (long long)c_out * (c_in / groups);
    if (rows_ci >= 131072)
      hipLaunchKernelGGL(wgrad_pk_reduce_kernel, dim3((unsigned)((rows_ci + 255) / 256)), dim3(256), (size_t)k * 256 * sizeof(float), s, ws_dev,
                         dw_dev, rows_ci, k, pl.splits, a.split_stride, accumulate);
    else
      hipLaunchKernelGGL(wgrad_pk_reduce_planes_kernel, dim3((unsigned)((rows_ci + 255) / 256), k), dim3(256), 0, s, ws_dev, dw_dev, rows_ci, k,
                         pl.splits, a.split_stride, accumulate);
    EVMI_LAUNCH_CHECK("wgrad_pk_reduce (flat)");
  }
  return EVMI_OK;
}

static int wgrad_pk_impl(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev, long long ws_elems, int B, int c_in, int t_in,
                         int c_out, int n_out, int k, int stride, int pad, int dil, int groups, int accumulate, float x_pre_slope,
                         const float* dy_mask_dev, float dy_mask_slope, void* stream, const void* x_packed_dev,
                         const void* dy_packed_dev) {
  if ((!x_dev && !x_packed_dev) || (!dy_dev && !dy_packed_dev) || !dw_dev || !ws_dev)
    return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_cbt_bf16pk: null pointer");
  if ((x_packed_dev || dy_packed_dev) && !pk_shared_shape(k, stride, pad, dil, groups))
    return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_cbt_bf16pk: pre-packed operands exist for pointwise stride-1 layers only");
  WgradPkArgs a = {};
  WgradPkPlan pl;
  if (const char* why = plan_wgrad_pk(a, pl, B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_wgrad_cbt_bf16pk: ") + why);
  const long long need = (pl.dy_units + pl.x_units) * 4 + pl.part_elems;
  if (ws_elems < need || (reinterpret_cast<uintptr_t>(ws_dev) & 15))
    return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_cbt_bf16pk: workspace too small or unaligned");
  hipStream_t s = (hipStream_t)stream;
  uint4* dyp = reinterpret_cast<uint4*>(ws_dev);
  uint4* xp = dyp + pl.dy_units;
  float* part = reinterpret_cast<float*>(xp + pl.x_units);
  const int cin_g = c_in / groups, cout_g = c_out / groups;
  // both operands packed in one launch (the slack behind the packed tensors is read, never used: the pack zeroes it)
  const int Tpx = pl.Tq * stride;
  PackArgs py = make_pack_args(dy_dev, dyp, cout_g, a.octs_y, B, n_out, pl.Tq, 0, (int)(pl.dy_units - (long long)groups * a.octs_y * a.plane_y),
                               groups);
  PackArgs px = make_pack_args(x_dev, xp, cin_g, a.octs_x, B, t_in, Tpx, pad, (int)(pl.x_units - (long long)groups * a.octs_x * a.plane_x),
                               groups);
  px.pre_slope = x_pre_slope;
  py.mask = dy_mask_dev; py.mask_slope = dy_mask_slope;
  // an operand that arrives packed (the forward convolution's input / the input-gradient convolution's dy, in the shared item
  // layout: plan_wgrad_pk's Tq is the tight t_in for these shapes) is read where it lies; its pack is an empty grid
  if ((x_packed_dev || dy_packed_dev) && !(pk_shared_items(B, n_out) && pl.Tq == pk_shared_pitch(B, n_out) && t_in == n_out))
    return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_cbt_bf16pk: packed operands need rows that end on a K step (B * t a multiple of 64)");
  if (dy_packed_dev) py.gx = py.gy = py.gz = 0;
  if (x_packed_dev) px.gx = px.gy = px.gz = 0;
  const long long n_pack = (long long)py.gx * py.gy * py.gz + (long long)px.gx * px.gy * px.gz;
  if (n_pack > 0x7fffffffLL) return fail(EVMI_ERR_UNSUPPORTED, "conv1d_wgrad_cbt_bf16pk: grid limits (pack)");
  if (n_pack > 0) hipLaunchKernelGGL(pack2_kernel, dim3((unsigned)n_pack), dim3(256), 0, s, py, px);
  a.dyp = dy_packed_dev ? reinterpret_cast<const uint4*>(dy_packed_dev) : dyp;
  a.xp = x_packed_dev ? reinterpret_cast<const uint4*>(x_packed_dev) : xp;
  a.out = pl.splits > 1 ? part : dw_dev;
  a.accumulate = pl.splits > 1 ? 0 : accumulate;
  a.partial = pl.splits > 1;
  a.c_out = c_out;
  if (int rc = launch_wgrad_packed(a, pl, s)) return rc;
  if (pl.splits > 1) {
    const long long rows_ci = (long long)c_out * cin_g;
    if (rows_ci >= 131072)  // enough (co, ci) pairs to fill the chip with one workgroup per 256 of them: coalesced both ways
      hipLaunchKernelGGL(wgrad_pk_reduce_kernel, dim3((unsigned)((rows_ci + 255) / 256)), dim3(256), (size_t)k * 256 * sizeof(float), s, part,
                         dw_dev, rows_ci, k, pl.splits, a.split_stride, accumulate);
    else
      hipLaunchKernelGGL(wgrad_pk_reduce_planes_kernel, dim3((unsigned)((rows_ci + 255) / 256), k), dim3(256), 0, s, part, dw_dev, rows_ci, k,
                         pl.splits, a.split_stride, accumulate);
    EVMI_LAUNCH_CHECK("wgrad_pk_reduce_kernel");
  }
  return EVMI_OK;
}

}  // extern "C"
