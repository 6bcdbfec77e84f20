// fp32 implicit-GEMM weight gradient of a 1-D convolution on the fp32-input matrix cores, channel-major layout:
//
//   dW[co][ci][j] (+)= sum_{b, to} dY[co][b][to] * X[ci][b][to*s + j*d - p]        (any stride / dilation / groups)
//
// GEMM view: M = output channels, N = (input channel, tap) -- exactly the memory order of a weight row, so the
// result tile stores coalesced -- and the contraction runs over the flattened (item, position) index.  No unfold:
// two cheap preparation passes make every operand load of the main kernel a contiguous, 16-byte aligned copy:
//   * pad_x_kernel   xp[c][b][i]  = x[c][b][i - p] (0 outside): every item becomes one contiguous segment of
//                    SEG = 4-aligned((n_out_pad - 1)*s + (k-1)*d + 1) samples, so the window a tile of 64 contraction
//                    indices needs (several short items or a piece of a long one, taps and item gaps included) is ONE
//                    contiguous range of the row;
//   * pad_dy_kernel  dyp[co][b*n_out_pad + to] = dY (0 for to >= n_out), n_out_pad = 4-aligned: K pairs of an MFMA never
//                    straddle items and rows are 64-float aligned.
// Main kernel: per step 64 contraction indices; dY tile [BM][64] and the X windows [BC channels][XW] travel global ->
// LDS with global_load_lds_dwordx4 into a ring (loads of step t+1/t+2 fly under the MFMAs of step t).  The dY rows
// are stored rotated by 4*(row mod 16) floats (the loader picks the source address per lane; LDS-direct writes are
// lane-linear) so that the A-operand reads, 32 rows at the same K index, spread over 16 banks instead of one.
// Per-lane operand addresses are constants plus a wave-uniform scalar, as in the forward kernel.
// Long contractions with few output tiles are split over workgroups (grid.z); partial tiles go to a workspace and
// are added in a fixed order (reproducible).
#include <algorithm>
#include <cstdint>
#include <cstdlib>

#include "common.h"
#include "conv_cbt_direct.h"

namespace evmi {

struct WgradArgs {
  const float* xp;   // [c_in][RX] padded input rows
  const float* dyp;  // [c_out][RD] padded output-gradient rows
  float* out;        // dW [c_out][cin_g][k], or partial workspace [S][c_out][cin_g*k]
  int cin_g, cout_g, k, stride, dil;
  int n_out_pad, seg;           // per item: contraction indices (4-aligned) and padded input samples
  long long rx, rd;             // row strides of xp / dyp
  int nchunks, chunks_per_split;
  int bc;                       // input channels per column tile (bc * k <= BN)
  int xw;                       // staged window length per channel (multiple of 4)
  int nxi;                      // X load instructions per step per workgroup = ceil(bc * xw / 256)
  int stage;                    // floats per ring slot
  int accumulate, splits;
  int mtiles_per_group, ctiles;
  long long out_split_stride;   // floats between partial results of consecutive splits
};

typedef __attribute__((address_space(3))) float wg_lds_float_t;
typedef __attribute__((address_space(1))) const float wg_glb_float_t;
__device__ __forceinline__ void wg_lds_direct_b128(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((wg_glb_float_t*)g, (wg_lds_float_t*)l, 16, 0, 0);
}

// xp[c][b*seg + i] = x[c][b][i - pad] for 0 <= i - pad < t_in, i < seg; 0 elsewhere (including the row tail)
__global__ void pad_x_kernel(const float* __restrict__ x, float* __restrict__ xp, int B, int t_in, int seg, int pad, long long rx) {
  const long long c = blockIdx.y;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rx) return;
  const long long b = i / seg;
  const int r = (int)(i - b * seg) - pad;
  xp[c * rx + i] = (b < B && r >= 0 && r < t_in) ? x[(c * B + b) * t_in + r] : 0.f;
}

// dyp[co][b*n_out_pad + to] = dy[co][b][to] for to < n_out; 0 elsewhere (including the row tail)
__global__ void pad_dy_kernel(const float* __restrict__ dy, float* __restrict__ dyp, int B, int n_out, int n_out_pad, long long rd) {
  const long long c = blockIdx.y;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rd) return;
  const long long b = i / n_out_pad;
  const int to = (int)(i - b * n_out_pad);
  dyp[c * rd + i] = (b < B && to < n_out) ? dy[(c * B + b) * n_out + to] : 0.f;
}

// dw[i] (+)= sum_s part[s][i]
__global__ void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long long n, int splits, int accumulate) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  dw[i] = ordered_sum_strided(part + i, n, splits, accumulate ? dw[i] : 0.f);
}

constexpr int WG_NK = 64;    // contraction indices per step
constexpr int WG_BN = 128;   // (channel, tap) columns per tile
constexpr int WG_NXI = 8;    // X load instructions per step per wave (registers for their source offsets)

template <int BM, int WM, int WN, int NST>
__global__ __launch_bounds__(256, 2) void conv_wgrad_f32_mfma_kernel(WgradArgs a) {
  static_assert(WM * WN == 4, "four waves");
  constexpr int MT = BM / (WM * 32), NT = WG_BN / (WN * 32);
  constexpr int A_FLOATS = BM * WG_NK;
  constexpr int A_INSTR = A_FLOATS / 256 / 4;  // dY load instructions per step per wave
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int kh = lane >> 5, ln = lane & 31;
  const int g = blockIdx.y / a.mtiles_per_group, mt_idx = blockIdx.y % a.mtiles_per_group;
  const int ci0 = blockIdx.x * a.bc;
  const int bc_cur = min(a.bc, a.cin_g - ci0);
  const int k = a.k, s = a.stride, d = a.dil, xw = a.xw;
  const int m_valid = min(BM, a.cout_g - mt_idx * BM);
  const int co0 = g * a.cout_g + mt_idx * BM;
  const int c_lo = blockIdx.z * a.chunks_per_split;
  const int c_hi = min(a.nchunks, c_lo + a.chunks_per_split);
  const int nsteps = c_hi - c_lo;

  // per-lane operand constants
  int abase[MT], arot[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + ln;
    abase[mt] = m * WG_NK;
    arot[mt] = kh + 4 * (m & 15);
  }
  int xbase[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (wn * NT + nt) * 32 + ln;
    const int cl = c / k, j = c - cl * k;
    xbase[nt] = A_FLOATS + (cl < bc_cur ? cl * xw + j * d : 0) + kh * s;  // columns past the tile read valid data, never stored
  }
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // loader constants: dY: instruction i of this wave covers rows (wave*A_INSTR + i)*4 .. +3, lane = (row, 16-byte slot)
  const int a_row_in = lane >> 4, a_slot = lane & 15;
  long long a_src[A_INSTR];
#pragma unroll
  for (int i = 0; i < A_INSTR; ++i) {
    const int row = (wave * A_INSTR + i) * 4 + a_row_in;
    const int rsrc = min(row, m_valid - 1);  // rows past the group re-read the last one (never stored)
    a_src[i] = (long long)(co0 + rsrc) * a.rd + ((4 * a_slot - 4 * (row & 15)) & 63);
  }
  // X windows: linear LDS region [bc][xw], 256 floats per instruction, instructions round-robin over the waves
  int x_off[WG_NXI];
  bool x_live[WG_NXI];
#pragma unroll
  for (int i = 0; i < WG_NXI; ++i) {
    const int f = ((i * 4 + wave) * 64 + lane) * 4;  // first of this lane's 4 floats in the staged region
    const int row = f / xw, col = f - row * xw;
    x_live[i] = (i * 4 + wave) < a.nxi && row < a.bc;
    x_off[i] = min(row, max(bc_cur - 1, 0)) * (int)a.rx + col;  // channels past the group re-read the last one
  }
  const float* xp_tile = a.xp + (long long)(g * a.cin_g + ci0) * a.rx;

  // window start of a chunk: P(n') = b*seg + to*s with n' = b*n_out_pad + to
  int nb, nto;
  {
    const long long n0 = (long long)c_lo * WG_NK;
    nb = (int)(n0 / a.n_out_pad);
    nto = (int)(n0 - (long long)nb * a.n_out_pad);
  }
  auto issue = [&](int t, int slot, int b0, int to0) {
    float* sa = smem + slot * a.stage;
    float* sx = sa + A_FLOATS;
    const long long n0 = (long long)(c_lo + t) * WG_NK;
#pragma unroll
    for (int i = 0; i < A_INSTR; ++i)
      wg_lds_direct_b128(a.dyp + a_src[i] + n0, sa + (wave * A_INSTR + i) * 256);
    const float* xsrc = xp_tile + (long long)b0 * a.seg + (long long)to0 * s;
#pragma unroll
    for (int i = 0; i < WG_NXI; ++i)
      if (x_live[i]) wg_lds_direct_b128(xsrc + x_off[i], sx + (i * 4 + wave) * 256);
  };
  auto advance_chunk = [&]() {
    nto += WG_NK;
    while (nto >= a.n_out_pad) { nto -= a.n_out_pad; ++nb; }
  };

  // prologue: NST - 1 steps in flight
  int ib = nb, ito = nto;  // issue cursor
  auto issue_next = [&](int t) {
    issue(t, t % NST, ib, ito);
    ito += WG_NK;
    while (ito >= a.n_out_pad) { ito -= a.n_out_pad; ++ib; }
  };
#pragma unroll
  for (int t = 0; t < NST - 1; ++t)
    if (t < nsteps) issue_next(t);

  for (int t = 0; t < nsteps; ++t) {
    const int slot = t % NST;
    // loads of step t have landed when at most the younger groups are outstanding (same count every step)
    const int younger = min(NST - 2, nsteps - 1 - t);
    int per = A_INSTR;  // every step issues the same number of loads per wave: dY rows + this wave's window pieces
#pragma unroll
    for (int i = 0; i < WG_NXI; ++i) per += (i * 4 + wave) < a.nxi ? 1 : 0;
    wait_vmcnt_le(younger > 0 ? per * younger : 0);
    lds_barrier();
    if (t + NST - 1 < nsteps) issue_next(t + NST - 1);

    // contraction over the 64 indices of this step: pairs (2q, 2q+1) of one item
    int to = nto, off_x = slot * a.stage, q2 = 0;
    const int off_a = slot * a.stage;
    float fa[2][MT], fb[2][NT], ga[2][MT], gb[2][NT];
    auto load_group = [&](float (&da)[2][MT], float (&db)[2][NT]) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) da[u][mt] = smem[off_a + abase[mt] + ((q2 + arot[mt]) & 63)];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) db[u][nt] = smem[xbase[nt] + off_x];
        q2 += 2; to += 2; off_x += 2 * s;
        if (to >= a.n_out_pad) { to = 0; off_x += a.seg - a.n_out_pad * s; }
      }
    };
    load_group(fa, fb);
    for (int gi = 0; gi < WG_NK / 4; ++gi) {
      load_group(ga, gb);  // past the end of the step: in-bounds LDS, never used
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
            acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][mt], fb[u][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[u][mt] = ga[u][mt];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[u][nt] = gb[u][nt];
      }
#pragma unroll
      for (int i = 0; i < 2 * MT * NT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
      }
    }
    advance_chunk();
  }

  // ---- epilogue: D layout: lane = (channel, tap) column = consecutive floats of a weight row ----
  float* outp = a.out + (long long)blockIdx.z * a.out_split_stride;
  const long long kg = (long long)a.cin_g * k;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = (wn * NT + nt) * 32 + ln;
    if (c >= bc_cur * k) continue;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      // previous values of 8 rows at a time, from clamped rows (a read under a per-element condition drains vmcnt each time)
#pragma unroll
      for (int r0 = 0; r0 < 16; r0 += 8) {
        float* dst[8];
        float prev[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = r0 + e;
          const int m = min((wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, m_valid - 1);
          dst[e] = outp + (long long)(co0 + m) * kg + (long long)ci0 * k + c;
          prev[e] = -0.f;
        }
        if (a.accumulate && a.splits == 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) prev[e] = *dst[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int r = r0 + e;
          if ((wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh < m_valid) *dst[e] = prev[e] + acc[mt][nt][r];
        }
      }
    }
  }
}

struct WgradPlan {
  int bm, nst, n_out_pad, seg, bc, xw, nxi, stage, nchunks, splits, chunks_per_split, mtiles, ctiles;
  long long rx, rd, ws_elems;
  size_t lds;
};

// nullptr = runnable
static const char* plan_wgrad(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups,
                              WgradPlan& p) {
  if (groups <= 0 || c_in <= 0 || c_out <= 0 || c_in % groups || c_out % groups || B <= 0 || n_out <= 0 || k <= 0 || stride <= 0 || dil <= 0)
    return "bad shape";
  const int cin_g = c_in / groups, cout_g = c_out / groups;
  if (k > WG_BN) return "kernel size above 128";
  p.n_out_pad = (n_out + 3) & ~3;
  p.seg = ((p.n_out_pad - 1) * stride + (k - 1) * dil + 1 + 3) & ~3;
  p.bc = std::min(cin_g, WG_BN / k);
  // window of one step: 64 indices = up to ceil(63 / n_out_pad) + 1 items; every item boundary adds (seg - n_out_pad*stride)
  const int items = (WG_NK - 1) / p.n_out_pad + 2;
  const int gap = std::max(0, p.seg - p.n_out_pad * stride);
  p.xw = (WG_NK * stride + items * gap + (k - 1) * dil + 1 + 3) & ~3;
  p.nxi = (p.bc * p.xw + 255) / 256;
  if (p.nxi > 4 * WG_NXI) return "input window too large for the loader";
  p.bm = cout_g > 32 ? 64 : 32;  // 16 KB of dY per slot: three slots and two workgroups per CU
  if (cout_g >= 128) p.bm = 128;  // 64x64 wave tiles, one workgroup per CU: +10-13 % on the dense 1024-channel layers
  const int a_floats = p.bm * WG_NK;
  p.stage = (a_floats + p.nxi * 256 + 2 * stride + 64 + 3) & ~3;
  p.nst = 3;
  if ((size_t)3 * p.stage * sizeof(float) > 80 * 1024) p.nst = 2;
  p.lds = (size_t)p.nst * p.stage * sizeof(float);
  if (p.lds > 160 * 1024) return "LDS budget";
  const long long np = (long long)B * p.n_out_pad;
  p.nchunks = (int)((np + WG_NK - 1) / WG_NK);
  p.rd = (long long)p.nchunks * WG_NK;
  p.rx = (((long long)B * p.seg + p.xw + 3) & ~3LL) + 256;
  if (p.rx * p.bc >= (1LL << 31) || (long long)B * p.seg >= (1LL << 31)) return "row too long";
  p.mtiles = (cout_g + p.bm - 1) / p.bm;
  p.ctiles = (cin_g + p.bc - 1) / p.bc;
  const long long tiles = (long long)p.mtiles * p.ctiles * groups;
  long long splits = std::max<long long>(1, std::min<long long>((640 + tiles - 1) / tiles, p.nchunks / 4));
  splits = std::min<long long>(splits, 64);
  p.chunks_per_split = (int)((p.nchunks + splits - 1) / splits);
  p.splits = (p.nchunks + p.chunks_per_split - 1) / p.chunks_per_split;
  if (groups * p.mtiles > 65535 || p.splits > 65535) return "grid limits";
  const long long w_elems = (long long)c_out * cin_g * k;
  p.ws_elems = (long long)c_in * p.rx + (long long)c_out * p.rd + (p.splits > 1 ? (long long)p.splits * w_elems : 0) + 64;
  return nullptr;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

/* Floats of workspace evmi_conv1d_wgrad_cbt_f32 needs for this shape; 0 when the shape is not supported. */
long long evmi_conv1d_wgrad_cbt_f32_ws_elems(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil,
                                             int groups) {
  if (const long long direct = wgrad_cin1_plan(B, c_in, n_out, c_out, k, groups)) return direct + 64;  // (conv_cbt_direct.hip)
  WgradPlan p;
  return plan_wgrad(B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups, p) ? 0 : p.ws_elems;
}

int evmi_conv1d_wgrad_cbt_f32(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev, long long ws_elems, int B,
                              int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups,
                              int accumulate, void* stream) {
  if (!x_dev || !dy_dev || !dw_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_cbt_f32: null pointer");
  if (wgrad_cin1_plan(B, c_in, n_out, c_out, k, groups))  // one input channel: an outer product per column, not a GEMM
    return launch_wgrad_cin1(x_dev, dy_dev, dw_dev, ws_dev, ws_elems, B, t_in, n_out, c_out, k, stride, pad, dil, accumulate, (hipStream_t)stream);
  WgradPlan p;
  if (const char* why = plan_wgrad(B, c_in, t_in, c_out, n_out, k, stride, pad, dil, groups, p))
    return fail(EVMI_ERR_UNSUPPORTED, std::string("conv1d_wgrad_cbt_f32: ") + why);
  if (ws_elems < p.ws_elems || (reinterpret_cast<uintptr_t>(ws_dev) & 15))
    return fail(EVMI_ERR_INVALID_ARG, "conv1d_wgrad_cbt_f32: workspace too small or unaligned");
  hipStream_t s = (hipStream_t)stream;
  const int cin_g = c_in / groups, cout_g = c_out / groups;
  float* xp = ws_dev;
  float* dyp = xp + (((long long)c_in * p.rx + 3) & ~3LL);
  float* part = dyp + (((long long)c_out * p.rd + 3) & ~3LL);
  hipLaunchKernelGGL(pad_x_kernel, dim3((unsigned)((p.rx + 255) / 256), c_in), dim3(256), 0, s, x_dev, xp, B, t_in, p.seg, pad, p.rx);
  hipLaunchKernelGGL(pad_dy_kernel, dim3((unsigned)((p.rd + 255) / 256), c_out), dim3(256), 0, s, dy_dev, dyp, B, n_out, p.n_out_pad, p.rd);
  EVMI_LAUNCH_CHECK("conv1d_wgrad pad");

  WgradArgs a;
  a.xp = xp; a.dyp = dyp;
  a.cin_g = cin_g; a.cout_g = cout_g; a.k = k; a.stride = stride; a.dil = dil;
  a.n_out_pad = p.n_out_pad; a.seg = p.seg; a.rx = p.rx; a.rd = p.rd;
  a.nchunks = p.nchunks; a.chunks_per_split = p.chunks_per_split; a.bc = p.bc; a.xw = p.xw; a.nxi = p.nxi; a.stage = p.stage;
  a.accumulate = accumulate; a.splits = p.splits; a.mtiles_per_group = p.mtiles; a.ctiles = p.ctiles;
  const long long w_elems = (long long)c_out * cin_g * k;
  a.out = p.splits > 1 ? part : dw_dev;
  a.out_split_stride = p.splits > 1 ? w_elems : 0;
  const dim3 grid(p.ctiles, groups * p.mtiles, p.splits);
  static thread_local size_t configured_dev[kMaxDevices][6] = {};
  size_t* configured = configured_dev[device_slot()];
#define EVMI_WG_LAUNCH(BM, WM, WN, NST, IDX)                                                                        \
  {                                                                                                                 \
    if (p.lds > configured[IDX]) {                                                                                  \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)conv_wgrad_f32_mfma_kernel<BM, WM, WN, NST>,                  \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));                 \
      configured[IDX] = p.lds;                                                                                      \
    }                                                                                                               \
    hipLaunchKernelGGL((conv_wgrad_f32_mfma_kernel<BM, WM, WN, NST>), grid, dim3(256), p.lds, s, a);                \
  }
  if (p.bm == 128 && p.nst == 3) EVMI_WG_LAUNCH(128, 2, 2, 3, 0)
  else if (p.bm == 128) EVMI_WG_LAUNCH(128, 2, 2, 2, 1)
  else if (p.bm == 64 && p.nst == 3) EVMI_WG_LAUNCH(64, 2, 2, 3, 2)
  else if (p.bm == 64) EVMI_WG_LAUNCH(64, 2, 2, 2, 3)
  else if (p.nst == 3) EVMI_WG_LAUNCH(32, 1, 4, 3, 4)
  else EVMI_WG_LAUNCH(32, 1, 4, 2, 5)
#undef EVMI_WG_LAUNCH
  EVMI_LAUNCH_CHECK("conv_wgrad_f32_mfma");
  if (p.splits > 1) {
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((w_elems + 255) / 256)), dim3(256), 0, s, part, dw_dev, w_elems, p.splits,
                       accumulate);
    EVMI_LAUNCH_CHECK("wgrad_reduce");
  }
  return EVMI_OK;
}

}  // extern "C"
