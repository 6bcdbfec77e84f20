// Instantiations and launcher of the fused residual-pair kernel (resblock_pair_kernel.h).
#include <cstdlib>

#include "resblock_branch_kernel.h"
#include "resblock_pair_chunked_kernel.h"

namespace evmi {

//                  C  KS  BN  TAPS MAXDIL WAVES NWBUF OVL WRES WM   (sixteen waves at 32 channels: four per SIMD keep the VALU-bound activation
//                  passes and the LDS latency of the short MFMA steps covered, -7 ... -9 %; at 64 channels the 2 (channels) x 8 (rows) split
//                  costs a fragment read per MFMA instead of 0.75 and measured equal or slower: eight waves there)
// C = 32: all taps of one convolution fit LDS at once (single buffer): 2 weight steps per tile.  (Measured and removed in round 5:
// a two-workgroups-per-CU form with T1 overlaying the operand tile, 1.22 vs 1.16 ms on c32 / k11; a conv1 / conv2 wave pipeline
// with the weights in registers, 17.82 vs 17.63 ms per forward -- DESIGN.md §9.7, §9.11.)
#define EVMI_PAIR_TABLE(X)               \
  X(64, 3, 256, 2, 5, 8, 2, 0, WR, 1)    \
  X(64, 7, 256, 2, 5, 8, 2, 0, WR, 1)    \
  X(64, 11, 256, 2, 5, 8, 2, 0, WR, 1)   \
  X(32, 3, 512, 3, 5, 8, 2, 0, 0, 1)     \
  X(32, 7, 512, 7, 5, 8, 1, 0, 0, 1)     \
  X(32, 11, 512, 11, 5, 16, 1, 0, WR, 1)

static const PairLaunch* pair_table(int* n) {
#define WR 1  // weights resident in registers (PairCfg::WRES)
#define X(c, ks, bn, taps, md, waves, nwbuf, ovl, wres, wm) \
  make_pair_launch<PairCfg<c, ks, bn, taps, md, waves, 0, nwbuf, ovl, wres, wm>>("resblock_pair_mfma<c" #c ",k" #ks ",bn" #bn ",t" #taps ">"),
  static const PairLaunch table[] = {
      EVMI_PAIR_TABLE(X)
      // C = 128: chunked variant (64-channel operand chunks, 2 x 4 waves of 64 x 64).  Measured on MI355X
      // (B=32, T=49152): k3 0.49 ms fused vs 0.56 ms as two conv_tc launches; k7 / k11 are MFMA/LDS-bound
      // either way and 3-4 % slower fused (0.86 vs 0.83, 1.19 vs 1.14 ms), so only k3 is routed here.
      make_pair_chunked_launch<PairChunkedCfg<128, 64, 3, 256, 5, 2, 4>>("resblock_pair_mfma<c128,k3,bn256,kc64>"),
  };
#undef X
#undef WR
  *n = (int)(sizeof(table) / sizeof(table[0]));
  return table;
}

const PairLaunch* find_resblock_pair(int c, int ks, int dil) {
  // A/B switch: EVMI_PAIR_C128=0 routes the 128-channel k = 3 pairs to two LDS-DMA convolution launches instead
  static const bool c128_pairs = [] {
    const char* e = getenv("EVMI_PAIR_C128");
    return !(e && e[0] == '0');
  }();
  if (c >= 128 && !c128_pairs) return nullptr;
  int n = 0;
  const PairLaunch* t = pair_table(&n);
  for (int i = 0; i < n; ++i)
    if (t[i].c == c && t[i].ks == ks && dil <= t[i].max_dil) return &t[i];
  return nullptr;
}

int launch_resblock_pair(const PairLaunch* L, PairArgs a, int B, int n_cu, hipStream_t stream) {
  static thread_local const void* configured_dev[kMaxDevices][32];
  static thread_local int n_configured_dev[kMaxDevices] = {};
  const int dev_slot = device_slot();
  const void** configured = configured_dev[dev_slot];
  int& n_configured = n_configured_dev[dev_slot];
  bool seen = false;
  for (int i = 0; i < n_configured; ++i) seen |= (configured[i] == (const void*)L->kernel);
  if (!seen) {
    EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)L->kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)L->lds_bytes));
    if (n_configured < 32) configured[n_configured++] = (const void*)L->kernel;
  }
  a.tiles_per_item = (a.T + L->tt - 1) / L->tt;
  a.n_tiles = a.tiles_per_item * B;
  // persistent: one workgroup per CU (LDS-bound residency), a multiple of 8 so every XCD gets the
  // same number of workgroups (the kernel's tile walk relies on it)
  int grid = (n_cu > 0 ? n_cu : 256) * L->wg_per_cu;
  grid = (grid + 7) / 8 * 8;
  const int needed = (a.n_tiles + 7) / 8 * 8;
  if (grid > needed) grid = needed;
  hipLaunchKernelGGL(L->kernel, dim3(grid), dim3(L->threads), L->lds_bytes, stream, a);
  EVMI_LAUNCH_CHECK(L->name);
  return EVMI_OK;
}

// ---- whole branches (resblock_branch_kernel.h) ------------------------------------------------------------------------------------------
//                    C  KS  BN  TAPS WAVES NWBUF
// Only where the pair kernels are bound by the residual stream's round trips and the branch's halo stays small: k = 3 and k = 7 at 32
// channels, k = 3 at 64 (k = 11 recomputes 23 % of every tile and is MFMA-bound as pairs already; 64 channels x k = 7 does not fit).
static const BranchLaunch* branch_table(int* n) {
  static const BranchLaunch table[] = {
      make_branch_launch<BranchCfg<32, 3, 512, 3, 16, 2>>("resblock_branch_mfma<c32,k3,bn512>"),
      make_branch_launch<BranchCfg<32, 7, 512, 7, 16, 1>>("resblock_branch_mfma<c32,k7,bn512>"),
      make_branch_launch<BranchCfg<64, 3, 256, 2, 8, 2>>("resblock_branch_mfma<c64,k3,bn256>"),
  };
  *n = (int)(sizeof(table) / sizeof(table[0]));
  return table;
}

// valid rows per tile of a branch with these dilations, or 0 when a convolution would read past the LDS tiles
static int branch_valid_rows(const BranchLaunch& L, int np, const int* dil) {
  const int H = (L.ks - 1) / 2;
  if (np != 3 || dil[0] < 1 || L.bn + 2 * dil[0] * H > L.rb) return 0;
  int m = 0;
  for (int p = 0; p < np; ++p) {
    const int h1 = dil[p] * H;
    if (dil[p] < 1 || m + 2 * h1 + L.bn > L.rb || m + h1 + L.bn + 2 * H > L.rb) return 0;
    m += h1 + H;
  }
  const int tt = L.bn + 2 * dil[0] * H - 2 * m;
  return tt > 0 ? tt : 0;
}

const BranchLaunch* find_resblock_branch(int c, int ks, int np, const int* dil) {
  // A/B switch: EVMI_BRANCH=0 keeps the pair kernels (same bits: tests/test_gpu_generator.py compares the two)
  static const bool enabled = [] {
    const char* e = getenv("EVMI_BRANCH");
    return !(e && e[0] == '0');
  }();
  if (!enabled) return nullptr;
  int n = 0;
  const BranchLaunch* t = branch_table(&n);
  for (int i = 0; i < n; ++i)
    if (t[i].c == c && t[i].ks == ks && 4 * branch_valid_rows(t[i], np, dil) >= 3 * t[i].bn) return &t[i];  // (a halo above 25 % does not pay)
  return nullptr;
}

int launch_resblock_branch(const BranchLaunch* L, BranchArgs a, int B, int n_cu, hipStream_t stream) {
  static thread_local const void* configured_dev[kMaxDevices][8];
  static thread_local int n_configured_dev[kMaxDevices] = {};
  const int dev_slot = device_slot();
  const void** configured = configured_dev[dev_slot];
  int& n_configured = n_configured_dev[dev_slot];
  bool seen = false;
  for (int i = 0; i < n_configured; ++i) seen |= (configured[i] == (const void*)L->kernel);
  if (!seen) {
    EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)L->kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L->lds_bytes));
    if (n_configured < 8) configured[n_configured++] = (const void*)L->kernel;
  }
  a.tt = branch_valid_rows(*L, a.np, a.dil);
  if (a.tt <= 0) return fail(EVMI_ERR_UNSUPPORTED, "resblock_branch: dilations beyond the LDS tiles");
  a.tiles_per_item = (a.T + a.tt - 1) / a.tt;
  a.n_tiles = a.tiles_per_item * B;
  int grid = (n_cu > 0 ? n_cu : 256);  // persistent: one workgroup per CU, a multiple of 8 (the kernel's tile walk relies on it)
  grid = (grid + 7) / 8 * 8;
  const int needed = (a.n_tiles + 7) / 8 * 8;
  if (grid > needed) grid = needed;
  hipLaunchKernelGGL(L->kernel, dim3(grid), dim3(L->threads), L->lds_bytes, stream, a);
  EVMI_LAUNCH_CHECK(L->name);
  return EVMI_OK;
}

}  // namespace evmi
