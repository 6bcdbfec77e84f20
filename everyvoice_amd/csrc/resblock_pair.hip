// Instantiations and launcher of the fused residual-pair kernel (resblock_pair_kernel.h).
#include <cstdlib>

#include "resblock_pair32_kernel.h"
#include "resblock_pair_chunked_kernel.h"

namespace evmi {

//                  C  KS  BN  TAPS MAXDIL WAVES NWBUF OVL
// C = 32: all taps of one convolution fit LDS at once (single buffer): 2 weight steps per tile.  The first three entries are
// the two-workgroups-per-CU form (4 waves, 256 rows, T1 over XA), selected by EVMI_PAIR_OVL=1 only: measured 1.22 vs 1.16 ms
// on c32 / k11 (and 3.1 vs 1.95 ms for a 128-row two-workgroup form at C = 64) -- with half the waves per workgroup the phases
// of one workgroup are not filled by the other.
#define EVMI_PAIR_TABLE(X)         \
  X(32, 3, 256, 3, 5, 4, 2, 1)     \
  X(32, 7, 256, 7, 5, 4, 1, 1)     \
  X(32, 11, 256, 11, 5, 4, 1, 1)   \
  X(64, 3, 256, 2, 5, 8, 2, 0)     \
  X(64, 7, 256, 2, 5, 8, 2, 0)     \
  X(64, 11, 256, 2, 5, 8, 2, 0)    \
  X(32, 3, 512, 3, 5, 8, 2, 0)     \
  X(32, 7, 512, 7, 5, 8, 1, 0)     \
  X(32, 11, 512, 11, 5, 8, 1, 0)

static const PairLaunch* pair_table(int* n) {
#define X(c, ks, bn, taps, md, waves, nwbuf, ovl) \
  make_pair_launch<PairCfg<c, ks, bn, taps, md, waves, 0, nwbuf, ovl>>("resblock_pair_mfma<c" #c ",k" #ks ",bn" #bn ",t" #taps ">"),
  static const PairLaunch table[] = {
      // C = 32 as a conv1 / conv2 wave pipeline with the weights in registers (resblock_pair32_kernel.h): EVMI_PAIR32=1 only
      // (measured equal or slower than the single-team kernels below, see the header)
      make_pair32_launch<Pair32Cfg<3, 5>>("resblock_pair32<k3>"),
      make_pair32_launch<Pair32Cfg<7, 5>>("resblock_pair32<k7>"),
      make_pair32_launch<Pair32Cfg<11, 5>>("resblock_pair32<k11>"),
      EVMI_PAIR_TABLE(X)
      // C = 128: chunked variant (64-channel operand chunks, 2 x 4 waves of 64 x 64).  Measured on MI355X
      // (B=32, T=49152): k3 0.49 ms fused vs 0.56 ms as two conv_tc launches; k7 / k11 are MFMA/LDS-bound
      // either way and 3-4 % slower fused (0.86 vs 0.83, 1.19 vs 1.14 ms), so only k3 is routed here.
      make_pair_chunked_launch<PairChunkedCfg<128, 64, 3, 256, 5, 2, 4>>("resblock_pair_mfma<c128,k3,bn256,kc64>"),
  };
#undef X
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
  static const int ovl = [] {
    const char* e = getenv("EVMI_PAIR_OVL");
    return e ? atoi(e) : 0;
  }();
  static const bool pair32 = [] {
    const char* e = getenv("EVMI_PAIR32");
    return e && e[0] == '1';
  }();
  int n = 0;
  const PairLaunch* t = pair_table(&n);
  for (int i = pair32 ? 0 : 3; i < n; ++i)
    if (t[i].c == c && t[i].ks == ks && dil <= t[i].max_dil && (t[i].wg_per_cu == 1 || ovl)) return &t[i];
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

}  // namespace evmi
