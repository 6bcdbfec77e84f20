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
#include <cstdlib>

#include "conv_tc_dma_kernel.h"
#include "conv_tc_kernel.h"

namespace evmi {

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
  e.launch = make_conv_tc_launch<C>(name);
  return e;
}

//                       CIN  KC   BM   BN  WM WN KS TAPS MAXDIL
// Shapes picked by tools/sweep_conv.py on MI355X (profiles/r01b_sweep*.txt): 512-thread blocks of
// 256 rows (8 waves) beat 256-thread / 128-row blocks by 1.3-1.5x at every channel count.
#define EVMI_CONV_TC_TABLE(X)                                                        \
  /* conv_pre 80 -> 512, k7 */                                                       \
  X(80, 80, 128, 128, 2, 2, 7, 1, 1)                                                 \
  /* transposed-conv upsamplers in polyphase form (2 taps) */                        \
  X(512, 64, 128, 256, 2, 4, 2, 1, 1)                                                \
  X(256, 64, 128, 256, 2, 4, 2, 1, 1)                                                \
  X(128, 64, 128, 256, 2, 4, 2, 1, 1)                                                \
  X(64, 64, 64, 256, 1, 8, 2, 2, 1)                                                  \
  /* MRF residual-block convolutions, dilation <= 5 */                               \
  X(256, 64, 128, 256, 2, 4, 3, 1, 5)                                                \
  X(256, 64, 128, 256, 2, 4, 7, 1, 5)                                                \
  X(256, 64, 128, 256, 2, 4, 11, 1, 5)                                               \
  X(128, 64, 128, 256, 2, 4, 3, 1, 5)                                                \
  X(128, 64, 128, 256, 2, 4, 7, 1, 5)                                                \
  X(128, 64, 128, 256, 2, 4, 11, 1, 5)                                               \
  X(64, 64, 64, 256, 1, 8, 3, 3, 5)                                                  \
  X(64, 64, 64, 256, 1, 8, 7, 1, 5)                                                  \
  X(64, 64, 64, 256, 1, 8, 11, 1, 5)                                                 \
  X(32, 32, 32, 512, 1, 8, 3, 3, 5)                                                  \
  X(32, 32, 32, 512, 1, 8, 7, 7, 5)                                                  \
  X(32, 32, 32, 512, 1, 8, 11, 11, 5)

static const ConvTcEntry* conv_tc_table(int* n) {
#define X(cin, kc, bm, bn, wm, wn, ks, taps, md)                                               \
  make_entry<ConvTcCfg<cin, kc, bm, bn, wm, wn, ks, taps, md>>(                                \
      "conv_tc_mfma<c" #cin ",k" #ks ",bm" #bm ",bn" #bn ",kc" #kc ",t" #taps ">"),
  static const ConvTcEntry table[] = {EVMI_CONV_TC_TABLE(X)};
#undef X
  *n = (int)(sizeof(table) / sizeof(table[0]));
  return table;
}

// LDS-DMA kernels for the wide layers (conv_tc_dma_kernel.h); EVMI_CONV_DMA=0 selects the register-staged kernels (A/B)
#define EVMI_CONV_DMA_TABLE(X) \
  X(512, 2, 1) X(256, 2, 1) X(128, 2, 1) X(256, 3, 5) X(256, 7, 5) X(256, 11, 5) X(128, 3, 5) X(128, 7, 5) X(128, 11, 5)

// narrow row tiles of the residual-stack layers (bn 128: four waves), for grids that would not fill the chip.  (64-row tiles, two
// waves, were measured too: every workgroup streams the layer's whole weight set, and at 64 rows that costs more than the extra
// workgroups give -- 1 x 400 frames 1.13 ms with 256-row tiles, 1.00 ms with 128-row tiles, 1.14-1.23 ms with 64-row tiles.)
#define EVMI_CONV_DMA_NARROW_TABLE(X) X(256, 3, 5) X(256, 7, 5) X(256, 11, 5) X(128, 3, 5) X(128, 7, 5) X(128, 11, 5)

static const ConvTcEntry* conv_dma_table(int* n) {
#define X(cin, ks, md)                                                                          \
  ConvTcEntry{cin, ks, md, 64, make_conv_dma_launch<ConvDmaCfg<cin, ks, md>>("conv_tc_dma<c" #cin ",k" #ks ",bm128,bn256,kc64>")},
  static ConvTcEntry table[] = {EVMI_CONV_DMA_TABLE(X)};
#undef X
#define X(cin, ks, md)                                                                                                                   \
  ConvTcEntry{cin, ks, md, 64, make_conv_dma_launch<ConvDmaCfg<cin, ks, md, 0, 0, 2>>("conv_tc_dma<c" #cin ",k" #ks ",bm128,bn128,kc64>")},
  static const ConvTcEntry narrow[] = {EVMI_CONV_DMA_NARROW_TABLE(X)};
#undef X
  static const bool linked = [] {
    for (ConvTcEntry& e : table)
      for (const ConvTcEntry& nr : narrow)
        if (nr.c_in == e.c_in && nr.ks == e.ks && nr.max_dil == e.max_dil) e.launch.narrow = &nr.launch;
    return true;
  }();
  (void)linked;
  *n = (int)(sizeof(table) / sizeof(table[0]));
  return table;
}

const ConvTcLaunch* find_conv_tc(int c_in, int c_out, int ks, int dil) {
  static const bool use_dma = [] {
    const char* e = getenv("EVMI_CONV_DMA");
    return !(e && e[0] == '0');
  }();
  if (use_dma) {
    int nd = 0;
    const ConvTcEntry* d = conv_dma_table(&nd);
    for (int i = 0; i < nd; ++i)
      if (d[i].c_in == c_in && d[i].ks == ks && dil <= d[i].max_dil && c_out % d[i].launch.bm == 0) return &d[i].launch;
  }
  int n = 0;
  const ConvTcEntry* t = conv_tc_table(&n);
  for (int i = 0; i < n; ++i) {
    if (t[i].c_in == c_in && t[i].ks == ks && dil <= t[i].max_dil && c_out % t[i].launch.bm == 0)
      return &t[i].launch;
  }
  return nullptr;
}

int launch_conv_tc(const ConvTcLaunch* L, const ConvTcArgs& a, int B, hipStream_t stream) {
  // 128-row tiles while the 256-row grid is far from filling the chip (two workgroups per CU = 512 slots): one utterance at a time
  // (1 x 400 frames: 1.13 -> 1.00 ms per forward), the GAN step's generator (16 items of 256 / 2048 rows per stage: 32 / 128
  // workgroups; the step itself does not move, those launches are not on its critical chain).  EVMI_CONV_NARROW=0: never.
  // The bits do not depend on the tile (same K order per output element).
  static const int narrow_on = [] {
    const char* e = getenv("EVMI_CONV_NARROW");
    return e ? atoi(e) : 1;
  }();
  if (narrow_on && !L->persistent && L->narrow && (long long)((a.n_rows + L->bn - 1) / L->bn) * B * (a.c_out / L->bm) < 256) L = L->narrow;
  static thread_local const void* configured_dev[kMaxDevices][64];
  static thread_local int n_configured_dev[kMaxDevices] = {};
  const int dev_slot = device_slot();
  const void** configured = configured_dev[dev_slot];
  int& n_configured = n_configured_dev[dev_slot];
  bool seen = false;
  for (int i = 0; i < n_configured; ++i) seen |= (configured[i] == (const void*)L->kernel);
  if (!seen) {
    EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)L->kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)L->lds_bytes));
    if (n_configured < 64) configured[n_configured++] = (const void*)L->kernel;
  }
  dim3 grid((a.n_rows + L->bn - 1) / L->bn, B, a.c_out / L->bm);
  ConvTcArgs args = a;
  args.n_items = B;
  if (L->persistent) {
    static thread_local int n_cu_dev[kMaxDevices] = {};
    if (!n_cu_dev[dev_slot]) {
      int dev = 0;
      hipDeviceProp_t prop;
      EVMI_HIP_CHECK(hipGetDevice(&dev));
      EVMI_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
      n_cu_dev[dev_slot] = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long n_rb = (long long)B * grid.x;
    const long long total = (n_rb + 7) / 8 * 8 * grid.z;
    grid = dim3((unsigned)(total < n_cu_dev[dev_slot] ? total : n_cu_dev[dev_slot]), 1, 1);
  }
  hipLaunchKernelGGL(L->kernel, grid, dim3(L->threads), L->lds_bytes, stream, args);
  EVMI_LAUNCH_CHECK(L->name);
  return EVMI_OK;
}

}  // namespace evmi
