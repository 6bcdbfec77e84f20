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

static const ConvTcEntry* conv_dma_table(int* n) {
#define X(cin, ks, md)                                                                          \
  ConvTcEntry{cin, ks, md, 64, make_conv_dma_launch<ConvDmaCfg<cin, ks, md>>("conv_tc_dma<c" #cin ",k" #ks ",bm128,bn256,kc64>")},
  static const ConvTcEntry table[] = {EVMI_CONV_DMA_TABLE(X)};
#undef X
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
