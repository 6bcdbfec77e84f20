// Implicit-GEMM 1-D convolution on MFMA for time-major / channel-last bf16 activations.
//
//   out[r, m] = post( [accumulate ? out : 0] + out_scale * ( residual[r, m] + maskfac[r, m] * ( bias[m]
//                      + sum_{j < KS} sum_{c < CIN} W[m][j][c] * pre(x[r + j*dil - pad, c]) ) ) )
//   (maskfac = 1 without a mask tensor, else mask[r, m] > 0 ? 1 : mask_slope: the activation backward of a training launch)
//
// GEMM view per block: D[BM channels x BN rows] = A[BM x (KS*CIN)] * B[(KS*CIN) x BN], where
// B is never materialised: the activation tile (BN + (KS-1)*dil rows x KC channels) sits in LDS
// once per channel chunk and tap j simply reads it at a row offset of j*dil.  Weights stream
// global -> registers -> LDS one tap-group ahead of the MFMAs (double-buffered).
//
// The same kernel runs ConvTranspose1d(k = 2*stride): in polyphase form that is a 2-tap
// convolution producing stride*C_out channels per input position, whose rows land contiguously
// in the [T*stride, C_out] output — expressed through out_row_stride / out_shift / out_limit.
#pragma once

#include "common.h"

namespace evmi {

struct ConvTcArgs {
  const bf16_t* x;     // [B][t_in][CIN]
  const bf16_t* w;     // [c_out][KS][CIN]
  const float* bias;   // [c_out]
  const bf16_t* res;   // same indexing as out, or nullptr
  bf16_t* out;
  int t_in;            // valid input rows
  int n_rows;          // output rows to produce
  int c_out;           // total output channels (all M tiles)
  int dil;
  int pad;             // input row of (r, j) = r + j*dil - pad
  long long x_batch_stride;    // elements
  long long out_batch_stride;  // elements
  long long out_row_stride;    // elements between consecutive output rows
  long long out_shift;         // added to row*out_row_stride + channel
  long long out_limit;         // flat indices outside [0, out_limit) are dropped
  float pre_slope;     // leaky-relu slope applied to x on load (1 = identity)
  float post_slope;    // leaky-relu slope applied to the final value (1 = identity)
  float out_scale;
  int accumulate;
  // training (input-gradient launches): the convolution's own value is multiplied by (mask > 0 ? 1 : mask_slope) BEFORE the
  // residual is added -- mask: a tensor indexed like out, the INPUT of the leaky ReLU in front of the convolution whose input
  // gradient this launch computes (or the output of that activation: same sign).  nullptr: no mask.
  const bf16_t* mask = nullptr;
  float mask_slope = 1.f;
  long long* timeline = nullptr;  // debug instantiations only (ABL bit 128): s_memtime stamps of wave 0 of workgroup (5, 1, 0)
  int n_items = 0;                // batch items (persistent kernels walk (item, row tile, m-tile) themselves; set by launch_conv_tc)
};

// One launch description, so the runtime can size LDS / grid without instantiating templates.
struct ConvTcLaunch {
  void (*kernel)(ConvTcArgs);
  int bm, bn, kc, threads;
  size_t lds_bytes;
  const char* name;
  // weight layout the kernel reads: 0 = [mtile][chunk][tap][BM][KC]; 1 = the same with the eight 16-byte channel vectors of a
  // row permuted for the LDS-DMA kernels (slot p of row m holds vector p ^ ((m >> 1) & 7), see conv_tc_dma_kernel.h)
  // 3 = [mtile][32-channel sub-chunk][tap][BM][32]: rows of four 16-byte vectors, slot p of row m holds vector p ^ ((m >> 2) & 3)
  // (tools/microbench/conv_tc_pp_kernel.h: the ping-pong kernel of round 5, measured slower and kept as a microbenchmark only)
  int wlayout = 0;
  int persistent = 0;  // 1: one workgroup per CU walks the tiles (grid = min(tiles, CUs))
  // the same kernel on 128-row tiles (same weight layout, same bits): picked by launch_conv_tc when the 256-row grid would leave
  // most of the chip idle; nullptr: none
  const ConvTcLaunch* narrow = nullptr;
};

// Returns nullptr if no instantiation covers (c_in, ks, max dilation).
const ConvTcLaunch* find_conv_tc(int c_in, int c_out, int ks, int dil);
int launch_conv_tc(const ConvTcLaunch* L, const ConvTcArgs& a, int B, hipStream_t stream);

}  // namespace evmi
