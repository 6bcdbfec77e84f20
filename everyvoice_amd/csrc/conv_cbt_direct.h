// Direct fp32 convolutions for GEMV / outer-product shapes of the training layout (conv_cbt_direct.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace evmi {

struct ConvDirectArgs {
  const float* x;
  const float* w;
  const float* bias;
  float* y;
  float* partial;  // few-output-channel kernel, channel-split mode: [nchunks][c_out][B*n_out]
  int B, t_in, t_out_total, n_out;
  int c_in, c_out, k, stride, dil, pad;
  int out_stride, out_offset, accumulate;
  int act;          // epilogue: 0 none, 1 leaky-relu(act_param), 2 SiLU, 3 ReLU, 4 tanh
  float act_param;
  int cc, nchunks;  // channels per workgroup / number of channel chunks
};

// 0: not a direct-kernel shape; otherwise 1 + floats of scratch the few-output-channel kernel needs
long long conv_direct_plan(const ConvDirectArgs& in, int groups, int& cc, int& nchunks);
int launch_conv_direct(ConvDirectArgs a, int groups, float* ws, long long ws_elems, hipStream_t stream);

// Weight gradient of a one-input-channel convolution (x [B][t_in], dy [c_out][B][n_out] -> dw [c_out][k]).  Plan: 0 = not such a
// shape, otherwise the floats of scratch (one partial per workgroup, added in index order by a second launch).
long long wgrad_cin1_plan(int B, int c_in, int n_out, int c_out, int k, int groups);
int launch_wgrad_cin1(const float* x, const float* dy, float* dw, float* ws, long long ws_elems, int B, int t_in, int n_out, int c_out, int k,
                      int stride, int pad, int dil, int accumulate, hipStream_t stream);

}  // namespace evmi
