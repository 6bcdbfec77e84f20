/*
 * evmi.h — C ABI of libevmi_hip.so: the MI355X (gfx950) native implementation of the
 * EveryVoice TTS hot path (HiFiGAN / iSTFTNet vocoder, FastSpeech2 length regulator,
 * STFT/mel front-end).
 *
 * The reference (EveryVoiceTTS/EveryVoice 0.5.0) is pure Python on PyTorch and has no FFI; this
 * header is the boundary a maintainer binds with ctypes.  Each entry point
 * cites the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - every pointer named *_dev is a DEVICE pointer the caller allocated (e.g. torch tensor
 *     .data_ptr()); *_host pointers are host memory; no ownership is transferred;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is enqueued
 *     on it and nothing synchronises unless stated;
 *   - return value 0 = EVMI_OK, otherwise an EVMI_ERR_* code; evmi_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - no torch types, no C++ types, no global state besides per-object workspaces.
 */
#ifndef EVMI_H
#define EVMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVMI_ABI_VERSION 2

enum {
  EVMI_OK = 0,
  EVMI_ERR_INVALID_ARG = 1,
  EVMI_ERR_HIP = 2,          /* a HIP runtime call or kernel launch failed */
  EVMI_ERR_NOT_READY = 3,    /* weights missing / object not finalised */
  EVMI_ERR_UNSUPPORTED = 4,  /* configuration outside what the kernels cover */
  EVMI_ERR_OOM = 5
};

/* Arithmetic the generator computes in. */
enum {
  EVMI_PREC_BF16 = 0, /* bf16 operands, fp32 accumulate on MFMA (the fast path)          */
  EVMI_PREC_F32 = 1   /* fp32 direct convolutions (exact-arithmetic path for parity work) */
};

int evmi_abi_version(void);
const char* evmi_last_error(void);
/* Fills name (<= name_len bytes, NUL terminated), compute-unit count and HBM bytes of `device`. */
int evmi_device_info(int device, char* name, int name_len, int* compute_units, int64_t* hbm_bytes);

/* ------------------------------------------------------------------------------------------
 * Length regulator — replaces everyvoice/utils/heavy.py:12-21 `expand(values, durations)` applied
 * per batch item and zero-padded (FastSpeech2 length regulator).
 *
 *   values_dev    [B, L, D]  elements of `elem_bytes` (2 or 4) bytes — copied bit-for-bit
 *   durations_dev [B, L]     int64; row i is emitted max(0, d_i) times (the reference's
 *                            int(d) truncation of float durations is done by the caller)
 *   out_dev       [B, t_max, D]  frames past an item's total are zero
 *   out_lens_dev  [B] int64  min(sum_i max(0,d_i), t_max)      (may be NULL)
 *   index_dev     [B, t_max] int32 source row per frame, -1 in the padding (may be NULL)
 * Integer/byte work: results are bit-exact.
 * ------------------------------------------------------------------------------------------ */
int evmi_length_regulate(const void* values_dev, const int64_t* durations_dev, void* out_dev,
                         int64_t* out_lens_dev, int32_t* index_dev, int B, int L, int D,
                         int t_max, int elem_bytes, void* stream);

/* Backward of the above: grad_values[b, i, :] = sum over frames t with index[b,t]==i of
 * grad_out[b, t, :] (fp32), frames summed in increasing t (deterministic). */
int evmi_length_regulate_bwd_f32(const float* grad_out_dev, const int64_t* durations_dev,
                                 float* grad_values_dev, int B, int L, int D, int t_max,
                                 void* stream);

/* ------------------------------------------------------------------------------------------
 * Generic fp32 direct convolutions on torch-native layouts (the exact-arithmetic path and the
 * building block of EVMI_PREC_F32).  Semantics of torch.nn.functional.conv1d /
 * conv_transpose1d, which is what every Conv1d/ConvTranspose1d of the reference's vocoder
 * resolves to (call sites: SURVEY.md §2.2).
 *
 *   y = [accumulate ? y : 0] + out_scale * ( conv(lrelu(x, pre_slope)) + bias + residual )
 * pre_slope = 1 disables the input activation; bias/residual may be NULL.
 *   x [B, c_in, t_in]   w [c_out, c_in/groups, k]   y [B, c_out, t_out]
 *   t_out = (t_in + 2*pad - dil*(k-1) - 1) / stride + 1
 * ------------------------------------------------------------------------------------------ */
int evmi_conv1d_f32(const float* x_dev, const float* w_dev, const float* bias_dev,
                    const float* residual_dev, float* y_dev, int B, int c_in, int t_in, int c_out,
                    int k, int stride, int pad, int dil, int groups, float pre_slope,
                    float out_scale, int accumulate, void* stream);

/*   x [B, c_in, t_in]   w [c_in, c_out, k]   y [B, c_out, t_out],  t_out = (t_in-1)*stride - 2*pad + k */
int evmi_conv_transpose1d_f32(const float* x_dev, const float* w_dev, const float* bias_dev,
                              float* y_dev, int B, int c_in, int t_in, int c_out, int k, int stride,
                              int pad, float pre_slope, void* stream);

/* ------------------------------------------------------------------------------------------
 * STFT / mel front-end — replaces the transform returned by everyvoice/utils/heavy.py:69-100
 * get_spectral_transform("mel-librosa", ...) followed by dynamic_range_compression_torch
 * (everyvoice/utils/heavy.py:39-40), i.e. Preprocessor.extract_spectral_features
 * (everyvoice/preprocessor/preprocessor.py:220-233), and extract_energy (:302-309).
 *
 *   audio_dev      [B, n_samples] fp32
 *   dft_basis_dev  [n_fft, 2*n_bins_padded] fp32: column 2b = window[k]*cos(2 pi b k / n_fft),
 *                  column 2b+1 = -window[k]*sin(...), zero beyond bin n_fft/2 (host-built constant;
 *                  n_bins_padded = n_fft/2+1 rounded up to a multiple of 16)
 *   mel_basis_dev  [n_mels, n_fft/2+1] fp32 (librosa Slaney filterbank)
 *   mel_dev        [B, n_mels, 1 + n_samples/hop] fp32 = log(max(basis @ sqrt(|STFT|^2 + 1e-9), 1e-5))
 *                  (apply_log = 0: the linear mel)
 *   energy_dev     [B, frames] L2 norm over the mel bins of mel_dev (NULL to skip)
 *   mag_dev        [B, n_fft/2+1, frames] sqrt(|STFT|^2 + 1e-9) (NULL to skip)
 * center=True, reflect padding, one-sided, win_length == n_fft, n_fft a multiple of hop.
 * ------------------------------------------------------------------------------------------ */
int evmi_mel_spectrogram_f32(const float* audio_dev, const float* dft_basis_dev,
                             const float* mel_basis_dev, float* mel_dev, float* energy_dev,
                             float* mag_dev, int B, int n_samples, int n_fft, int hop,
                             int n_bins_padded, int n_mels, int apply_log, void* stream);
/* The same transform on a RAGGED batch -- how the batched preprocessor packs several utterances into one launch (SURVEY.md 8f
 * N2): audio [B][n_samples_max] zero padded, lens [B]; item b reflects at its own end and its frames 0 .. lens[b] / hop are
 * exactly what evmi_mel_spectrogram_f32 gives for that utterance alone (later frames of its row are padding: ignore them). */
int evmi_mel_spectrogram_ragged_f32(const float* audio_dev, const int* lens_dev, const float* dft_basis_dev,
                                    const float* mel_basis_dev, float* mel_dev, float* energy_dev, float* mag_dev, int B,
                                    int n_samples_max, int n_fft, int hop, int n_bins_padded, int n_mels, int apply_log,
                                    void* stream);

/* The other spec types of get_spectral_transform (everyvoice/utils/heavy.py:59-68 "mel" = torchaudio MelSpectrogram(norm="slaney"),
 * :101-114 "linear" / "raw" = torchaudio Spectrogram(power = 2 / None), :115-118 "istft" = InverseSpectrogram): the DFT itself is
 * evmi_stft_frames_f32 + evmi_gemm_f32 with the windowed basis, which leaves real / imaginary planes [C][B*F]; this is the finishing
 * pass into torchaudio's batch-major tensors.
 *   mode 0: out [B][C][F] = re^2 + im^2      mode 4: sqrt(re^2 + im^2)      mode 1: out [B][C][F][2] = (re, im)  (complex64)
 *   mode 2: out [B][C][F] = re               (a mel projection [n_mels][B*F] back to batch-major; im_dev unused)
 *   mode 3: re_dev = complex [B][C][F][2] -> out [2C][B*F]: real rows, then imaginary rows (the inverse transform's operand) */
int evmi_spectrogram_layout_f32(int mode, const float* re_dev, const float* im_dev, float* out_dev, int B, int C, int F,
                                void* stream);

/* ------------------------------------------------------------------------------------------
 * HiFiGAN / iSTFTNet generator — replaces the forward of `hfgl.utils.HiFiGANGenerator` /
 * the generator inside `hfgl.model.HiFiGAN` (absent submodule; call sites
 * everyvoice/demo/app.py:28-33,457-459, everyvoice/base_cli/checkpoint.py:92-103;
 * hyper-parameters everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:293-415).
 * ------------------------------------------------------------------------------------------ */
#define EVMI_MAX_UPSAMPLES 8
#define EVMI_MAX_RESBLOCK_KERNELS 8
#define EVMI_MAX_DILATIONS 8

typedef struct evmi_generator_config {
  int n_mels;                   /* AudioConfig.n_mels (80)                                  */
  int upsample_initial_channel; /* 512                                                       */
  int num_upsamples;
  int upsample_rates[EVMI_MAX_UPSAMPLES];        /* [8,8,2,2]                              */
  int upsample_kernel_sizes[EVMI_MAX_UPSAMPLES]; /* [16,16,4,4]                            */
  int resblock_type;                             /* 1 or 2                                   */
  int num_kernels;
  int resblock_kernel_sizes[EVMI_MAX_RESBLOCK_KERNELS]; /* [3,7,11]                        */
  int num_dilations[EVMI_MAX_RESBLOCK_KERNELS];
  int resblock_dilations[EVMI_MAX_RESBLOCK_KERNELS][EVMI_MAX_DILATIONS]; /* [[1,3,5]]*3  */
  float lrelu_slope;      /* activation_function: original_hifigan_leaky_relu = 0.1          */
  float post_lrelu_slope; /* slope before conv_post (upstream F.leaky_relu default 0.01)     */
  int istft_layer;        /* 0: conv_post -> tanh ; 1: iSTFTNet head                         */
  int istft_n_fft;        /* 16 */
  int istft_hop;          /* 4  */
} evmi_generator_config;

typedef struct evmi_generator evmi_generator; /* opaque */

int evmi_generator_create(const evmi_generator_config* cfg, int device, evmi_generator** out);
void evmi_generator_destroy(evmi_generator* g);

/* Weights are given with weight norm already folded (w = g * v / ||v||), fp32, HOST memory, in
 * torch's native layouts under the upstream state-dict names: "conv_pre.weight" [C0,n_mels,7],
 * "conv_pre.bias", "ups.{i}.weight" [Cin,Cout,k], "ups.{i}.bias",
 * "resblocks.{n}.convs1.{m}.weight" [C,C,k], "...bias", "resblocks.{n}.convs2.{m}.*"
 * ("resblocks.{n}.convs.{m}.*" for resblock type 2), "conv_post.weight", "conv_post.bias". */
int evmi_generator_set_weight(evmi_generator* g, const char* name, const float* data_host,
                              int64_t numel);
/* Number of weight tensors the config expects; name of the i-th and its element count. */
int evmi_generator_num_weights(const evmi_generator* g);
int evmi_generator_weight_info(const evmi_generator* g, int i, char* name, int name_len,
                               int64_t* numel);
/* Re-lays out and uploads the weights for both precisions.  Must be called after all
 * evmi_generator_set_weight calls and before the first forward. */
int evmi_generator_finalize(evmi_generator* g);

/* Samples produced per mel frame (prod(upsample_rates) [* istft_hop]). */
int evmi_generator_hop(const evmi_generator* g);
/* Device bytes of scratch one forward at (B, T) needs in `precision`; the object grows its own
 * workspace to this on first use (hipMalloc outside the timed path). */
int64_t evmi_generator_workspace_bytes(const evmi_generator* g, int B, int T, int precision);

/* mel_dev [B, n_mels, T] fp32 (torch layout)  ->  wav_dev [B, 1, T*hop] fp32. */
int evmi_generator_forward(evmi_generator* g, const float* mel_dev, float* wav_dev, int B, int T,
                           int precision, void* stream);

/* Same forward with every kernel launch bracketed by HIP events on `stream` (synchronises).
 * Fills up to `cap` records; returns the number of launches through *n_out. */
typedef struct evmi_launch_record {
  char kernel[48]; /* kernel family, e.g. "conv_tc_mfma<c128,k11>" */
  char layer[48];  /* e.g. "resblocks.4.convs1.2"                   */
  float ms;        /* HIP-event duration of this launch             */
  double flops;    /* algorithmic FLOPs of this launch (2*MAC)      */
  double bytes;    /* algorithmic HBM bytes (inputs+outputs+weights once) */
} evmi_launch_record;
int evmi_generator_forward_profiled(evmi_generator* g, const float* mel_dev, float* wav_dev, int B,
                                    int T, int precision, void* stream, evmi_launch_record* records,
                                    int cap, int* n_out);

/* Algorithmic MACs per output sample of this configuration (V1: 1,199,424). */
double evmi_generator_macs_per_sample(const evmi_generator* g);

/* ------------------------------------------------------------------------------------------
 * GAN training building blocks (fp32) — what the training_step of `hfgl.model.HiFiGAN` (absent
 * submodule; call sites everyvoice/base_cli/helpers.py:36,184, SURVEY.md §3.1 hot loop) executes
 * through ATen: convolutions of generator / MPD / MSD and their gradients, activations, pooling,
 * the mel loss, weight / spectral norm, AdamW.
 *
 * Activations are channel-major "CBT": x[c][b][t] == a row-major [C][B*T] matrix (audio [B,1,T]
 * and logits have the same bytes as torch's layout).  A convolution is unfold + one GEMM
 * (everyvoice_amd/train/ops.py composes them):
 *   fwd  Y[C_out][B*T_out] = W[C_out][C_in*k] . col      dgrad  dcol = W^T . dY -> fold
 *   wgrad dW = dY . col^T
 * ------------------------------------------------------------------------------------------ */
/* Row-major C[M][N] = alpha * op(A)[M][K] . op(B)[K][N] + beta * C   (own fp32 matrix-core kernel, csrc/gemm_f32.hip; no BLAS library is linked). */
int evmi_gemm_f32(int trans_a, int trans_b, int M, int N, int K, float alpha, const float* a_dev,
                  int lda, const float* b_dev, int ldb, float beta, float* c_dev, int ldc,
                  void* stream);
/* fp32 implicit-GEMM convolution on the fp32-input matrix cores (no unfold, no library GEMM):
 *   y[co][b][to*out_stride + out_offset] (+)= bias[co] + conv(x, w)[co][b][to]   for to < n_out
 * x [c_in][B][t_in], w [c_out][c_in/groups][k], y [c_out][B][t_out_total]; out_stride 1 / offset 0 /
 * n_out = t_out_total is the plain convolution, other values place a polyphase component of a strided
 * convolution's input gradient.  c_out <= 4 and c_in == 1 (GEMV / outer-product shapes) run as direct kernels.
 * act: epilogue on conv + bias: 0 none, 1 leaky-relu(act_param), 2 SiLU, 3 ReLU, 4 tanh (not with accumulate). */
int evmi_conv1d_cbt_f32(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev,
                        float* wfrag_ws_dev, long long wfrag_ws_elems, int B, int c_in, int t_in, int c_out,
                        int t_out_total, int n_out, int k, int stride, int pad, int dil, int groups,
                        int out_stride, int out_offset, int accumulate, int act, float act_param,
                        void* stream);
/* 1 when evmi_conv1d_cbt_f32 can run this shape (degenerate ones -- rows of a few samples under a 41-tap kernel --
 * exceed its staging limits and return EVMI_ERR_UNSUPPORTED; callers use the unfold + GEMM path for those). */
int evmi_conv1d_cbt_f32_supported(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int dil, int groups);
/* Floats of the 16-byte aligned device workspace (`wfrag_ws_dev`) evmi_conv1d_cbt_f32 needs for this shape: the
 * MFMA-fragment copy of the weights (re-laid on the stream before every convolution: the weights change every
 * optimiser step), or the channel-split scratch of the few-output-channel kernel. */
long long evmi_conv1d_cbt_f32_ws_elems(int B, int c_in, int c_out, int n_out, int k, int groups);
/* Input gradient of the same convolution: dx [c_in][B][t_in] from dy [c_out][B][t_out] and the forward weights w.
 * A strided layer's gradient is min(stride, k) polyphase stride-1 convolutions of dy; they run as ONE launch (the weight
 * fragments of all phases are built straight from w, phases with fewer taps zero padded).  ws: 16-byte aligned scratch of
 * evmi_conv1d_dgrad_cbt_f32_ws_elems floats; 0 = shape not taken here (GEMV / outer-product shapes, dilated strided
 * layers): compose evmi_dgrad_weights_f32 + evmi_conv1d_cbt_f32 per phase instead. */
long long evmi_conv1d_dgrad_cbt_f32_ws_elems(int B, int c_in, int t_in, int c_out, int t_out, int k, int stride,
                                             int pad, int dil, int groups);
int evmi_conv1d_dgrad_cbt_f32(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev,
                              long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out, int k,
                              int stride, int pad, int dil, int groups, void* stream);
/* bf16-operand variants of the two calls above (same arguments, same workspaces, fp32 tensors in HBM, fp32 accumulation):
 * the kernels round both operands to bf16 on their way into v_mfma_f32_32x32x16_bf16 -- the mixed-precision ("bf16
 * autocast") counterpart of the exact fp32 path; results differ from it by the operand rounding (2^-9 relative per
 * operand).  Shapes whose staging does not fit the 16-channel K blocks of this mode run the fp32 kernels. */
int evmi_conv1d_cbt_bf16(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev,
                         float* wfrag_ws_dev, long long wfrag_ws_elems, int B, int c_in, int t_in, int c_out,
                         int t_out_total, int n_out, int k, int stride, int pad, int dil, int groups,
                         int out_stride, int out_offset, int accumulate, int act, float act_param,
                         void* stream);
/* 1 where precision="bf16" rounds this convolution's operands to bf16 (the packed kernel or the in-LDS rounding mode takes the
 * shape), 0 where the exact fp32 kernels run it (GEMV / outer-product shapes, groups narrower than the staging).  The one
 * statement of that dispatch rule: parity tests restating the arithmetic for the oracle ask here. */
int evmi_conv1d_cbt_bf16_rounds(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups);
int evmi_conv1d_dgrad_cbt_bf16(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev,
                               long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out, int k,
                               int stride, int pad, int dil, int groups, void* stream);
/* Packed-input bf16 convolution (csrc/conv_cbt_bf16_pk.hip), the kernel precision="bf16" training and inference use wherever it
 * takes the shape (at least 8 input channels per group, more than 4 output channels per group): x is first re-laid as
 * 16-byte units of 8 channels per position with the zero padding materialised (one pass), so that every operand load of the
 * implicit GEMM is a full 1 KB LDS-direct load and a B fragment one ds_read_b128.  Same arguments as evmi_conv1d_cbt_f32 /
 * evmi_conv1d_dgrad_cbt_f32; `ws_dev`: 16-byte aligned scratch of the matching *_ws_elems floats (packed input + bf16 weight
 * fragments + K-block table); *_ws_elems == 0 means "shape not taken here": use the calls above. */
long long evmi_conv1d_cbt_bf16pk_ws_elems(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad,
                                          int dil, int groups);
int evmi_conv1d_cbt_bf16pk(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev,
                           long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out_total, int n_out, int k,
                           int stride, int pad, int dil, int groups, int out_stride, int out_offset, int accumulate,
                           int act, float act_param, void* stream);
long long evmi_conv1d_dgrad_cbt_bf16pk_ws_elems(int B, int c_in, int t_in, int c_out, int t_out, int k, int stride,
                                                int pad, int dil, int groups);
int evmi_conv1d_dgrad_cbt_bf16pk(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev,
                                 long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out, int k, int stride,
                                 int pad, int dil, int groups, void* stream);
/* Weight gradient with bf16 operands (csrc/conv_wgrad_bf16_pk.hip): dy and x packed as above, the contraction over
 * (item, position) fed to the matrix cores through gfx950's transposing LDS read (ds_read_b64_tr_b16), split-K with a
 * fixed-order reduction (bitwise reproducible).  Takes groups of at least 32 x 32 channels (cin_g * cout_g >= 4096);
 * *_ws_elems == 0 otherwise (the fp32 implicit-GEMM / GEMM weight gradient below takes those). */
long long evmi_conv1d_wgrad_cbt_bf16pk_ws_elems(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride,
                                                int pad, int dil, int groups);
int evmi_conv1d_wgrad_cbt_bf16pk(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev,
                                 long long ws_elems, int B, int c_in, int t_in, int c_out, int n_out, int k, int stride,
                                 int pad, int dil, int groups, int accumulate, void* stream);
/* evmi_conv1d_dgrad_cbt_bf16pk in two steps (same arguments and workspace for both): stage 1 packs dy into the head of ws_dev and
 * prepares the weight fragments behind it (one launch), stage 2 runs the convolution on them.  Between them a training step forks its
 * weight-gradient stream: a pointwise layer's weight gradient reads the packed dy (evmi_conv1d_wgrad_cbt_bf16pk_prepacked) beside the
 * input gradient.  Stage 3: fragments + convolution on a packed dy that a producer's epilogue left at the head of ws_dev
 * (evmi_conv1d_dgrad_cbt_bf16pk_ffn_down). */
int evmi_conv1d_dgrad_cbt_bf16pk_staged(int stage, const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev, long long ws_elems,
                                        int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil, int groups,
                                        void* stream);
/* The middle of a feed-forward block, dense2(dropout(silu(a), p)), and its backward with the activation and the dropout mask applied while
 * the operands are PACKED -- neither dropout(silu(a)) nor dropout(ds) * silu'(a) is ever stored in fp32 (reference: the Conformer
 * feed-forward module, Linear -> SiLU -> Dropout -> Linear; arithmetic and mask stream of evmi_dropout_fused_f32 modes 2 / 3:
 * seed_value + *seed_base_dev, element index = the element's index in the tensor):
 *   evmi_conv1d_cbt_bf16pk_silu_dropout:              y = conv(dropout(silu(x), p)) + bias   (stride 1, no dilation, one group; the head of
 *                                                     ws is the packed activated input, what the layer's weight gradient reads again)
 *   evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout: evmi_conv1d_dgrad_cbt_bf16pk_staged on dy = dropout(ds, p) * silu'(pre): stage 1
 *                                                     packs that product into the head of ws, stage 2 runs the convolution on it. */
int evmi_conv1d_cbt_bf16pk_silu_dropout(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev,
                                        long long ws_elems, int B, int c_in, int t_in, int c_out, int k, int pad, float p,
                                        unsigned long long seed_value, const unsigned long long* seed_base_dev, void* stream);
int evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout(int stage, const float* ds_dev, const float* pre_dev, float p, unsigned long long seed_value,
                                                     const unsigned long long* seed_base_dev, const float* w_dev, float* dx_dev, float* ws_dev,
                                                     long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out, int k, int stride,
                                                     int pad, int dil, int groups, void* stream);
/* A Conformer sub-layer's last pointwise layer with the residual add + dropout behind it in the epilogue (the reference:
 * torchaudio-style Conformer sub-layers `x + dropout(sublayer(x))`, FastSpeech2_lightning -- absent submodule), and the
 * matching pack of the backward:
 *   evmi_conv1d_cbt_bf16pk_resdrop:              y = residual + out_scale * dropout(conv(in) + bias, out_p); in_mode 0: in = x, 1: in =
 *                                                dropout(silu(x), in_p), 2: packed input already at the head of ws (x not read), 3: and
 *                                                the weight fragments behind it (evmi_layernorm_pack_bf16pk_w)
 *   evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout: evmi_conv1d_dgrad_cbt_bf16pk_staged on dz = scale * dropout(dy, p) (stage 1 packs it)
 * Mask streams of evmi_dropout_fused_f32 (seed + *seed_base_dev, element index = index in the tensor). */
int evmi_conv1d_cbt_bf16pk_resdrop(int in_mode, const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                                   float* y_dev, float* ws_dev, long long ws_elems, int B, int c_in, int t_in, int c_out, float in_p,
                                   unsigned long long in_seed, float out_p, unsigned long long out_seed, float out_scale,
                                   const unsigned long long* seed_base_dev, void* stream);
int evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout(int stage, const float* dy_dev, float p, unsigned long long seed_value,
                                                const unsigned long long* seed_base_dev, float scale, const float* w_dev, float* dx_dev,
                                                float* ws_dev, long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out, int k,
                                                int stride, int pad, int dil, int groups, void* stream);
/* A feed-forward block's wide middle tensor kept packed in both directions (FastSpeech2_lightning's Conformer feed-forward modules
 * Linear -> SiLU -> Dropout -> Linear -- absent submodule; pointwise layers c_in -> c_mid -> c_out on tight items):
 *   (a_pk_dev NULL in evmi_conv1d_cbt_bf16pk_ffn_up: inference -- only the activated tensor is written; p = 0: no mask, no hash)
 *   evmi_conv1d_cbt_bf16pk_ffn_up:          the first layer on the packed input at the head of ws; a = conv + bias is written as
 *                                           a_pk [c_mid / 8][B * t] 16-byte units (bf16) and dropout(silu(a), p) as the packed input
 *                                           at the head of next_ws, the SECOND layer's workspace (run it with ..._prepacked /
 *                                           ..._resdrop in_mode 2)
 *   evmi_conv1d_dgrad_cbt_bf16pk_ffn_down:  the second layer's input gradient on the packed dz at the head of ws (stage 1 of
 *                                           ..._staged[_dropout]); dropout(ds, p) * silu'(a_pk) is written as the packed dy at the
 *                                           head of next_ws, the workspace of the FIRST layer's input gradient (run ..._staged stage 2)
 * Mask streams of evmi_conv1d_cbt_bf16pk_silu_dropout / ..._staged_silu_dropout; silu' is taken at bf16(a). */
int evmi_conv1d_cbt_bf16pk_ffn_up(const float* w_dev, const float* bias_dev, float* ws_dev, long long ws_elems, void* a_pk_dev,
                                  float* next_ws_dev, long long next_ws_elems, int B, int c_in, int t, int c_mid, int c_out, float p,
                                  unsigned long long seed_value, const unsigned long long* seed_base_dev, int fragments_ready, void* stream);
int evmi_conv1d_dgrad_cbt_bf16pk_ffn_down(const float* w_dev, float* ws_dev, long long ws_elems, const void* a_pk_dev, float* next_ws_dev,
                                          long long next_ws_elems, int B, int c_in, int t, int c_mid, int c_out, float p,
                                          unsigned long long seed_value, const unsigned long long* seed_base_dev, void* stream);
/* LayerNorm in front of a pointwise layer written as that layer's packed input, and the layer on an input already packed in the head
 * of ws (the Conformer's LayerNorm -> Linear pairs; layers of evmi_conv1d_bf16pk_shares_packed, 128 or 256 input channels): the
 * normalised tensor is never stored in fp32.  ws: evmi_conv1d_cbt_bf16pk_ws_elems floats of the layer (k = 1, stride 1, no padding). */
int evmi_layernorm_pack_bf16pk(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* ws_dev, long long ws_elems, int B,
                               int c_in, int t_in, int c_out, float eps, void* stream);
/* ... with the weight fragments of the layer(s) behind the LayerNorm prepared by the SAME launch: w_dev [c_out][c_in] into ws_dev (run the
 * layer with fragments_ready = 1) and, for a feed-forward block, w2_dev [c_out2][c_out] into ws2_dev, the second layer's workspace
 * (run it with evmi_conv1d_cbt_bf16pk_resdrop in_mode 3); w2_dev NULL: the first layer only. */
int evmi_layernorm_pack_bf16pk_w(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* ws_dev, long long ws_elems, int B,
                                 int c_in, int t_in, int c_out, float eps, const float* w_dev, const float* w2_dev, float* ws2_dev,
                                 long long ws2_elems, int c_out2, void* stream);
/* fragments_ready: the weight fragments are in ws already (evmi_layernorm_pack_bf16pk_w): the call is the convolution launch alone. */
int evmi_conv1d_cbt_bf16pk_prepacked(const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev, long long ws_elems, int B,
                                     int c_in, int t_in, int c_out, int act, float act_param, int fragments_ready, void* stream);
/* The packed bf16 convolution kernels with a residual block's neighbours fused in (no separate activation / add passes, no
 * activated copies in HBM) -- the training-side counterpart of SURVEY.md 8b's evmi_resblock1_fused_{fwd,bwd}:
 *   forward   y = act(conv(leaky_relu(x, pre_slope)) + bias) + residual
 *   dgrad     dx = conv_input_grad(dy * lrelu'(dy_mask)) * lrelu'(dx_mask) + residual        (lrelu'(m) = m > 0 ? 1 : slope;
 *             dy_mask = the output of the activation behind the convolution, dx_mask = the input of the one in front of it)
 *   wgrad     dw (+)= conv_weight_grad(leaky_relu(x, x_pre_slope), dy * lrelu'(dy_mask))
 * Masks / residual may be NULL, slopes 1 = none; shapes and workspaces as the unfused entry points. */
int evmi_conv1d_cbt_bf16pk_fused(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, float* ws_dev,
                                 long long ws_elems, int B, int c_in, int t_in, int c_out, int t_out_total, int n_out, int k,
                                 int stride, int pad, int dil, int groups, int act, float act_param, float pre_slope,
                                 const float* residual_dev, void* stream);
int evmi_conv1d_dgrad_cbt_bf16pk_fused(const float* dy_dev, const float* w_dev, float* dx_dev, float* ws_dev, long long ws_elems,
                                       int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil,
                                       int groups, float pre_slope, const float* dy_mask_dev, float dy_mask_slope,
                                       const float* dx_mask_dev, float dx_mask_slope, const float* residual_dev, void* stream);
/* Pointwise stride-1 layers (k = 1, no padding, one group) whose B * t is a multiple of 64 (evmi_conv1d_bf16pk_shares_packed(...) == 1:
 * the rows of the tight packing end on a K step of the weight gradient) pack their operands ONCE: the
 * forward call's packed input (the head of the workspace it was given) and the input-gradient call's packed dy (the head of ITS
 * workspace) are in the layout the weight gradient reads (conv_pk_common.h), so a caller that keeps those two
 * workspaces alive hands them over here and no operand is packed a second time (replaces the x / dy re-layout that
 * torch.nn.functional.conv1d's weight gradient would do internally; reference call sites as evmi_conv1d_wgrad_cbt_bf16pk).
 * Either packed pointer may be NULL: that operand is then packed from its fp32 tensor as usual. */
int evmi_conv1d_wgrad_cbt_bf16pk_prepacked(const float* x_dev, const void* x_packed_dev, const float* dy_dev, const void* dy_packed_dev,
                                           float* dw_dev, float* ws_dev, long long ws_elems, int B, int c_in, int t_in, int c_out, int n_out,
                                           int k, int stride, int pad, int dil, int groups, int accumulate, void* stream);
int evmi_conv1d_bf16pk_shares_packed(int B, int t, int k, int stride, int pad, int dil, int groups);
int evmi_conv1d_wgrad_cbt_bf16pk_fused(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev, long long ws_elems,
                                       int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil,
                                       int groups, int accumulate, float x_pre_slope, const float* dy_mask_dev,
                                       float dy_mask_slope, void* stream);
/* Which kernel instantiation the planner of the packed bf16 kernels picks for a shape (what tests and profiles name):
 *   evmi_conv1d_cbt_bf16pk_plan / evmi_conv1d_dgrad_cbt_bf16pk_plan: tile index 0..8 = conv_pk_kernel<128,128> <64,128> <64,64>
 *   <32,128> <64,256> <32,256> <128,256>, 7 / 8 = the eight-wave <128,256> / <128,128>, + 16 * the split-K factor when the contraction is split over workgroups (ksplit > 1);
 *   evmi_conv1d_wgrad_cbt_bf16pk_plan: taps per workgroup of wgrad_pk_kernel<4 | 8>, + 16 * the number of K splits.
 * -1 = shape not taken by that kernel (the matching *_ws_elems is 0).  Replaces nothing in the reference (it has no kernels:
 * torch picks its own through F.conv1d, e.g. everyvoice/model/utils.py:10-45); exists so that parity tests can assert that the
 * instantiation a profile shows is the one a test compared with torch. */
int evmi_conv1d_cbt_bf16pk_plan(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil, int groups);
int evmi_conv1d_dgrad_cbt_bf16pk_plan(int B, int c_in, int t_in, int c_out, int t_out, int k, int stride, int pad, int dil,
                                      int groups);
int evmi_conv1d_wgrad_cbt_bf16pk_plan(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride, int pad, int dil,
                                      int groups);
/* ---- Training on TIME-MAJOR bf16 activations (csrc/train_tm.hip): the generator's residual stacks (upstream ResBlock1 / MRF,
 * jik876 hifi-gan models.py; the reference reaches them through hfgl.model.HiFiGAN.training_step) run their
 * training forward and input gradients on the inference convolution kernels, with no layout pass between two convolutions.
 * A TM tensor: bf16 [B][Tp][C], the T valid rows of an item at rows [PL, PL + T), zero rows around them; pointers address row 0 of
 * item 0; the allocation keeps >= 64 zero rows in front and >= 256 behind (the weight gradient walks the flat row index).
 *   evmi_conv_tc_supported      1 when a time-major kernel takes (c_in, c_out, ks, dil)
 *   evmi_conv_tc_relayout_f32   w fp32 [c_out][c_in][ks] -> the bf16 tile layout the kernel streams; transpose = 1: the weights of
 *                               the INPUT GRADIENT of the convolution with w [c_in][c_out][ks] (channels swapped, taps reversed)
 *   evmi_conv_tc_tm_bf16        out = post( out_scale * ( residual + maskfac * ( bias + conv(pre(x)) ) ) ), "same" padding,
 *                               pre / post = leaky ReLU with the given slopes (1 = none), maskfac = mask > 0 ? 1 : mask_slope
 *                               (mask NULL: 1) -- the activation backward of an input-gradient launch
 *   evmi_conv1d_wgrad_tm_bf16   dw[co][ci][j] (+)= sum_rows dy[row][co] * x[row + j*dil - pad][ci] over all `rows` = B * Tp flat rows
 *                               (split-K, fixed summation order); ws: evmi_conv1d_wgrad_tm_bf16_ws_elems floats (-1: shape not taken)
 *   evmi_tm_colsum_bf16         db[c] (+)= sum_rows dy[row][c] (bias gradient); ws: evmi_tm_colsum_bf16_ws_elems floats
 *   evmi_cbt_f32_to_tm_bf16     fp32 [C][B][T] -> TM, v = leaky_relu(x, slope) * scale;  evmi_tm_bf16_to_cbt_f32: scale * (a + b + c) -> fp32
 *   evmi_tm_lrelu_bf16          elementwise over a whole TM buffer (zero rows stay zero) */
int evmi_conv_tc_supported(int c_in, int c_out, int ks, int dil);
int evmi_conv_tc_relayout_f32(const float* w_dev, void* dst_bf16_dev, int c_in, int c_out, int ks, int dil, int transpose, void* stream);
/* ... all C x C layers of a residual stack in one launch: table_dev[l] = {source offset (floats from w_base_dev), destination offset
 * (bf16 elements from dst_base), ks, BM, KC, wlayout (evmi_conv_tc_tile_layout), transpose, 0} as 8 int64 each. */
int evmi_conv_tc_tile_layout(int c_in, int c_out, int ks, int dil, int* bm, int* kc, int* wlayout);
int evmi_conv_tc_relayout_batched_f32(const float* w_base_dev, void* dst_base_bf16_dev, const long long* table_dev, int n_layers, int C,
                                      int ks_max, void* stream);
int evmi_conv_tc_tm_bf16(const void* x_tm, const void* w_laid, const float* bias_dev, const void* res_tm, const void* mask_tm, void* out_tm,
                         int B, int T, int Tp, int PL, int c_in, int c_out, int ks, int dil, float pre_slope, float post_slope,
                         float mask_slope, float out_scale, void* stream);
long long evmi_conv1d_wgrad_tm_bf16_ws_elems(long long rows, int c_in, int c_out, int k, int dil);
int evmi_conv1d_wgrad_tm_bf16(const void* x_tm, const void* dy_tm, float* dw_dev, float* ws_dev, long long ws_elems, long long rows,
                              int c_in, int c_out, int k, int pad, int dil, int accumulate, void* stream);
long long evmi_tm_colsum_bf16_ws_elems(long long rows, int C);
int evmi_tm_colsum_bf16(const void* dy_tm, float* db_dev, float* ws_dev, long long ws_elems, long long rows, int C, int accumulate,
                        void* stream);
/* ... the bias gradients of n (<= 24) tensors of one shape in one launch pair (a whole residual stack); ws: n x the single call's. */
int evmi_tm_colsum_batch_bf16(int n, const void* const* dy_tm, float* const* db_dev, float* ws_dev, long long ws_elems, long long rows, int C, int accumulate,
                              void* stream);
int evmi_cbt_f32_to_tm_bf16(const float* x_dev, void* tm_dev, int C, int B, int T, int Tp, int PL, float slope, float scale, void* stream);
int evmi_tm_bf16_to_cbt_f32(const void* a_tm, const void* b_tm, const void* c_tm, float* out_dev, int C, int B, int T, int Tp, int PL,
                            float scale, void* stream);
int evmi_tm_lrelu_bf16(const void* x_tm, void* y_tm, long long n_elems, float slope, void* stream);
/* ---- Discriminator chains on FLAT PACKED bf16 tensors (csrc/conv_cbt_bf16_pk.hip, conv_wgrad_bf16_pk.hip, disc_chain.hip; host side
 * train/disc_chain.py).  MPD / MSD (jik876 hifi-gan models.py DiscriminatorP / DiscriminatorS -- the reference reaches them through
 * hfgl.model.HiFiGAN.training_step) are chains of strided / grouped convolutions with a leaky ReLU behind each; with
 * precision "bf16" every activation between two of their layers is kept ONCE, in the layout the matrix-core kernels load:
 *   flat packed tensor = bf16 [C / 8 octet rows][units], a unit = the 8 channels of one position (16 bytes); rows `plane` units apart;
 *   n_items items laid end to end, T units apart, the first `valid` units of an item are data; the gap behind every item, >= 64 units
 *   in front of unit 0 and >= 2048 behind the last item are ZERO and stay zero (the buffer is zeroed once, kernels write valid units
 *   only).  The gap is the next convolution's zero padding (right side of this item = left side of the next one), so a convolution
 *   runs over the whole row as over ONE item: T_in = stride * T_out, T_in - valid_in >= max(pad, taps read past the item's end).
 *   Pointers address unit 0 of row 0.
 *   evmi_conv_pkflat_fwd      y = act(bias + conv(x)); column v of the flat output (item v / Tc, position v % Tc, Tc = T_x / stride) is
 *                             stored at item * T_store + position when position < valid, dropped otherwise
 *   evmi_conv_pkflat_dgrad    dx = conv_input_grad(dy) [+ fm_scale * sign(mask - fm)] * (mask > 0 ? 1 : mask_slope): the polyphase input
 *                             gradient over the flat dy (items T_dy apart; dx items stride * T_dy apart in the compute geometry, stored
 *                             T_store apart); mask = the layer's own activated input of the forward pass (a flat packed tensor of dx's
 *                             shape, items T_mask apart): leaky-ReLU backward; fm = the same activation of the real waveform: the
 *                             feature-matching gradient (NULL: none)
 *   evmi_conv_pkflat_wgrad    dw[co][ci][j] (+)= sum_f dy[co][f] * x[ci][f * stride + j * dil - pad] over the flat index f of dy (groups
 *                             narrower than 32 channels run block-diagonally on the same kernel); ws: _wgrad_ws_elems floats
 *   evmi_conv_pkflat_fragments  the bf16 matrix-core fragments of the weights of any number of (layer, direction) pairs, 32 per launch;
 *                             they depend on the layer alone, so one buffer (evmi_conv_pkflat_frag_elems floats) serves every call of the
 *                             layer while its weights stand: wf_dev of _fwd / _dgrad
 *   evmi_conv_pkflat_tab      once per (call shape, workspace): the call's static K-block offset table at the head of its
 *                             workspace (evmi_conv_pkflat_ws_elems floats, 0: shape not taken; the rest: split-K partial tiles, added in
 *                             split order by a reduce pass)
 *   evmi_disc_first_*         the one-input-channel first layer on the waveform itself: item (b, c) of the period view is
 *                             x[h] = audio[b][h * period + c], reflected past the end (period 1: the waveform); forward -> flat packed,
 *                             weight / bias gradient from a flat packed dy, input gradient -> fp32 [n_items][H]
 *   evmi_disc_post_*          the one-output-channel logit layer: flat packed x -> fp32 logits [n_items][n]; dlogits -> flat packed dx
 *                             (with the mask / feature-matching tail); weight gradient
 *   evmi_pkflat_absdiff       out[0] += sum_l scale_l * sum |a_l - b_l| over pairs of whole buffers (gaps are zero in both), or of two item
 *                             ranges of one tensor row by row (the generator step's [real | generated] batch)
 *   evmi_pkflat_rowsum        db_l[c] += sum of row c of dy_l (bias gradients of a whole chain in one launch pair) */
typedef struct {
  int mode; /* 0: forward, 1: input gradient */
  int c_in, c_out, k, stride, groups;
  const float* w; /* [c_out][c_in / groups][k] fp32 */
  void* wf;       /* fragment buffer of evmi_conv_pkflat_frag_elems floats, 16-byte aligned */
  long long wf_elems;
} evmi_pkflat_job;
typedef struct {
  const void* a;
  const void* b;
  long long units; /* 16-byte units compared per row */
  long long plane; /* units between two rows (rows > 1) */
  int rows;        /* 1: a / b are whole buffers; C / 8: two item ranges of packed tensors, row by row */
  float scale;
} evmi_pkflat_pair;
typedef struct {
  const void* dy;
  long long plane, units; /* units per row that are summed (n_items * T) */
  int C;
  float* db;
} evmi_pkflat_rows;
long long evmi_conv_pkflat_ws_elems(int mode, int n_items, int T, int c_in, int c_out, int k, int stride, int pad, int dil, int groups);
int evmi_conv_pkflat_plan(int mode, int n_items, int T, int c_in, int c_out, int k, int stride, int pad, int dil, int groups);
int evmi_conv_pkflat_tab(int mode, int n_items, int T, int c_in, int c_out, int k, int stride, int pad, int dil, int groups, float* ws_dev,
                         long long ws_elems, void* stream);
long long evmi_conv_pkflat_frag_elems(int mode, int c_in, int c_out, int k, int stride, int groups);
int evmi_conv_pkflat_fragments(int n_jobs, const evmi_pkflat_job* jobs, void* stream);
int evmi_conv_pkflat_fwd(const void* x_pk, long long x_plane, const void* wf_dev, const float* bias_dev, void* y_pk, long long y_plane,
                         float* ws_dev, long long ws_elems, int n_items, int T_x, int c_in, int c_out, int k, int stride, int pad,
                         int dil, int groups, int valid, int T_store, int act, float act_param, void* stream);
int evmi_conv_pkflat_dgrad(const void* dy_pk, long long dy_plane, const void* wf_dev, void* dx_pk, long long dx_plane, float* ws_dev,
                           long long ws_elems, int n_items, int T_dy, int c_in, int c_out, int k, int stride, int pad, int dil,
                           int groups, int valid, int T_store, const void* mask_pk, const void* fm_pk, long long mask_plane, int T_mask,
                           float mask_slope, float fm_scale, void* stream);
long long evmi_conv_pkflat_wgrad_ws_elems(int n_items, int T_dy, int c_in, int c_out, int k, int stride, int dil, int groups);
int evmi_conv_pkflat_wgrad(const void* x_pk, long long x_plane, const void* dy_pk, long long dy_plane, float* dw_dev, float* ws_dev,
                           long long ws_elems, int n_items, int T_dy, int c_in, int c_out, int k, int stride, int pad, int dil, int groups,
                           int accumulate, void* stream);
int evmi_pkflat_zero(void* buf, long long n_units, void* stream);
int evmi_disc_first_fwd(const float* audio_dev, int n_audio, int t_audio, int period, const float* w_dev, const float* bias_dev, void* y_pk,
                        long long y_plane, int T_store, int n_out, int c_out, int k, int stride, int pad, float slope, void* stream);
long long evmi_disc_first_wgrad_ws_elems(int n_items, int n_out, int c_out, int k);
int evmi_disc_first_wgrad(const float* audio_dev, int n_audio, int t_audio, int period, const void* dy_pk, long long dy_plane, int T_dy, int n_out,
                          float* dw_dev, float* db_dev, float* ws_dev, long long ws_elems, int c_out, int k, int stride, int pad, int accumulate,
                          void* stream);
int evmi_disc_first_dgrad(const void* dy_pk, long long dy_plane, int T_dy, int n_out, const float* w_dev, float* dx_dev, int n_items, int H, int c_out,
                          int k, int stride, int pad, void* stream);
long long evmi_disc_post_fwd_ws_elems(int n_items, int n, int C);
int evmi_disc_post_fwd(const void* x_pk, long long x_plane, int T_x, int n_items, int n, const float* w_dev, const float* bias_dev, float* logits_dev,
                       float* ws_dev, long long ws_elems, int C, int k, int pad, void* stream);
int evmi_disc_post_dgrad(const float* dlogits_dev, const float* w_dev, void* dx_pk, long long dx_plane, int T_store, int n_items, int n, int C, int k,
                         int pad, const void* mask_pk, const void* fm_pk, long long mask_plane, int T_mask, float mask_slope, float fm_scale,
                         void* stream);
long long evmi_disc_post_wgrad_ws_elems(int n_items, int n, int C, int k);
int evmi_disc_post_wgrad(const void* x_pk, long long x_plane, int T_x, int n_items, int n, const float* dlogits_dev, float* dw_dev, float* ws_dev,
                         long long ws_elems, int C, int k, int pad, int accumulate, void* stream);
long long evmi_pkflat_absdiff_ws_elems(int n_pairs);
int evmi_pkflat_absdiff(int n_pairs, const evmi_pkflat_pair* pairs, float* out_dev, float* ws_dev, long long ws_elems, void* stream);
long long evmi_pkflat_rowsum_ws_elems(int n_jobs, const evmi_pkflat_rows* jobs);
int evmi_pkflat_rowsum(int n_jobs, const evmi_pkflat_rows* jobs, float* ws_dev, long long ws_elems, void* stream);
/* Weight gradient of the same convolution as an implicit GEMM on the fp32 matrix cores (no unfold):
 *   dw[co][ci][j] (+)= sum_{b,to} dy[co][b][to] * x[ci][b][to*stride + j*dil - pad]
 * x [c_in][B][t_in], dy [c_out][B][n_out], dw [c_out][c_in/groups][k]; `ws_dev`: 16-byte aligned scratch of
 * evmi_conv1d_wgrad_cbt_f32_ws_elems floats (padded operand copies and split-K partial tiles; 0 = shape not supported,
 * use unfold + evmi_gemm_f32). */
long long evmi_conv1d_wgrad_cbt_f32_ws_elems(int B, int c_in, int t_in, int c_out, int n_out, int k, int stride,
                                             int pad, int dil, int groups);
int evmi_conv1d_wgrad_cbt_f32(const float* x_dev, const float* dy_dev, float* dw_dev, float* ws_dev,
                              long long ws_elems, int B, int c_in, int t_in, int c_out, int n_out, int k,
                              int stride, int pad, int dil, int groups, int accumulate, void* stream);
/* Weights of the stride-1 convolution that yields phase `phi` of a convolution's input gradient:
 * wt[c_in][c_out/groups][M], M = ceil((k - phi) / stride), wt[g*cin_g+ci][co][m] = w[g*cout_g+co][ci][phi + stride*(M-1-m)]. */
int evmi_dgrad_weights_f32(const float* w_dev, float* wt_dev, int c_in, int c_out, int k, int groups,
                           int stride, int phi, void* stream);
/* `batch` such GEMMs at fixed element strides (the groups of a grouped convolution). */
int evmi_gemm_batched_f32(int trans_a, int trans_b, int M, int N, int K, float alpha,
                          const float* a_dev, int lda, long long stride_a, const float* b_dev, int ldb,
                          long long stride_b, float beta, float* c_dev, int ldc, long long stride_c,
                          int batch, void* stream);
/* col[(c*k + j)][b][to] = x[c][b][to*stride + j*dil - pad] (0 outside);  fold is its adjoint. */
int evmi_unfold_cbt_f32(const float* x_dev, float* col_dev, int C, int B, int t_in, int t_out, int k,
                        int stride, int pad, int dil, void* stream);
int evmi_fold_cbt_f32(const float* dcol_dev, float* dx_dev, int C, int B, int t_in, int t_out, int k,
                      int stride, int pad, int dil, int accumulate, void* stream);
int evmi_bias_add_rows_f32(float* y_dev, const float* bias_dev, int rows, long long n_per_row,
                           void* stream);
/* Backward of leaky_relu(conv(x)) in one pass over dy [rows][n]: dpre = dy * (y > 0 ? 1 : slope) (the convolution's output
 * gradient) and db[r] (+)= sum_n dpre[r][n] (its bias gradient), fixed summation order. */
int evmi_lrelu_bwd_rowsum_f32(const float* dy_dev, const float* y_dev, float* dpre_dev, float* db_dev, int rows,
                              long long n_per_row, float slope, int accumulate, void* stream);
/* out[r] (+)= scale * sum_n f;  mode 0: a, 1: a*b, 2: a*a  (bias gradients, per-row dots). */
int evmi_row_reduce_f32(int mode, const float* a_dev, const float* b_dev, float* out_dev, int rows,
                        long long n_per_row, float scale, int accumulate, void* stream);
/* Elementwise ops (op codes documented in csrc/train_ops.hip: leaky-relu / tanh and their
 * derivatives, axpby, products, L1 / LSGAN loss derivatives, log-clamp, the mel-loss chain, SiLU / ReLU / GLU). */
int evmi_elementwise_f32(int op, const float* a_dev, const float* b_dev, const float* c_dev,
                         float* y_dev, long long n, float p0, float p1, void* stream);
/* out[0] (+)= scale * sum f;  mode 0: |a-b|, 1: (a-p)^2, 2: a   (fixed order: reproducible). */
int evmi_scalar_reduce_f32(int mode, const float* a_dev, const float* b_dev, float* out_dev,
                           long long n, float scale, float p, int accumulate, void* stream);
/* AvgPool1d(4, 2, padding=2) on [rows][t_in] (MSD) and its adjoint. */
int evmi_avgpool4s2_f32(const float* x_dev, float* y_dev, long long rows, int t_in, int backward,
                        void* stream);
/* MPD view: reflect-pad [B][T] on the right to a multiple of `period`, then [B*period][T'/period]. */
int evmi_period_view_f32(const float* x_dev, float* x2_dev, int B, int T, int period, int backward,
                         void* stream);
/* frames[k][b][f] = reflect_pad(x)[b][f*hop + k] -> [n_fft][B*(1 + T/hop)] (centred STFT) / adjoint. */
int evmi_stft_frames_f32(const float* x_dev, float* frames_dev, int B, int T, int n_fft, int hop,
                         int backward, void* stream);
/* torch.nn.utils.weight_norm (dim 0): w = g * v / ||v|| per row, and its backward. */
int evmi_weight_norm_fwd_f32(const float* g_dev, const float* v_dev, float* w_dev, float* norm_dev,
                             int rows, int n_per_row, void* stream);
int evmi_weight_norm_bwd_f32(const float* g_dev, const float* v_dev, const float* norm_dev,
                             const float* dw_dev, float* dg_dev, float* dv_dev, int rows,
                             int n_per_row, void* stream);
/* torch.nn.utils.weight_norm of all weight-normed convolutions of one optimiser in ONE launch (w = g v / ||v|| per row), and
 * the matching backward (dg, dv from the gradient of w; the gradient sink is zeroed behind the read).  flat / grad: the
 * optimiser's flat parameter / gradient buffers; eff / dw_eff: effective weights and their gradient sink (same layout);
 * table [6][n_layers + 1] int64: row_start prefix sums, n_per_row, offsets of g, v (flat), w (eff), norms; rows [row_lo, row_hi)
 * of the concatenated row list (a gradient bucket = a contiguous layer range).  Same arithmetic as the per-layer calls. */
int evmi_weight_norm_fwd_batched_f32(const float* flat_dev, float* eff_dev, float* norms_dev, const long long* table_dev,
                                     int n_layers, long long row_lo, long long row_hi, void* stream);
int evmi_weight_norm_bwd_batched_f32(const float* flat_dev, float* grad_dev, const float* norms_dev, float* dw_eff_dev,
                                     const long long* table_dev, int n_layers, long long row_lo, long long row_hi,
                                     void* stream);
/* iSTFTNet head in training (generator with `istft_layer: true`, everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:387-392;
 * upstream rishikksh20/iSTFTNet-pytorch): between conv_post and the inverse STFT.  a [2H][n]: H log-magnitude rows then H phase
 * rows; s [2H][n]: real rows then imaginary rows of exp(a) * exp(i sin(b)).  The inverse STFT itself is a transposed
 * convolution with the windowed inverse-DFT basis (evmi_gemm_f32 + evmi_fold_cbt_f32) and the window-envelope division. */
int evmi_istft_polar_f32(const float* a_dev, float* s_dev, int H, long long n, void* stream);
int evmi_istft_polar_bwd_f32(const float* a_dev, const float* ds_dev, float* da_dev, int H, long long n, void* stream);
/* torch.nn.ReflectionPad1d((1, 0)) on `rows` rows of T samples -> T + 1 (backward = 1: its adjoint, src = dy [rows][T+1]). */
int evmi_reflect_pad_left1_f32(const float* src_dev, float* dst_dev, long long rows, int T, int backward, void* stream);
/* y = x / max(||x||, eps) (spectral norm's power iteration). */
int evmi_normalize_vec_f32(const float* x_dev, float* y_dev, int n, float eps, void* stream);
/* One optimiser step on a flat parameter buffer: kind 0 torch.optim.AdamW, 1 torch.optim.Adam (L2 weight decay), 2
 * torch.optim.RMSprop (beta1 = alpha; m unused) -- the optimiser union of the reference's training config
 * (everyvoice/.schema/everyvoice-spec-to-wav-0.5.json:434-622).  `step` is the 1-based step number, read from
 * step_dev[0] instead when given (HIP-graph replay); clip > 0 clamps the updated parameters to +-clip (gan_type "wgan",
 * wgan_clip_value, same schema :573-605). */
int evmi_optimizer_step_f32(int kind, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n, float lr,
                            float beta1, float beta2, float eps, float weight_decay, int step, const int* step_dev,
                            float clip, void* stream);
/* The same step with the learning rate read from lr_dev[0] and the step number from step_dev[0]: a scheduled rate (Noam,
 * everyvoice/config/shared_types.py:311-320) that the host stores on the device before each step, so that a captured HIP graph of
 * the step replays with the current value. */
int evmi_optimizer_step_lrdev_f32(int kind, float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n,
                                  const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, const int* step_dev,
                                  float clip, void* stream);
/* dst_dev[0..n) = values_host[0..n) (n <= 8) / dst_dev[0] = value, the values travelling in the kernel's argument block (read when
 * the call is made, ordered on the stream like any launch): per-step scalars of a captured training step (learning rate, counts,
 * the dropout seed base) without a host buffer that would have to outlive the copy. */
int evmi_store_f32(float* dst_dev, int n, const float* values_host, void* stream);
int evmi_store_u64(unsigned long long* dst_dev, unsigned long long value, void* stream);
/* out[c][b][t] = in[b][c][t]: a torch [B, C, T] batch into the channel-major layout of the training kernels. */
int evmi_transpose_bct_cbt_f32(const float* in_dev, float* out_dev, int B, int C, int T, void* stream);
/* counter[0] += delta (the device-side step counters of the optimisers). */
int evmi_counter_add_i32(int* counter_dev, int delta, void* stream);
/* torch.nn.utils.spectral_norm backward with sigma and <dw, W> left on the device:
 * gw[r][c] += dw[r][c] / sigma - (dot / sigma^2) u[r] v[c]. */
int evmi_spectral_norm_grad_f32(float* gw_dev, const float* dw_dev, const float* u_dev, const float* v_dev,
                                const float* sigma_dev, const float* dot_dev, int rows, int cols, void* stream);
/* out[0] += weight * sqrt(sq[0] / sq[1]): the spectral-convergence term of the multi-resolution STFT loss. */
int evmi_ratio_accumulate_f32(float* out_dev, const float* sq_dev, float weight, void* stream);
/* torch.optim.AdamW step `step` (1-based) on a flat parameter buffer. */
int evmi_adamw_f32(float* p_dev, const float* g_dev, float* m_dev, float* v_dev, long long n, float lr,
                   float beta1, float beta2, float eps, float weight_decay, int step, void* stream);


/* ------------------------------------------------------------------------------------------
 * Data-parallel gradient exchange (what Lightning's DDP strategy does for the reference,
 * base_cli/helpers.py:252-270: --devices N --strategy ddp): RCCL all-reduce of one contiguous fp32 bucket of a flat gradient
 * buffer, in place, summed over the ranks and scaled (1 / world for a mean) on `stream`.  One process per GPU; rank 0 makes the
 * 128-byte id and hands it to the others over any side channel.  RCCL is resolved at run time (a copy already in the process,
 * else librccl.so): EVMI_ERR_UNSUPPORTED when there is none. */
int evmi_comm_unique_id(void* id_out_128_bytes);
int evmi_comm_init_rank(void** comm_out, int world, const void* id_128_bytes, int rank);
int evmi_comm_destroy(void* comm);
int evmi_allreduce_bucket(void* comm, float* grad_dev, long long n, float scale, void* stream);

/* ------------------------------------------------------------------------------------------
 * Audio gating of the preprocessor (everyvoice/preprocessor/preprocessor.py:131-218).
 * evmi_loudness_lkfs_f32: integrated loudness (ITU-R BS.1770-4 as torchaudio.transforms.Loudness computes it: K-weighting
 * biquads, 400 ms blocks at 75 % overlap, absolute -70 and relative -10 LU gates) of `items` zero-padded utterances
 * x [items][channels][t_max] with lens [items]; the reference skips a file when the value is NaN or below -36
 * (preprocessor.py:177-185).  Scratch: y2 [items * channels * t_max] floats, z [evmi_loudness_scratch_elems(...)] floats.
 * evmi_peak_normalize_f32: dst = src / max|src| * target per utterance (preprocessor.py:199-201, target 0.95), zeros past lens.
 * Resampling (preprocessor.py:196-198, torchaudio.functional.resample) is evmi_conv1d_f32 with the windowed-sinc polyphase
 * kernel built by the host (everyvoice_amd/pipeline.py: sinc_resample_kernel). */
int evmi_loudness_lkfs_f32(const float* x_dev, const int* lens_dev, float* y2_scratch_dev, float* z_scratch_dev,
                           float* lkfs_dev, int items, int channels, int t_max, int sample_rate, void* stream);
long long evmi_loudness_scratch_elems(int items, int channels, int t_max, int sample_rate);
int evmi_peak_normalize_f32(const float* src_dev, float* dst_dev, const int* lens_dev, int items, int t_max, float target,
                            void* stream);
/* Frame-level F0 for FastSpeech2's pitch targets (call site everyvoice/preprocessor/preprocessor.py:244-285).
 * The reference's estimator is pyworld's dio + stonemask (third-party CPU code) and is NOT reproduced: this is a
 * normalised-autocorrelation tracker with the same interface -- f0 [items][t_max / hop + 1] in Hz at t = f * hop, 0 where
 * unvoiced, search range [f0_floor, f0_ceil] (WORLD's defaults are 71 and 800 Hz).  The NaN-interpolation over unvoiced
 * frames that follows in the reference is host-side in everyvoice_amd/pipeline.py: extract_pitch. */
int evmi_pitch_acf_f32(const float* audio_dev, const int* lens_dev, float* f0_dev, int items, int t_max, int hop,
                       int sample_rate, float f0_floor, float f0_ceil, float threshold, void* stream);
/* The reference's estimator itself: WORLD's DIO + StoneMask in float64 on the device (replaces pyworld.dio(x, fs, frame_period = hop / fs
 * * 1000, speed) -> pyworld.stonemask at everyvoice/preprocessor/preprocessor.py:244-285; the algorithm of WORLD's src/dio.cpp,
 * src/stonemask.cpp, src/matlabfunctions.cpp, restated independently in oracle/pitch_world_ref.py, which reproduces the reference's own
 * pyworld fixture to 1e-13 Hz).  audio [items][t_max] fp32 (zero padded; lens NULL = t_max each) -> f0 [items][frames_max] fp32 in Hz at
 * t = f * hop / fs, 0 where unvoiced, frames_max = (int)(1000.0 * t_max / fs / (hop / fs * 1000.0)) + 1 (pyworld's frame count, in
 * the same double arithmetic); item i has (int)(1000.0 * lens[i] / fs / frame_period) + 1 frames, the rest of its row is 0.
 * f0_floor / f0_ceil / channels_in_octave / allowed_range: pyworld's defaults are 71, 800, 2, 0.1; speed 1..12 (the reference passes 4).
 * ws: evmi_pitch_world_ws_elems doubles, 16-byte aligned.  evmi_pitch_world_decimator: the anti-alias filter the path designs for a
 * decimation ratio (host-side, no GPU: a[3], b[2] of WORLD's FilterForDecimate). */
long long evmi_pitch_world_ws_elems(int items, int t_max, int sample_rate, int hop, int speed, float f0_floor, float f0_ceil,
                                    float channels_in_octave);
int evmi_pitch_world_f64(const float* audio_dev, const int* lens_dev, float* f0_dev, double* ws_dev, long long ws_elems, int items, int t_max,
                         int sample_rate, int hop, int speed, float f0_floor, float f0_ceil, float channels_in_octave, float allowed_range,
                         void* stream);
int evmi_pitch_world_decimator(int ratio, double* a3_host, double* b2_host);

/* ------------------------------------------------------------------------------------------
 * FastSpeech2 feature-prediction forward path, channel-major fp32 x[c][b][t].
 * The reference keeps these inside the absent submodule FastSpeech2_lightning (package fs2); call
 * sites: everyvoice/tests/model_stubs.py:44-58 (constructor), everyvoice/demo/app.py:84-106
 * (synthesize_helper).  Linear / pointwise / postnet convolutions go through evmi_conv1d_cbt_f32.
 * ------------------------------------------------------------------------------------------ */
/* out[c][b][l] = l < lens[b] ? table[ids[b][l]][c] + pe(l, c) : 0, pe = cat(sin(l*inv_freq), cos(l*inv_freq))
 * (`position_embedding.inv_freq` [D/2] is the tensor everyvoice/tests/data/test.ckpt holds); inv_freq NULL = no pe term. */
int evmi_fs2_embed_f32(const int* ids_dev, const int* lens_dev, const float* table_dev,
                       const float* inv_freq_dev, float* out_dev, int B, int L, int D, void* stream);
/* x[c][b][t] = t < lens[b] ? x + pe(t, c) : 0   (decoder input). */
int evmi_fs2_add_posemb_f32(float* x_dev, const int* lens_dev, const float* inv_freq_dev, int B, int T,
                            int D, void* stream);
/* x[c][b][t] = 0 for t >= lens[b]. */
int evmi_mask_cols_f32(float* x_dev, const int* lens_dev, int C, int B, int T, void* stream);
/* LayerNorm over the channel axis of every column (biased variance, eps inside the root). */
int evmi_layernorm_cbt_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev,
                           int C, long long n_cols, float eps, void* stream);
/* Depthwise Conv1d(C, C, k, groups=C, padding=pad) per item, act: 0 none, 1 SiLU, 2 ReLU
 * (the conformer's convolution module with its eval-mode BatchNorm folded into w / bias; the depthwise half
 * of everyvoice/model/utils.py:5-48). */
int evmi_dwconv1d_cbt_f32(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int C,
                          int B, int T, int k, int pad, int act, void* stream);
/* x[c][b][l] += table[bucketize(values[b][l] * control, bins)][c]  (pitch / energy embeddings; bins has
 * n_bins - 1 boundaries, torch.bucketize semantics). */
int evmi_fs2_bucket_embed_add_f32(float* x_dev, const float* values_dev, const float* bins_dev,
                                  const float* table_dev, int n_bins, int B, int L, int D, float control,
                                  void* stream);
/* dur[b][l] = l < lens[b] ? max(0, round_half_even(exp(log_d) - 1) * control) : 0. */
int evmi_fs2_durations_i32(const float* log_d_dev, const int* lens_dev, int* dur_dev, int B, int L,
                           float control, void* stream);
/* Length regulator in this layout (everyvoice/utils/heavy.py:12-21 per item, zero padded to T):
 * out[c][b][t] = x[c][b][token(b, t)], cum = inclusive prefix sums of the integer durations [B][L]. */
int evmi_length_regulate_cbt_f32(const float* x_dev, const int* cum_dev, float* out_dev, int C, int B, int L,
                                 int T, void* stream);
/* Multi-head self-attention with key padding mask from lens: qkv [3*D][B][T] -> out [D][B][T]
 * (softmax(Q K^T / sqrt(d_head)) V per head; fp32 matrix cores, online softmax; d_head 32 / 64 / 128). */
int evmi_attention_cbt_f32(const float* qkv_dev, const int* lens_dev, float* out_dev, int B, int T, int D,
                           int heads, void* stream);
/* The same with bf16 operands (Q, K, V and the probabilities rounded to bf16; scores, softmax statistics and the output in
 * fp32): precision="bf16" inference. */
int evmi_attention_cbt_bf16(const float* qkv_dev, const int* lens_dev, float* out_dev, int B, int T, int D,
                            int heads, void* stream);
/* Speaker / language embedding of a multi-speaker / multilingual model: x[c][b][l] += table[ids[b]][c] for l < lens[b]. */
int evmi_fs2_add_item_embedding_f32(float* x_dev, const int* ids_dev, const int* lens_dev, const float* table_dev,
                                    int B, int L, int D, void* stream);
/* Beta-binomial attention prior (everyvoice/preprocessor/attention_prior.py:34-67): the pmf table of
 * BetaBinomial(n = grid_mel, a = i, b = grid_text + 1 - i), i = 1..grid_text, on a [grid_mel][grid_text] grid,
 * zoomed with order-1 interpolation (scipy.ndimage.zoom semantics) to out [T][L], float64. */
int evmi_attention_prior_f64(double* out_dev, int T, int L, int grid_mel, int grid_text, void* stream);
/* Alignment learning (Badlani et al. 2021; FastPitch ConvAttention / AttentionCTCLoss / AttentionBinarizationLoss):
 * scores -temperature * ||q_t - k_l||^2 over the projected mel q [A][B][T] and text k [A][B][L], log-softmax over the tokens
 * plus log(prior + 1e-8) (prior [B][T][L] float64 or NULL) -> logprob [B][T][L]; soft = softmax over the unpadded tokens. */
int evmi_align_attention_f32(const float* q_dev, const float* k_dev, const double* prior_dev, const int* text_lens_dev,
                             float* soft_dev, float* logprob_dev, int A, int B, int T, int L, float temperature,
                             void* stream);
/* CTC forward-sum loss per item over logprob [B][T][L] (blank with log-probability `blank_logprob` prepended, targets
 * 1..L_b, zero_infinity, divided by L_b); the batch loss is their mean. */
int evmi_forward_sum_loss_f32(const float* logprob_dev, const int* text_lens_dev, const int* mel_lens_dev,
                              float* loss_per_item_dev, int B, int T, int L, float blank_logprob, void* stream);
/* Binarisation loss partial sums: partials[2*i] = sum log(max(soft, 1e-12)) over hard == 1, partials[2*i+1] = count. */
int evmi_binarization_partials_f64(const int* hard_dev, const float* soft_dev, double* partials_dev, int n_blocks,
                                   long long n, void* stream);
/* Monotonic alignment search (the reference's hard alignments: third-party ilt-monotonic-align 1.2.1 = Glow-TTS
 * maximum_path): value [B][T][L] log-likelihoods, mel_lens / text_lens [B] -> path [B][T][L] (0/1, one token per
 * frame, monotonic, every token used) and durations [B][L] = frames per token.  scratch: B*T*L bytes. */
int evmi_monotonic_align_f32(const float* value_dev, const int* mel_lens_dev, const int* text_lens_dev,
                             int* path_dev, int* dur_dev, unsigned char* scratch_dev, int B, int T, int L,
                             void* stream);

/* ------------------------------------------------------------------------------------------
 * FastSpeech2 training (BASELINE config 3): backward and training-mode operators,
 * channel-major fp32 x[c][b][t].  The reference gets these from PyTorch autograd inside the absent
 * submodule FastSpeech2_lightning (driver contract: everyvoice/base_cli/helpers.py:173-195).  Dense
 * layers use evmi_conv1d_cbt_f32 / evmi_conv1d_dgrad_cbt_f32 / evmi_conv1d_wgrad_cbt_f32 (k = 1).
 * ------------------------------------------------------------------------------------------ */
/* LayerNorm over channels, backward: dx (+)= ..., dgamma += sum_cols dy * xhat, dbeta += sum_cols dy (C <= 256).
 * ws: evmi_layernorm_bwd_cbt_f32_ws_elems floats (per-workgroup partial sums, reduced in a fixed order). */
long long evmi_layernorm_bwd_cbt_f32_ws_elems(int C, long long n_cols);
int evmi_layernorm_bwd_cbt_f32(const float* x_dev, const float* gamma_dev, const float* dy_dev, float* dx_dev,
                               float* dgamma_dev, float* dbeta_dev, float* ws_dev, long long ws_elems, int C,
                               long long n_cols, float eps, int accumulate_dx, void* stream);
/* dgamma_dev == dbeta_dev == NULL above: the parameter gradients are NOT reduced -- their partial sums stay in ws (which the caller
 * then keeps) and a later call reduces the partial lists of many LayerNorms in one launch (same fixed order, same bits): a training
 * step's backward issues it once where the chain ends instead of one small launch per LayerNorm on the chain. */
typedef struct evmi_ln_partials {
  const float* ws;     /* the ws of an evmi_layernorm_bwd_cbt_f32 call with NULL gradients */
  float* dgamma;       /* [C], accumulated into */
  float* dbeta;        /* [C], accumulated into */
  int C;
  long long n_cols;    /* of that call */
} evmi_ln_partials;
int evmi_layernorm_bwd_partials_reduce(int n_jobs, const evmi_ln_partials* jobs, void* stream);
/* BatchNorm1d in training mode over every column of a channel row, followed by act (0 none, 2 SiLU, 4 tanh):
 * writes the batch mean / 1/sqrt(var + eps) [C] and, when given, updates the running statistics (unbiased variance).
 * momentum < 0: evaluation mode (model.eval() in the reference's validation loop) -- the running statistics normalise and are
 * not updated. */
int evmi_batchnorm_fwd_cbt_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, float* y_dev,
                               float* mean_dev, float* rstd_dev, float* running_mean_dev, float* running_var_dev,
                               int C, long long n_cols, float eps, float momentum, int act, void* stream);
/* dx = ..., dgamma += ..., dbeta += ... through act and the batch statistics. */
int evmi_batchnorm_bwd_cbt_f32(const float* x_dev, const float* gamma_dev, const float* beta_dev, const float* mean_dev,
                               const float* rstd_dev, const float* dy_dev, float* dx_dev, float* dgamma_dev,
                               float* dbeta_dev, int C, long long n_cols, int act, void* stream);
/* Depthwise convolution backward: dx (NULL = skip), dw [C][k] += ..., db [C] += ... (dw NULL = skip both; otherwise
 * ws: evmi_dwconv1d_bwd_cbt_f32_ws_elems floats of per-(channel, item) partial sums, reduced in a fixed order). */
long long evmi_dwconv1d_bwd_cbt_f32_ws_elems(int C, int B, int k);
int evmi_dwconv1d_bwd_cbt_f32(const float* x_dev, const float* w_dev, const float* dy_dev, float* dx_dev, float* dw_dev,
                              float* db_dev, float* ws_dev, long long ws_elems, int C, int B, int T, int k, int pad,
                              void* stream);
/* Multi-head self-attention in TRAINING mode (csrc/attention_train.hip; the reference reaches torch.nn.MultiheadAttention
 * through torchaudio's Conformer inside the absent submodule FastSpeech2_lightning; SURVEY.md 8b names evmi_mha_{fwd,bwd}).
 * qkv [3D][B][T] channel-major, lens [B] = key padding mask.  Forward: out [D][B][T] and the per-query log-sum-exp
 * lse [B][heads][T] -- the only thing the backward needs besides its inputs (the probabilities are recomputed tile by tile;
 * attention dropout p regenerates its mask from (seed + head, (b * T + q) * T + k) of the counter-based generator that
 * evmi_dropout_f32 uses).  Backward: dqkv [3D][B][T] from d out; dsum [B][heads][T] is scratch.  fp32 matrix cores, one
 * writer per output element, fixed summation order (bitwise reproducible). */
int evmi_mha_fwd_f32(const float* qkv_dev, const int* lens_dev, float* out_dev, float* lse_dev, int B, int T, int D, int heads,
                     float p_drop, unsigned long long seed, const unsigned long long* seed_base_dev, void* stream);
int evmi_mha_bwd_f32(const float* qkv_dev, const int* lens_dev, const float* out_dev, const float* dout_dev,
                     const float* lse_dev, float* dsum_dev, float* dqkv_dev, int B, int T, int D, int heads, float p_drop,
                     unsigned long long seed, const unsigned long long* seed_base_dev, void* stream);
/* The same two passes with bf16 operands (precision "bf16", BASELINE config 3): Q / K / V / dO, the probabilities and the score
 * gradients are rounded to bf16 into v_mfma_f32_32x32x16_bf16; scores, softmax statistics and accumulators are fp32. */
int evmi_mha_fwd_bf16(const float* qkv_dev, const int* lens_dev, float* out_dev, float* lse_dev, int B, int T, int D, int heads,
                      float p_drop, unsigned long long seed, const unsigned long long* seed_base_dev, void* stream);
int evmi_mha_bwd_bf16(const float* qkv_dev, const int* lens_dev, const float* out_dev, const float* dout_dev,
                      const float* lse_dev, float* dsum_dev, float* dqkv_dev, int B, int T, int D, int heads, float p_drop,
                      unsigned long long seed, const unsigned long long* seed_base_dev, void* stream);
/* scores [B][Tq][Tk] -> softmax over the keys tk < lens[b] in place (0 beyond); with p > 0 also
 * dropped = dropout(probabilities, p) from the counter-based generator keyed by `seed`. */
int evmi_softmax_rows_f32(float* scores_dev, float* dropped_dev, const int* lens_dev, int B, int Tq, int Tk, float p,
                          unsigned long long seed, void* stream);
/* In place on dprobs: dS = scale * P * (dPm - sum_k P * dPm), dPm = dprobs through the same dropout mask. */
int evmi_softmax_bwd_rows_f32(const float* probs_dev, float* dprobs_dev, long long rows, int Tk, float scale, float p,
                              unsigned long long seed, void* stream);
/* GLU over two halves p = [a; b] of n_half elements each: dp = [dy * sigmoid(b); dy * a * sigmoid'(b)]. */
int evmi_glu_bwd_f32(const float* p_dev, const float* dy_dev, float* dp_dev, long long n_half, void* stream);
/* y[i] = keep(seed, i) ? x[i] / (1 - p) : 0 ; calling it on a gradient with the same seed is the backward.
 * `seed_base_dev` (may be NULL) here and in evmi_dropout_fused_f32 / evmi_mha_*: a device-resident 64-bit base; the effective
 * seed is then (seed + *seed_base_dev) & (2^63 - 1).  A training step captured into a HIP graph passes the draw's index as `seed`
 * and rewrites the base (seed, step, rank) before every replay: new masks per step, none baked into the graph (the reference's
 * torch.nn.Dropout draws from the process's generator state, which advances on its own). */
int evmi_dropout_f32(const float* x_dev, float* y_dev, long long n, float p, unsigned long long seed,
                     const unsigned long long* seed_base_dev, void* stream);
/* The same dropout stream fused with its neighbour in the Conformer block (torchaudio's `x + 0.5 * dropout(ffn(x))`,
 * `dropout(silu(.))` and their backwards): drop(v)[i] = keep(seed, i) ? v[i] / (1 - p) : 0 and
 *   mode 1: y = b + scale * drop(a)    2: y = drop(silu(a))    3: y = drop(a) * silu'(b)    4: y = scale * drop(a)
 * (b is read by modes 1 and 3 only; operands 16-byte aligned). */
int evmi_dropout_fused_f32(int mode, const float* a_dev, const float* b_dev, float* y_dev, long long n, float p,
                           unsigned long long seed, const unsigned long long* seed_base_dev, float scale, void* stream);
/* Embedding backward: dtable[ids[b][l]][c] += dx[c][b][l] for l < lens[b], ids != skip_id (padding_idx).  One thread per
 * (table row, channel) adds its tokens in order: bitwise reproducible, no atomics.  `rows` = rows of the table. */
int evmi_fs2_embed_bwd_f32(const float* dx_dev, const int* ids_dev, const int* lens_dev, float* dtable_dev, int rows,
                           int B, int L, int D, int skip_id, void* stream);
/* idx_ws: B*L ints of scratch (the bucket of every position, pads included as the forward adds there too). */
int evmi_fs2_bucket_embed_bwd_f32(const float* dx_dev, const float* values_dev, const float* bins_dev,
                                  float* dtable_dev, int* idx_ws_dev, int n_bins, int B, int L, int D, float control,
                                  void* stream);
int evmi_fs2_item_embedding_bwd_f32(const float* dx_dev, const int* ids_dev, const int* lens_dev, float* dtable_dev,
                                    int rows, int B, int L, int D, void* stream);
/* Length regulator backward in this layout: dx[c][b][l] = sum of dframes[c][b][t] over the token's frames
 * (cum = inclusive cumulative durations, as evmi_length_regulate_cbt_f32 takes them). */
int evmi_length_regulate_bwd_cbt_f32(const float* dframes_dev, const int* cum_dev, float* dx_dev, int C, int B, int L,
                                     int T, void* stream);

/* Alignment learning, backward side (F5).  CTC forward-sum loss per item AND grad [B][T][L] = weight / (B * L_b) * d loss_b /
 * d logprob (softmax row - state occupancy; zero outside the item's frames / tokens; zero_infinity).
 * ws: evmi_forward_sum_grad_f32_ws_elems floats (the stored alpha lattice and per-frame normalisers). */
long long evmi_forward_sum_grad_f32_ws_elems(int B, int T, int L);
int evmi_forward_sum_grad_f32(const float* logprob_dev, const int* text_lens_dev, const int* mel_lens_dev,
                              float* loss_per_item_dev, float* grad_dev, float* ws_dev, long long ws_elems, int B, int T,
                              int L, float blank_logprob, float weight, void* stream);
/* Backward of evmi_align_attention_f32 down to the distance scores: da [B][T][L] from the CTC gradient dlogprob (or NULL) and
 * the binarisation loss on `hard` (or NULL; bin_scale = weight / number of hard cells -- or the weight alone with the count in
 * bin_count_dev[0], so that a captured step serves batches with different frame counts); also rowsum [B][T] = sum_l da and
 * colsum [B][L] = sum_t da.  prior as given to the forward (or NULL). */
int evmi_align_attention_bwd_f32(const float* soft_dev, const float* logprob_dev, const double* prior_dev, const int* hard_dev,
                                 const float* dlogprob_dev, const int* text_lens_dev, float* da_dev, float* rowsum_dev,
                                 float* colsum_dev, int B, int T, int L, float bin_scale, const float* bin_count_dev,
                                 void* stream);
/* In place m[c][n] = coef * (x[c][n] * sums[n] - m[c][n]) over A rows of BN columns: finishes dq (m = K . da^T, coef = -2 temp)
 * and dk (m = Q . da, coef = -2 temp with the sign folded: dk = 2 temp (m - k * colsum)). */
int evmi_align_qk_grad_f32(const float* x_dev, const float* sums_dev, float* m_dev, int A, long long BN, float coef,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EVMI_H */
