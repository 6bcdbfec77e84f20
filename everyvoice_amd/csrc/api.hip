// C-ABI entry points that map 1:1 onto a kernel family (see include/evmi.h).
#include "common.h"

namespace evmi {
int length_regulate(const void*, const int64_t*, void*, int64_t*, int32_t*, int, int, int, int, int, hipStream_t);
int length_regulate_bwd_f32(const float*, const int64_t*, float*, int, int, int, int, hipStream_t);
int launch_conv1d_f32(const float*, const float*, const float*, const float*, float*, int, int, int, int,
                      int, int, int, int, int, float, float, int, hipStream_t);
int launch_conv_transpose1d_f32(const float*, const float*, const float*, float*, int, int, int, int, int,
                                int, int, float, hipStream_t);
int launch_mel_frontend(const float*, const float*, const float*, float*, float*, float*, int, int, int, int, int,
                        int, int, int, int, hipStream_t, const int* lens);
}  // namespace evmi

using namespace evmi;

extern "C" {

int evmi_length_regulate(const void* values_dev, const int64_t* durations_dev, void* out_dev,
                         int64_t* out_lens_dev, int32_t* index_dev, int B, int L, int D, int t_max,
                         int elem_bytes, void* stream) {
  if ((!values_dev && B * L * D > 0) || (!durations_dev && B * L > 0) || (!out_dev && B * t_max * D > 0))
    return fail(EVMI_ERR_INVALID_ARG, "length_regulate: null pointer");
  return length_regulate(values_dev, durations_dev, out_dev, out_lens_dev, index_dev, B, L, D, t_max,
                         elem_bytes, (hipStream_t)stream);
}

int evmi_length_regulate_bwd_f32(const float* grad_out_dev, const int64_t* durations_dev,
                                 float* grad_values_dev, int B, int L, int D, int t_max, void* stream) {
  if (!grad_out_dev || !durations_dev || !grad_values_dev)
    return fail(EVMI_ERR_INVALID_ARG, "length_regulate_bwd: null pointer");
  return length_regulate_bwd_f32(grad_out_dev, durations_dev, grad_values_dev, B, L, D, t_max,
                                 (hipStream_t)stream);
}

int evmi_conv1d_f32(const float* x_dev, const float* w_dev, const float* bias_dev, const float* residual_dev,
                    float* y_dev, int B, int c_in, int t_in, int c_out, int k, int stride, int pad, int dil,
                    int groups, float pre_slope, float out_scale, int accumulate, void* stream) {
  if (!x_dev || !w_dev || !y_dev) return fail(EVMI_ERR_INVALID_ARG, "conv1d_f32: null pointer");
  return launch_conv1d_f32(x_dev, w_dev, bias_dev, residual_dev, y_dev, B, c_in, t_in, c_out, k, stride, pad,
                           dil, groups, pre_slope, out_scale, accumulate, (hipStream_t)stream);
}

int evmi_conv_transpose1d_f32(const float* x_dev, const float* w_dev, const float* bias_dev, float* y_dev, int B,
                              int c_in, int t_in, int c_out, int k, int stride, int pad, float pre_slope,
                              void* stream) {
  if (!x_dev || !w_dev || !y_dev) return fail(EVMI_ERR_INVALID_ARG, "conv_transpose1d_f32: null pointer");
  return launch_conv_transpose1d_f32(x_dev, w_dev, bias_dev, y_dev, B, c_in, t_in, c_out, k, stride, pad,
                                     pre_slope, (hipStream_t)stream);
}

int evmi_mel_spectrogram_f32(const float* audio_dev, const float* dft_basis_dev, const float* mel_basis_dev,
                             float* mel_dev, float* energy_dev, float* mag_dev, int B, int n_samples, int n_fft,
                             int hop, int n_bins_padded, int n_mels, int apply_log, void* stream) {
  if (!audio_dev || !dft_basis_dev || !mel_basis_dev || !mel_dev)
    return fail(EVMI_ERR_INVALID_ARG, "mel_spectrogram: null pointer");
  if (B <= 0 || n_samples <= 0 || n_mels <= 0 || n_bins_padded % 16 || n_bins_padded < n_fft / 2 + 1)
    return fail(EVMI_ERR_INVALID_ARG, "mel_spectrogram: bad shape");
  const int n_frames = 1 + n_samples / hop;
  return launch_mel_frontend(audio_dev, dft_basis_dev, mel_basis_dev, mel_dev, energy_dev, mag_dev, B, n_samples,
                             n_frames, n_fft, hop, n_bins_padded, n_fft / 2 + 1, n_mels, apply_log, (hipStream_t)stream, nullptr);
}

int evmi_mel_spectrogram_ragged_f32(const float* audio_dev, const int* lens_dev, const float* dft_basis_dev, const float* mel_basis_dev,
                                    float* mel_dev, float* energy_dev, float* mag_dev, int B, int n_samples_max, int n_fft, int hop,
                                    int n_bins_padded, int n_mels, int apply_log, void* stream) {
  if (!audio_dev || !lens_dev || !dft_basis_dev || !mel_basis_dev || !mel_dev)
    return fail(EVMI_ERR_INVALID_ARG, "mel_spectrogram_ragged: null pointer");
  if (B <= 0 || n_samples_max <= 0 || n_mels <= 0 || n_bins_padded % 16 || n_bins_padded < n_fft / 2 + 1)
    return fail(EVMI_ERR_INVALID_ARG, "mel_spectrogram_ragged: bad shape");
  const int n_frames = 1 + n_samples_max / hop;
  return launch_mel_frontend(audio_dev, dft_basis_dev, mel_basis_dev, mel_dev, energy_dev, mag_dev, B, n_samples_max,
                             n_frames, n_fft, hop, n_bins_padded, n_fft / 2 + 1, n_mels, apply_log, (hipStream_t)stream, lens_dev);
}

}  // extern "C"
