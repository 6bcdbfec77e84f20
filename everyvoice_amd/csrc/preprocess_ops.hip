// Audio gating of the preprocessor on the device: integrated loudness of every utterance of a batch, the
// quantity behind the reference's "audio_empty" gate (everyvoice/preprocessor/preprocessor.py:177-185:
// torchaudio.transforms.Loudness(sr)(audio) < -36 LUFS or NaN -> skipped).
//
// torchaudio.functional.loudness restated (ITU-R BS.1770-4; torchaudio is not in this image: parity unpinned, checked against
// the numpy restatement in oracle/preprocess_ref.py):
//   K-weighting: treble_biquad(+4 dB, 1500 Hz, Q = 1/sqrt 2) then highpass_biquad(38 Hz, Q = 0.5), each lfilter(clamp=True)
//   400 ms blocks, 75 % overlap: z[block] = mean(y^2);  l[block] = -0.691 + 10 log10(sum_ch g_ch z_ch)   (g = 1 for the first
//   three channels); absolute gate l > -70; relative gate l > (-0.691 + 10 log10(sum_ch g_ch mean_gated z_ch)) - 10;
//   LKFS = -0.691 + 10 log10(sum_ch g_ch mean_{both gates} z_ch).
// The two biquads are a sequential recurrence per (utterance, channel): one thread each (a batch of utterances runs its
// recurrences side by side: ~240 k dependent steps for 11 s of audio, milliseconds per batch); block means and the gating run
// one workgroup per utterance.
#include "common.h"

namespace evmi {

struct Biquad { float b0, b1, b2, a1, a2; };

// y2[item][ch][t] = (K-weighted x)^2 ; x [items][channels][t_max] (zero padded), lens [items]
__global__ void kweight_square_kernel(const float* __restrict__ x, const int* __restrict__ lens, float* __restrict__ y2, int items,
                                      int channels, int t_max, Biquad s1, Biquad s2) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= items * channels) return;
  const int n = min(lens[idx / channels], t_max);
  const float* xs = x + (long long)idx * t_max;
  float* ys = y2 + (long long)idx * t_max;
  float x1 = 0.f, x2 = 0.f, y1 = 0.f, yy2 = 0.f;  // stage 1 state
  float u1 = 0.f, u2 = 0.f, v1 = 0.f, v2 = 0.f;   // stage 2 state
  for (int t = 0; t < n; ++t) {
    const float xv = xs[t];
    float y = s1.b0 * xv + s1.b1 * x1 + s1.b2 * x2 - s1.a1 * y1 - s1.a2 * yy2;
    x2 = x1; x1 = xv; yy2 = y1; y1 = y;           // the recurrence runs on the unclamped output; the clamp applies to what leaves the filter
    y = fminf(fmaxf(y, -1.f), 1.f);
    float v = s2.b0 * y + s2.b1 * u1 + s2.b2 * u2 - s2.a1 * v1 - s2.a2 * v2;
    u2 = u1; u1 = y; v2 = v1; v1 = v;
    v = fminf(fmaxf(v, -1.f), 1.f);
    ys[t] = v * v;
  }
}

// one workgroup per utterance: block means (per channel), both gates, LKFS.  scratch: [items][channels][max_blocks]
__global__ __launch_bounds__(256) void loudness_gate_kernel(const float* __restrict__ y2, const int* __restrict__ lens, float* __restrict__ zbuf,
                                                            float* __restrict__ lkfs, int channels, int t_max, int gate, int step, int max_blocks) {
  const int item = blockIdx.x;
  const int n = min(lens[item], t_max);
  const int nb = n >= gate ? (n - gate) / step + 1 : 0;
  float* z = zbuf + (long long)item * channels * max_blocks;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int c = 0; c < channels; ++c) {
    const float* ys = y2 + ((long long)item * channels + c) * t_max;
    for (int b = wave; b < nb; b += 4) {  // one wave per block
      float acc = 0.f;
      for (int t = lane; t < gate; t += 64) acc += ys[(long long)b * step + t];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
      if (lane == 0) z[c * max_blocks + b] = acc / (float)gate;
    }
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  if (nb == 0) { lkfs[item] = NAN; return; }
  const float bias = -0.691f;
  auto pass = [&](float thresh_abs, float thresh_rel, bool use_rel, float& out_energy) {
    int count = 0;
    float sums[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = 0; b < nb; ++b) {
      float e = 0.f;
      for (int c = 0; c < channels; ++c) e += (c < 3 ? 1.f : 1.41f) * z[c * max_blocks + b];
      const float l = bias + 10.f * log10f(e);
      if (l > thresh_abs && (!use_rel || l > thresh_rel)) {
        ++count;
        for (int c = 0; c < channels; ++c) sums[c] += z[c * max_blocks + b];
      }
    }
    float e = 0.f;
    for (int c = 0; c < channels; ++c) e += (c < 3 ? 1.f : 1.41f) * (sums[c] / (float)count);  // count 0 -> NaN, as torchaudio
    out_energy = e;
  };
  float e1, e2;
  pass(-70.f, 0.f, false, e1);
  const float gamma_rel = bias + 10.f * log10f(e1) - 10.f;
  pass(-70.f, gamma_rel, true, e2);
  lkfs[item] = bias + 10.f * log10f(e2);
}

// dst[b][t] = scale[b] * src[b][t] for t < lens[b], 0 beyond (peak normalisation to 0.95 with the per-utterance peak on the device)
__global__ void peak_normalize_kernel(const float* __restrict__ src, float* __restrict__ dst, const int* __restrict__ lens, int t_max, float target) {
  __shared__ float part[4];
  const int item = blockIdx.x;
  const int n = min(lens[item], t_max);
  const float* xs = src + (long long)item * t_max;
  float* ys = dst + (long long)item * t_max;
  float mx = 0.f;
  for (int t = threadIdx.x; t < n; t += 256) mx = fmaxf(mx, fabsf(xs[t]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
  for (int t = threadIdx.x; t < t_max; t += 256) ys[t] = t < n ? xs[t] / mx * target : 0.f;  // (x / max|x|) * 0.95, the reference's two steps
}

// ---- frame-level F0 ------------------------------------------------------------------------------------
// The reference calls pyworld (dio + stonemask, third-party C, fp64) on the CPU; that estimator is NOT reproduced.  This is a
// normalised-autocorrelation tracker with the same interface -- one value per hop at t = f * hop, 0 for unvoiced frames, search
// range [f0_floor, f0_ceil] = WORLD's defaults [71, 800] Hz -- so that FastSpeech2's pitch targets can be made on the device:
//   r(lag) = sum_n x[n] x[n + lag] / sqrt(sum_n x[n]^2 * sum_n x[n + lag]^2)      n over a window of `win` samples centred on the frame
//   best   = the SMALLEST lag whose r is a local maximum >= 0.95 * max_lag r (guards against picking a multiple of the period;
//            at 0.85 formant structure put a half-period peak in reach on real speech: octave-up runs on LJ010-0008)
//   voiced = r(best) >= threshold;  f0 = sr / (best + parabolic offset)
// One workgroup per (frame, item); lags are spread over the threads, the window is staged once in LDS.
__global__ __launch_bounds__(256) void pitch_acf_kernel(const float* __restrict__ audio, const int* __restrict__ lens, float* __restrict__ f0,
                                                        int t_max, int n_frames, int hop, int sr, int win, int lag_lo, int lag_hi, float threshold) {
  extern __shared__ float sm[];
  float* xs = sm;                       // win + lag_hi + 1 samples starting at the window's first sample
  float* r = xs + win + lag_hi + 2;     // r[lag - lag_lo + 1] with one guard entry on each side
  const int f = blockIdx.x, b = blockIdx.y;
  const int n = min(lens[b], t_max);
  const int start = f * hop - win / 2;
  const int span = win + lag_hi + 1;
  const float* ab = audio + (long long)b * t_max;
  for (int i = threadIdx.x; i < span; i += 256) {
    const int g = start + i;
    xs[i] = (g >= 0 && g < n) ? ab[g] : 0.f;
  }
  __syncthreads();
  __shared__ float e0s;
  if (threadIdx.x == 0) {
    float e = 0.f;
    for (int i = 0; i < win; ++i) e = fmaf(xs[i], xs[i], e);
    e0s = e;
  }
  __syncthreads();
  const float e0 = e0s;
  const int nl = lag_hi - lag_lo + 3;  // lags lag_lo - 1 .. lag_hi + 1
  for (int li = threadIdx.x; li < nl; li += 256) {
    const int lag = lag_lo - 1 + li;
    float c = 0.f, e1 = 0.f;
    for (int i = 0; i < win; ++i) {
      const float y = xs[i + lag];
      c = fmaf(xs[i], y, c);
      e1 = fmaf(y, y, e1);
    }
    const float den = sqrtf(e0 * e1);
    r[li] = den > 1e-12f ? c / den : 0.f;
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  float rmax = 0.f;
  for (int li = 1; li < nl - 1; ++li) rmax = fmaxf(rmax, r[li]);
  float out = 0.f;
  if (f * hop < n + hop && rmax >= threshold) {
    for (int li = 1; li < nl - 1; ++li) {
      if (r[li] >= 0.95f * rmax && r[li] >= r[li - 1] && r[li] >= r[li + 1]) {
        const float a = r[li - 1], c0 = r[li], d = r[li + 1];
        const float den = a - 2.f * c0 + d;
        const float off = fabsf(den) > 1e-12f ? 0.5f * (a - d) / den : 0.f;
        out = (float)sr / ((float)(lag_lo - 1 + li) + fminf(fmaxf(off, -0.5f), 0.5f));
        break;
      }
    }
  }
  f0[(long long)b * n_frames + f] = out;
}


// Finishing pass of the generic spectrogram transforms (spec types "linear" / "raw" / "mel" / "istft" of
// everyvoice/utils/heavy.py:47-119): the DFT is evmi_stft_frames_f32 + evmi_gemm_f32 and leaves real / imaginary planes [C][B*F];
// torchaudio's tensors are batch-major.  One thread per output element, consecutive threads along F (both sides coalesced).
__global__ void spectrogram_layout_kernel(int mode, const float* __restrict__ re, const float* __restrict__ im, float* __restrict__ out,
                                          int B, int C, int F) {
  const long long n = (long long)B * C * F;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int f = (int)(i % F);
  const int c = (int)((i / F) % C);
  const int b = (int)(i / ((long long)F * C));
  const long long plane = (long long)c * B * F + (long long)b * F + f;  // [C][B*F]
  if (mode == 0) out[i] = re[plane] * re[plane] + im[plane] * im[plane];
  else if (mode == 1) { out[2 * i] = re[plane]; out[2 * i + 1] = im[plane]; }
  else if (mode == 2) out[i] = re[plane];
  else if (mode == 4) out[i] = sqrtf(re[plane] * re[plane] + im[plane] * im[plane]);
  else {  // 3: complex [B][C][F][2] (re) -> real rows then imaginary rows [2C][B*F] (out)
    out[plane] = re[2 * i];
    out[(long long)C * B * F + plane] = re[2 * i + 1];
  }
}

}  // namespace evmi

using namespace evmi;

static Biquad normalized(double b0, double b1, double b2, double a0, double a1, double a2) {
  return Biquad{(float)(b0 / a0), (float)(b1 / a0), (float)(b2 / a0), (float)(a1 / a0), (float)(a2 / a0)};
}

extern "C" {

int evmi_loudness_lkfs_f32(const float* x_dev, const int* lens_dev, float* y2_scratch_dev, float* z_scratch_dev, float* lkfs_dev, int items,
                           int channels, int t_max, int sample_rate, void* stream) {
  if (!x_dev || !lens_dev || !y2_scratch_dev || !z_scratch_dev || !lkfs_dev) return fail(EVMI_ERR_INVALID_ARG, "loudness: null pointer");
  if (items <= 0 || channels <= 0 || channels > 5 || t_max <= 0 || sample_rate <= 0) return fail(EVMI_ERR_INVALID_ARG, "loudness: shape");
  const double pi = 3.14159265358979323846;
  // treble_biquad(gain 4 dB, 1500 Hz, Q 1/sqrt 2)
  double w0 = 2.0 * pi * 1500.0 / sample_rate, alpha = sin(w0) / 2.0 / (1.0 / sqrt(2.0)), A = exp(4.0 / 40.0 * log(10.0));
  double t1 = 2.0 * sqrt(A) * alpha, t2 = (A - 1.0) * cos(w0), t3 = (A + 1.0) * cos(w0);
  const Biquad s1 = normalized(A * ((A + 1.0) + t2 + t1), -2.0 * A * ((A - 1.0) + t3), A * ((A + 1.0) + t2 - t1), (A + 1.0) - t2 + t1,
                               2.0 * ((A - 1.0) - t3), (A + 1.0) - t2 - t1);
  // highpass_biquad(38 Hz, Q 0.5)
  w0 = 2.0 * pi * 38.0 / sample_rate;
  alpha = sin(w0) / 2.0 / 0.5;
  const Biquad s2 = normalized((1.0 + cos(w0)) / 2.0, -1.0 - cos(w0), (1.0 + cos(w0)) / 2.0, 1.0 + alpha, -2.0 * cos(w0), 1.0 - alpha);
  const int gate = (int)llround(0.4 * sample_rate), step = (int)(gate * (1.0 - 0.75));
  const int max_blocks = t_max >= gate ? (t_max - gate) / step + 1 : 1;
  hipStream_t s = (hipStream_t)stream;
  const int n = items * channels;
  hipLaunchKernelGGL(kweight_square_kernel, dim3((n + 63) / 64), dim3(64), 0, s, x_dev, lens_dev, y2_scratch_dev, items, channels, t_max, s1, s2);
  hipLaunchKernelGGL(loudness_gate_kernel, dim3(items), dim3(256), 0, s, y2_scratch_dev, lens_dev, z_scratch_dev, lkfs_dev, channels, t_max, gate, step,
                     max_blocks);
  EVMI_LAUNCH_CHECK("loudness");
  return EVMI_OK;
}

long long evmi_loudness_scratch_elems(int items, int channels, int t_max, int sample_rate) {
  const int gate = (int)llround(0.4 * sample_rate), step = (int)(gate * (1.0 - 0.75));
  const int max_blocks = t_max >= gate ? (t_max - gate) / (step > 0 ? step : 1) + 1 : 1;
  return (long long)items * channels * max_blocks;
}

int evmi_pitch_acf_f32(const float* audio_dev, const int* lens_dev, float* f0_dev, int items, int t_max, int hop, int sample_rate, float f0_floor,
                       float f0_ceil, float threshold, void* stream) {
  if (!audio_dev || !lens_dev || !f0_dev || items <= 0 || t_max <= 0 || hop <= 0 || sample_rate <= 0 || f0_floor <= 0.f || f0_ceil <= f0_floor)
    return fail(EVMI_ERR_INVALID_ARG, "pitch_acf: arguments");
  const int lag_lo = (int)floorf((float)sample_rate / f0_ceil), lag_hi = (int)ceilf((float)sample_rate / f0_floor);
  const int win = 2 * lag_hi;  // two periods of the lowest pitch
  const int n_frames = t_max / hop + 1;
  const size_t lds = ((size_t)(win + lag_hi + 2) + (size_t)(lag_hi - lag_lo + 3)) * sizeof(float);
  if (lag_lo < 2 || lds > 64 * 1024 || items > 65535) return fail(EVMI_ERR_UNSUPPORTED, "pitch_acf: search range");
  hipLaunchKernelGGL(pitch_acf_kernel, dim3(n_frames, items), dim3(256), lds, (hipStream_t)stream, audio_dev, lens_dev, f0_dev, t_max, n_frames, hop,
                     sample_rate, win, lag_lo, lag_hi, threshold);
  EVMI_LAUNCH_CHECK("pitch_acf");
  return EVMI_OK;
}

int evmi_peak_normalize_f32(const float* src_dev, float* dst_dev, const int* lens_dev, int items, int t_max, float target, void* stream) {
  if (!src_dev || !dst_dev || !lens_dev || items <= 0 || t_max <= 0) return fail(EVMI_ERR_INVALID_ARG, "peak_normalize: arguments");
  hipLaunchKernelGGL(peak_normalize_kernel, dim3(items), dim3(256), 0, (hipStream_t)stream, src_dev, dst_dev, lens_dev, t_max, target);
  EVMI_LAUNCH_CHECK("peak_normalize");
  return EVMI_OK;
}

int evmi_spectrogram_layout_f32(int mode, const float* re_dev, const float* im_dev, float* out_dev, int B, int C, int F, void* stream) {
  if (mode < 0 || mode > 4 || !re_dev || !out_dev || ((mode == 0 || mode == 1 || mode == 4) && !im_dev) || B <= 0 || C <= 0 || F <= 0)
    return fail(EVMI_ERR_INVALID_ARG, "spectrogram_layout: arguments");
  const long long n = (long long)B * C * F;
  hipLaunchKernelGGL(spectrogram_layout_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mode, re_dev, im_dev, out_dev, B, C, F);
  EVMI_LAUNCH_CHECK("spectrogram_layout");
  return EVMI_OK;
}

}  // extern "C"
