// WORLD's DIO + StoneMask F0 estimators on the device, in float64 -- the estimator behind the reference's pitch targets
// (everyvoice/preprocessor/preprocessor.py:244-285: pyworld.dio(audio.f64, fs, frame_period = hop / fs * 1000, speed = 4) ->
// pyworld.stonemask).  pyworld wraps M. Morise's WORLD; what is implemented here is the published algorithm in the
// form of its sources (src/dio.cpp, src/stonemask.cpp, src/matlabfunctions.cpp), restated independently in oracle/pitch_world_ref.py,
// which reproduces the reference's own pyworld fixture (tests/data/ming024/eng-LJSpeech-pitch-LJ010-0008.npy) to 1e-13 Hz.
//
// WORLD filters in the frequency domain (one FFT of the whole utterance per band); its FFT sizes are chosen so that every circular
// convolution IS the linear one, so the same signals are computed here in the time domain -- a few hundred taps per sample and band,
// nothing for this chip -- and the StoneMask spectra, needed at <= 8 harmonic bins per frame, as direct DFT sums over an exact
// (index mod N) twiddle table.  Sums run in another order than WORLD's FFTs: values agree to ~1e-13, decisions (zero crossings,
// thresholds, rounded bin numbers) are the same ones.  One utterance = one batch item; the stages:
//   1  dio_decimate        IIR anti-alias filter (Chebyshev I, order 3: FilterForDecimate) forward + backward, every r-th sample, mean removed
//   2  dio_lowcut          50 Hz low cut (delta minus a normalised Hann window: DesignLowCutFilter)
//   3  dio_band            per band: Nuttall low-pass of length 4 h (GetFilteredSignal), delay-compensated
//   4  dio_events          per (band, kind of event): ordered list of sub-sample zero-crossing positions (ZeroCrossingEngine)
//   5  dio_candidates      per (band, frame): the four interval contours interpolated at the frame time, their mean and deviation
//   6  dio_contour         per item: best band per frame, FixStep1 .. FixStep4
//   7  stonemask           per voiced frame: instantaneous-frequency refinement from <= 6 harmonics
#include <cmath>

#include "common.h"

namespace evmi {

constexpr double kPiD = 3.14159265358979323846;
constexpr double kWorldMaximumValue = 100000.0;
constexpr double kWorldSafeGuard = 1e-12;
constexpr int kMaxBands = 16;

__host__ __device__ inline int world_round(double x) { return x > 0 ? (int)(x + 0.5) : (int)(x - 0.5); }

struct DioGeom {  // per launch
  int items, t_max, fs, ratio, n_bands, frames_max, y_max, c_lowcut, z_len;
  double actual_fs, frame_period, f0_floor, f0_ceil, allowed_range;
  double boundary[kMaxBands];
  int half[kMaxBands];
  double a[3], b[2];  // FilterForDecimate
  // workspace (doubles, per item unless noted)
  double* tmp1;     // [t_max + 18]
  double* tmp2;     // [t_max + 18]
  double* y;        // [y_max]
  double* z;        // [z_len]  = low-cut signal on [-c, y_len + c)
  double* filt;     // [n_bands][y_max]
  double* edges;    // [n_bands][4][y_max]
  int* n_edges;     // [n_bands][4]
  double* cand;     // [n_bands][frames_max]
  double* score;    // [n_bands][frames_max]
  double* f0;       // [frames_max]   DIO's contour
  double* scratch;  // [4][frames_max]
  double* twiddle;  // shared: [2][tw_n]
  int tw_n;
};

__device__ __forceinline__ int item_len(const int* lens, int item, int t_max) {
  const int n = lens ? lens[item] : t_max;
  return n < 0 ? 0 : (n > t_max ? t_max : n);
}
__device__ __forceinline__ int item_frames(int n, int fs, double frame_period) { return (int)(1000.0 * n / fs / frame_period) + 1; }

// ---- 1: decimation (matlabfunctions.cpp: decimate / FilterForDecimate) ---------------------------------------------------------
__global__ __launch_bounds__(256) void dio_decimate_kernel(DioGeom g, const float* __restrict__ audio, const int* __restrict__ lens) {
  const int item = blockIdx.x;
  const int n = item_len(lens, item, g.t_max);
  const float* x = audio + (long long)item * g.t_max;
  double* y = g.y + (long long)item * g.y_max;
  const int y_len = 1 + n / g.ratio;
  __shared__ double red[256];
  if (n < 32) {  // (shorter than the filter's reflection: nothing to estimate)
    for (int i = threadIdx.x; i < g.y_max; i += 256) y[i] = 0.0;
    return;
  }
  if (g.ratio == 1) {
    for (int i = threadIdx.x; i < g.y_max; i += 256) y[i] = i < n ? (double)x[i] : 0.0;
  } else {
    constexpr int NF = 9;
    double* t1 = g.tmp1 + (long long)item * (g.t_max + 2 * NF);
    double* t2 = g.tmp2 + (long long)item * (g.t_max + 2 * NF);
    const int N = n + 2 * NF;
    for (int i = threadIdx.x; i < N; i += 256) {
      double v;
      if (i < NF) v = 2.0 * (double)x[0] - (double)x[NF - i];
      else if (i < NF + n) v = (double)x[i - NF];
      else v = 2.0 * (double)x[n - 1] - (double)x[n - 2 - (i - (NF + n))];
      t1[i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // the recursion is sequential: one lane per item, items side by side
      const double a0 = g.a[0], a1 = g.a[1], a2 = g.a[2], b0 = g.b[0], b1 = g.b[1];
      double w0 = 0.0, w1 = 0.0, w2 = 0.0;
      for (int i = 0; i < N; ++i) {
        const double wt = t1[i] + a0 * w0 + a1 * w1 + a2 * w2;
        t2[i] = b0 * wt + b1 * w0 + b1 * w1 + b0 * w2;
        w2 = w1; w1 = w0; w0 = wt;
      }
      w0 = w1 = w2 = 0.0;
      for (int i = N - 1; i >= 0; --i) {  // (reverse, filter, reverse)
        const double wt = t2[i] + a0 * w0 + a1 * w1 + a2 * w2;
        t1[i] = b0 * wt + b1 * w0 + b1 * w1 + b0 * w2;
        w2 = w1; w1 = w0; w0 = wt;
      }
    }
    __syncthreads();
    const int nout = (n - 1) / g.ratio + 1;
    const int nbeg = g.ratio - g.ratio * nout + n;
    for (int c = threadIdx.x; c < g.y_max; c += 256) {
      const long long i = (long long)nbeg + (long long)c * g.ratio;
      y[c] = (i < n + NF && c < y_len) ? t1[i + NF - 1] : 0.0;
    }
  }
  __syncthreads();
  // removal of the DC component over y[0, y_len)
  double s = 0.0;
  for (int i = threadIdx.x; i < y_len; i += 256) s += y[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
    __syncthreads();
  }
  const double mean = red[0] / y_len;
  for (int i = threadIdx.x; i < g.y_max; i += 256) y[i] = i < y_len ? y[i] - mean : 0.0;
}

// ---- 2: low cut (dio.cpp: DesignLowCutFilter, applied in GetSpectrumForEstimation) -------------------------------------------------
__global__ __launch_bounds__(256) void dio_lowcut_kernel(DioGeom g, const int* __restrict__ lens) {
  extern __shared__ double hann[];  // [2 c + 1], normalised
  const int item = blockIdx.y;
  const int n = item_len(lens, item, g.t_max);
  const int y_len = 1 + n / g.ratio;
  const int c = g.c_lowcut, N = 2 * c + 1;
  __shared__ double hann_sum;
  if (threadIdx.x == 0) {  // (one thread, ascending order: the sum every workgroup divides by has the same bits)
    double sum = 0.0;
    for (int j = 0; j < N; ++j) sum += 0.5 - 0.5 * cos((j + 1) * 2.0 * kPiD / (N + 1));
    hann_sum = sum;
  }
  __syncthreads();
  for (int j = threadIdx.x; j < N; j += 256) hann[j] = (0.5 - 0.5 * cos((j + 1) * 2.0 * kPiD / (N + 1))) / hann_sum;
  __syncthreads();
  const double* y = g.y + (long long)item * g.y_max;
  double* z = g.z + (long long)item * g.z_len;
  const int i = blockIdx.x * 256 + threadIdx.x;  // z index: sample i - c
  if (i >= g.z_len) return;
  const int p = i - c;
  double acc = 0.0;
  if (p >= -c && p < y_len + c) {
    // z[p] = y[p] - sum_d hann[d + c] y[p - d], d in [-c, c]
    const int d_lo = max(-c, p - (y_len - 1)), d_hi = min(c, p);
    for (int d = d_lo; d <= d_hi; ++d) acc += hann[d + c] * y[p - d];
    acc = ((p >= 0 && p < y_len) ? y[p] : 0.0) - acc;
  }
  z[i] = acc;
}

// ---- 3: band filter (dio.cpp: GetFilteredSignal) -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dio_band_kernel(DioGeom g, const int* __restrict__ lens) {
  extern __shared__ double win[];  // Nuttall window, length 4 h
  const int item = blockIdx.z, band = blockIdx.y;
  const int n = item_len(lens, item, g.t_max);
  const int y_len = 1 + n / g.ratio;
  const int h = g.half[band], L = 4 * h;
  for (int k = threadIdx.x; k < L; k += 256) {
    const double t = k / (L - 1.0);
    win[k] = 0.355768 - 0.487396 * cos(2.0 * kPiD * t) + 0.144232 * cos(4.0 * kPiD * t) - 0.012604 * cos(6.0 * kPiD * t);
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= g.y_max) return;
  const double* z = g.z + (long long)item * g.z_len + g.c_lowcut;  // z[p], p in [-c, y_len + c)
  double acc = 0.0;
  if (i < y_len) {
    // filtered[i] = sum_k win[k] z[i + 2 h - k]
    for (int k = 0; k < L; ++k) {
      const int p = i + 2 * h - k;
      if (p >= -g.c_lowcut && p < y_len + g.c_lowcut) acc += win[k] * z[p];
    }
  }
  g.filt[((long long)item * g.n_bands + band) * g.y_max + i] = acc;
}

// ---- 4: zero-crossing events (dio.cpp: ZeroCrossingEngine / GetFourZeroCrossingIntervals) ----------------------------------------
// kind 0: negative-going crossings of f; 1: of -f; 2: of d[i] = f[i] - f[i + 1] (peaks); 3: of -d (dips).  One workgroup per
// (item, band, kind) walks the signal in order: flags -> ballot -> prefix over the waves -> positions in order.
__global__ __launch_bounds__(256) void dio_events_kernel(DioGeom g, const int* __restrict__ lens) {
  const int item = blockIdx.z, band = blockIdx.y, kind = blockIdx.x;
  const int n = item_len(lens, item, g.t_max);
  const int y_len = 1 + n / g.ratio;
  const double* f = g.filt + ((long long)item * g.n_bands + band) * g.y_max;
  double* out = g.edges + (((long long)item * g.n_bands + band) * 4 + kind) * g.y_max;
  const int L = kind < 2 ? y_len : y_len - 1;  // length of the signal whose crossings are taken
  const double sgn = (kind & 1) ? -1.0 : 1.0;
  __shared__ int wave_count[4];
  __shared__ int base;
  if (threadIdx.x == 0) base = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i0 = 0; i0 < L - 1; i0 += 256) {
    const int i = i0 + threadIdx.x;
    bool hit = false;
    double s0 = 0.0, s1 = 0.0;
    if (i < L - 1) {
      if (kind < 2) {
        s0 = sgn * f[i];
        s1 = sgn * f[i + 1];
      } else {
        s0 = sgn * (f[i] - f[i + 1]);
        s1 = sgn * (f[i + 1] - f[i + 2]);
      }
      hit = s0 > 0.0 && s1 <= 0.0;
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_count[wave] = __popcll(m);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; ++w) off += wave_count[w];
    if (hit) out[off + __popcll(m & ((1ull << lane) - 1ull))] = (double)(i + 1) - s0 / (s1 - s0);
    __syncthreads();
    if (threadIdx.x == 0) base += wave_count[0] + wave_count[1] + wave_count[2] + wave_count[3];
    __syncthreads();
  }
  if (threadIdx.x == 0) g.n_edges[((long long)item * g.n_bands + band) * 4 + kind] = base;
}

// ---- 5: candidates (dio.cpp: GetF0CandidateContour / ...Sub; matlabfunctions.cpp: interp1 + histc) ---------------------------------
__global__ __launch_bounds__(256) void dio_candidates_kernel(DioGeom g, const int* __restrict__ lens) {
  const int item = blockIdx.z, band = blockIdx.y;
  const int n = item_len(lens, item, g.t_max);
  const int frames = min(item_frames(n, g.fs, g.frame_period), g.frames_max);
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= g.frames_max) return;
  const long long ib = (long long)item * g.n_bands + band;
  double cand = 0.0, score = kWorldMaximumValue;
  if (j < frames) {
    const int* ne = g.n_edges + ib * 4;
    const bool ok = ne[0] - 1 - 2 > 0 && ne[1] - 1 - 2 > 0 && ne[2] - 1 - 2 > 0 && ne[3] - 1 - 2 > 0;  // CheckEvent(number - 2) of every kind
    if (ok) {
      const double t = j * g.frame_period / 1000.0, fs = g.actual_fs;
      double v[4];
#pragma unroll
      for (int kind = 0; kind < 4; ++kind) {
        const double* e = g.edges + (ib * 4 + kind) * g.y_max;
        const int M = ne[kind] - 1;  // intervals: location m = (e[m] + e[m + 1]) / 2 / fs, value fs / (e[m + 1] - e[m])
        int lo = 0, hi = M;          // number of locations <= t
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if ((e[mid] + e[mid + 1]) / 2.0 / fs <= t) lo = mid + 1;
          else hi = mid;
        }
        const int k = min(max(lo, 1), M - 1);
        const double x0 = (e[k - 1] + e[k]) / 2.0 / fs, x1 = (e[k] + e[k + 1]) / 2.0 / fs;
        const double y0 = fs / (e[k] - e[k - 1]), y1 = fs / (e[k + 1] - e[k]);
        const double s = (t - x0) / (x1 - x0);
        v[kind] = y0 + s * (y1 - y0);
      }
      cand = (v[0] + v[1] + v[2] + v[3]) / 4.0;
      score = sqrt(((v[0] - cand) * (v[0] - cand) + (v[1] - cand) * (v[1] - cand) + (v[2] - cand) * (v[2] - cand) + (v[3] - cand) * (v[3] - cand)) / 3.0);
      const double bf = g.boundary[band];
      if (cand > bf || cand < bf / 2.0 || cand > g.f0_ceil || cand < g.f0_floor) {
        cand = 0.0;
        score = kWorldMaximumValue;
      }
    }
  }
  g.cand[ib * g.frames_max + j] = cand;
  g.score[ib * g.frames_max + j] = score / (cand + kWorldSafeGuard);
}

// ---- 6: best contour + FixF0Contour (dio.cpp: GetBestF0Contour, FixStep1 .. FixStep4, SelectBestF0) -----------------------------------
__device__ double dio_select_best(const DioGeom& g, const double* cand_item, double current, double past, int target) {
  const double ref = (current * 3.0 - past) / 2.0;
  double best = cand_item[target], err = fabs(ref - best);
  for (int b = 1; b < g.n_bands; ++b) {
    const double c = cand_item[(long long)b * g.frames_max + target], e = fabs(ref - c);
    if (e < err) {
      err = e;
      best = c;
    }
  }
  if (fabs(1.0 - best / ref) > g.allowed_range) return 0.0;
  return best;
}
__global__ void dio_contour_kernel(DioGeom g, const int* __restrict__ lens) {
  const int item = blockIdx.x * blockDim.x + threadIdx.x;
  if (item >= g.items) return;
  const int n = item_len(lens, item, g.t_max);
  const int F = min(item_frames(n, g.fs, g.frame_period), g.frames_max);
  const double* cand = g.cand + (long long)item * g.n_bands * g.frames_max;
  const double* score = g.score + (long long)item * g.n_bands * g.frames_max;
  double* f0 = g.f0 + (long long)item * g.frames_max;
  double* sa = g.scratch + (long long)item * 4 * g.frames_max;
  double* best = sa;
  double* s1 = sa + g.frames_max;
  double* s2 = sa + 2 * g.frames_max;
  double* s3 = sa + 3 * g.frames_max;
  for (int i = 0; i < g.frames_max; ++i) f0[i] = 0.0;
  if (n < 32) return;
  for (int i = 0; i < F; ++i) {
    double t = score[i], v = cand[i];
    for (int b = 1; b < g.n_bands; ++b) {
      const double sc = score[(long long)b * g.frames_max + i];
      if (t > sc) {
        t = sc;
        v = cand[(long long)b * g.frames_max + i];
      }
    }
    best[i] = v;
  }
  const int vrm = (int)(0.5 + 1000.0 / g.frame_period / g.f0_floor) * 2 + 1;
  if (F <= vrm) return;
  // step 1 (f0_base = best with both ends zeroed)
  auto base = [&](int i) { return (i < vrm || i >= F - vrm) ? 0.0 : best[i]; };
  for (int i = 0; i < vrm; ++i) s1[i] = 0.0;
  for (int i = vrm; i < F; ++i) s1[i] = fabs((base(i) - base(i - 1)) / (kWorldSafeGuard + base(i))) < g.allowed_range ? base(i) : 0.0;
  // step 2
  const int c = (vrm - 1) / 2;
  for (int i = 0; i < F; ++i) s2[i] = s1[i];
  for (int i = c; i < F - c; ++i)
    for (int j = -c; j <= c; ++j)
      if (s1[i + j] == 0) {
        s2[i] = 0.0;
        break;
      }
  // step 3: every voiced section forward (sections end where s2 turns to zero)
  for (int i = 0; i < F; ++i) s3[i] = s2[i];
  {
    int i = 1;
    while (i < F) {
      if (s2[i] == 0 && s2[i - 1] != 0) {
        const int start = i - 1;
        int limit = F - 1;  // the next section end, or the last frame
        for (int q = i + 1; q < F; ++q)
          if (s2[q] == 0 && s2[q - 1] != 0) {
            limit = q - 1;
            break;
          }
        for (int j = start; j < limit; ++j) {
          s3[j + 1] = dio_select_best(g, cand, s3[j], s3[j - 1], j + 1);
          if (s3[j + 1] == 0) break;
        }
      }
      ++i;
    }
  }
  // step 4: every voiced section backward (sections of s2 start where it leaves zero), last section first
  for (int i = 0; i < F; ++i) f0[i] = s3[i];
  {
    int i = F - 1;
    while (i >= 1) {
      if (s2[i - 1] == 0 && s2[i] != 0) {
        const int start = i;
        int limit = 1;  // the previous section start, or frame 1
        for (int q = i - 1; q >= 1; --q)
          if (s2[q - 1] == 0 && s2[q] != 0) {
            limit = q;
            break;
          }
        for (int j = start; j > limit; --j) {
          f0[j - 1] = dio_select_best(g, cand, f0[j], f0[j + 1], j - 1);
          if (f0[j - 1] == 0) break;
        }
      }
      --i;
    }
  }
}

// ---- 7: StoneMask (stonemask.cpp: GetRefinedF0, GetMeanF0, FixF0) ----------------------------------------------------------------------
__global__ void world_twiddle_kernel(double* tw, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  double s, c;
  sincospi(2.0 * i / N, &s, &c);
  tw[i] = c;
  tw[N + i] = s;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
constexpr int kStoneWaves = 4;
__global__ __launch_bounds__(64 * kStoneWaves) void stonemask_kernel(DioGeom g, const float* __restrict__ audio, const int* __restrict__ lens,
                                                                      float* __restrict__ out, int n_win_max) {
  extern __shared__ double lds[];  // per wave: main window [n_win_max], windowed samples are recomputed
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int item = blockIdx.y;
  const int frame = blockIdx.x * kStoneWaves + wave;
  if (frame >= g.frames_max) return;
  const int n = item_len(lens, item, g.t_max);
  const int F = min(item_frames(n, g.fs, g.frame_period), g.frames_max);
  float* dst = out + (long long)item * g.frames_max + frame;
  const double f0 = frame < F ? g.f0[(long long)item * g.frames_max + frame] : 0.0;
  const double fs = (double)g.fs;
  if (!(f0 > 40.0) || f0 > fs / 12.0 || n < 32) {  // kFloorF0StoneMask
    if (lane == 0) *dst = 0.f;
    return;
  }
  const float* x = audio + (long long)item * g.t_max;
  const double pos = frame * g.frame_period / 1000.0;
  const int half = (int)(1.5 * fs / f0 + 1.0);
  const int nw = 2 * half + 1;
  const double wt = (2.0 * half + 1.0) / fs;
  const int N = 1 << (2 + (int)(log(half * 2.0 + 1.0) / log(2.0)));
  const int base0 = world_round((pos - (double)half / fs) * fs + 0.001);  // GetBaseIndex: base_index[i] = base0 + i
  double* mw = lds + (long long)wave * n_win_max;
  if (nw > n_win_max || N > g.tw_n) {  // (cannot happen for f0 >= f0_floor: the launch sized both from it)
    if (lane == 0) *dst = (float)f0;
    return;
  }
  for (int i = lane; i < nw; i += 64) {
    const double tmp = (base0 + i - 1.0) / fs - pos;
    mw[i] = 0.42 + 0.5 * cos(2.0 * kPiD * tmp / wt) + 0.08 * cos(4.0 * kPiD * tmp / wt);
  }
  __builtin_amdgcn_wave_barrier();
  __threadfence_block();
  const int tstep = g.tw_n / N;
  auto spectra = [&](const int* bins, int nb, double* power, double* numer) {  // |main|^2 and Re(main) Im(diff) - Im(main) Re(diff) at the bins
    double mr[6], mi[6], dr[6], di[6];
#pragma unroll
    for (int b = 0; b < 6; ++b) mr[b] = mi[b] = dr[b] = di[b] = 0.0;
    for (int i = lane; i < nw; i += 64) {
      int idx = base0 + i - 1;
      idx = idx < 0 ? 0 : (idx > n - 1 ? n - 1 : idx);
      const double xv = (double)x[idx];
      const double wm = mw[i];
      const double wd = i == 0 ? -mw[1] / 2.0 : (i == nw - 1 ? mw[nw - 2] / 2.0 : -(mw[i + 1] - mw[i - 1]) / 2.0);
      const double am = xv * wm, ad = xv * wd;
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        if (b < nb) {
          const int m = (int)(((long long)bins[b] * i) & (N - 1)) * tstep;
          const double c = g.twiddle[m], s = g.twiddle[g.tw_n + m];
          mr[b] += am * c;
          mi[b] -= am * s;
          dr[b] += ad * c;
          di[b] -= ad * s;
        }
      }
    }
#pragma unroll
    for (int b = 0; b < 6; ++b) {
      if (b < nb) {
        const double a = wave_sum(mr[b]), bb = wave_sum(mi[b]), c = wave_sum(dr[b]), d = wave_sum(di[b]);
        power[b] = a * a + bb * bb;
        numer[b] = a * d - bb * c;
      }
    }
  };
  auto fix_f0 = [&](double initial, int nh, double* result) {  // stonemask.cpp: FixF0
    int bins[6];
    for (int h = 0; h < nh; ++h) bins[h] = min(world_round(initial * N / fs * (h + 1)), N / 2);
    double power[6], numer[6];
    spectra(bins, nh, power, numer);
    double num = 0.0, den = 0.0;
    for (int h = 0; h < nh; ++h) {
      const double inst = power[h] == 0.0 ? 0.0 : (double)bins[h] * fs / N + numer[h] / power[h] * fs / 2.0 / kPiD;
      const double amp = sqrt(power[h]);
      num += amp * inst;
      den += amp * (h + 1);
    }
    *result = num / (den + kWorldSafeGuard);
  };
  const int n_harm = min((int)(fs / 2.0 / f0), 6);
  double tentative, mean_f0;
  fix_f0(f0, 2, &tentative);
  if (tentative <= 0.0 || tentative > f0 * 2) mean_f0 = 0.0;
  else fix_f0(tentative, n_harm, &mean_f0);
  if (fabs(mean_f0 - f0) > f0 * 0.2) mean_f0 = f0;
  if (lane == 0) *dst = (float)mean_f0;
}

static int dio_geometry(DioGeom& g, int items, int t_max, int fs, int hop, int speed, double f0_floor, double f0_ceil, double channels,
                        double allowed_range) {
  if (items < 1 || t_max < 1 || fs < 1000 || hop < 1 || f0_floor <= 0 || f0_ceil <= f0_floor || channels <= 0)
    return fail(EVMI_ERR_INVALID_ARG, "pitch_world: bad arguments");
  g.items = items; g.t_max = t_max; g.fs = fs;
  g.ratio = std::max(std::min(speed, 12), 1);
  g.actual_fs = (double)fs / g.ratio;
  g.frame_period = (double)hop / fs * 1000.0;
  g.f0_floor = f0_floor; g.f0_ceil = f0_ceil; g.allowed_range = allowed_range;
  g.n_bands = 1 + (int)(std::log(f0_ceil / f0_floor) / std::log(2.0) * channels);
  if (g.n_bands < 1 || g.n_bands > kMaxBands) return fail(EVMI_ERR_UNSUPPORTED, "pitch_world: 1 .. 16 bands");
  for (int i = 0; i < g.n_bands; ++i) {
    g.boundary[i] = f0_floor * std::pow(2.0, (i + 1) / channels);
    g.half[i] = world_round(g.actual_fs / g.boundary[i] / 2.0);
    if (g.half[i] < 1) return fail(EVMI_ERR_UNSUPPORTED, "pitch_world: band above the decimated Nyquist range");
  }
  g.frames_max = (int)(1000.0 * t_max / fs / g.frame_period) + 1;
  g.y_max = 1 + t_max / g.ratio;
  g.c_lowcut = world_round(g.actual_fs / 50.0);
  g.z_len = g.y_max + 2 * g.c_lowcut;
  // FilterForDecimate: the tables of matlabfunctions.cpp are Chebyshev type I, order 3, 0.05 dB, cut-off 0.8 / r (bilinear transform);
  // designed here the same way (oracle/pitch_world_ref.py checks the two sets it knows by heart against scipy's design)
  {
    const double rp = 0.05, wn = 0.8 / g.ratio;
    const double eps = std::sqrt(std::pow(10.0, 0.1 * rp) - 1.0);
    const double mu = std::asinh(1.0 / eps) / 3.0;
    const double warped = 2.0 * 2.0 * std::tan(kPiD * wn / 2.0);  // fs = 2 in scipy's bilinear convention
    // analog poles p_k = -sinh(mu) sin(theta_k) + j cosh(mu) cos(theta_k), theta_k = pi (2k - 1) / 6, k = 1..3, scaled by `warped`
    double pr[3], pi_[3];
    for (int k = 0; k < 3; ++k) {
      const double th = kPiD * (2.0 * (k + 1) - 1.0) / 6.0;
      pr[k] = -std::sinh(mu) * std::sin(th) * warped;
      pi_[k] = std::cosh(mu) * std::cos(th) * warped;
    }
    // analog gain: prod(-p) (odd order: no ripple factor); bilinear with fs2 = 4: z = (fs2 + s) / (fs2 - s)
    const double fs2 = 4.0;
    // digital poles
    double zr[3], zi[3];
    double kr = 1.0, ki = 0.0;  // prod(-p) / prod(fs2 - p)
    for (int k = 0; k < 3; ++k) {
      const double ar = fs2 + pr[k], ai = pi_[k], br = fs2 - pr[k], bi = -pi_[k];
      const double den = br * br + bi * bi;
      zr[k] = (ar * br + ai * bi) / den;
      zi[k] = (ai * br - ar * bi) / den;
      // kr + j ki *= (-p) / (fs2 - p)
      const double nr = -pr[k], ni = -pi_[k];
      const double qr = (nr * br + ni * bi) / den, qi = (ni * br - nr * bi) / den;
      const double tr = kr * qr - ki * qi, ti = kr * qi + ki * qr;
      kr = tr; ki = ti;
    }
    // denominator (z - z0)(z - z1)(z - z2): real coefficients (z1 real, z0 = conj z2)
    const double s1r = zr[0] + zr[1] + zr[2];
    const double s2r = (zr[0] * zr[1] - zi[0] * zi[1]) + (zr[0] * zr[2] - zi[0] * zi[2]) + (zr[1] * zr[2] - zi[1] * zi[2]);
    const double p01r = zr[0] * zr[1] - zi[0] * zi[1], p01i = zr[0] * zi[1] + zi[0] * zr[1];
    const double s3r = p01r * zr[2] - p01i * zi[2];
    g.a[0] = s1r; g.a[1] = -s2r; g.a[2] = s3r;  // y[n] = ... + a0 y[n-1] + a1 y[n-2] + a2 y[n-3]
    g.b[0] = kr; g.b[1] = 3.0 * kr;             // numerator k (1 + z^-1)^3
  }
  return EVMI_OK;
}

static long long dio_ws_layout(DioGeom& g, double* ws, int tw_n) {
  long long off = 0;
  auto take = [&](long long n) {
    double* p = ws ? ws + off : nullptr;
    off += (n + 1) & ~1LL;
    return p;
  };
  const long long I = g.items;
  g.tmp1 = take(I * (g.t_max + 18));
  g.tmp2 = take(I * (g.t_max + 18));
  g.y = take(I * g.y_max);
  g.z = take(I * g.z_len);
  g.filt = take(I * g.n_bands * g.y_max);
  g.edges = take(I * g.n_bands * 4 * g.y_max);
  g.n_edges = reinterpret_cast<int*>(take((I * g.n_bands * 4 + 1) / 2 + 1));
  g.cand = take(I * g.n_bands * g.frames_max);
  g.score = take(I * g.n_bands * g.frames_max);
  g.f0 = take(I * g.frames_max);
  g.scratch = take(I * 4 * g.frames_max);
  g.tw_n = tw_n;
  g.twiddle = take(2LL * tw_n);
  return off;
}

static int stone_sizes(const DioGeom& g, int* n_win_max, int* tw_n) {
  const double f_lo = std::max(40.0, g.f0_floor * 0.99);  // the contour holds values >= f0_floor (or 0)
  const int half = (int)(1.5 * g.fs / f_lo + 1.0);
  *n_win_max = 2 * half + 1;
  *tw_n = 1 << (2 + (int)(std::log(half * 2.0 + 1.0) / std::log(2.0)));
  return EVMI_OK;
}

}  // namespace evmi

using namespace evmi;

extern "C" {

long long evmi_pitch_world_ws_elems(int items, int t_max, int sample_rate, int hop, int speed, float f0_floor, float f0_ceil,
                                    float channels_in_octave) {
  DioGeom g;
  if (dio_geometry(g, items, t_max, sample_rate, hop, speed, f0_floor, f0_ceil, channels_in_octave, 0.1)) return -1;
  int nwin, twn;
  stone_sizes(g, &nwin, &twn);
  return dio_ws_layout(g, nullptr, twn);
}

int evmi_pitch_world_f64(const float* audio_dev, const int* lens_dev, float* f0_dev, double* ws_dev, long long ws_elems, int items, int t_max,
                         int sample_rate, int hop, int speed, float f0_floor, float f0_ceil, float channels_in_octave, float allowed_range,
                         void* stream) {
  if (!audio_dev || !f0_dev || !ws_dev) return fail(EVMI_ERR_INVALID_ARG, "pitch_world: null pointer");
  DioGeom g;
  if (int rc = dio_geometry(g, items, t_max, sample_rate, hop, speed, f0_floor, f0_ceil, channels_in_octave, allowed_range)) return rc;
  int nwin, twn;
  stone_sizes(g, &nwin, &twn);
  if (ws_elems < dio_ws_layout(g, ws_dev, twn) || (reinterpret_cast<uintptr_t>(ws_dev) & 15)) return fail(EVMI_ERR_INVALID_ARG, "pitch_world: workspace too small or unaligned");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(world_twiddle_kernel, dim3((twn + 255) / 256), dim3(256), 0, s, g.twiddle, twn);
  hipLaunchKernelGGL(dio_decimate_kernel, dim3(items), dim3(256), 0, s, g, audio_dev, lens_dev);
  hipLaunchKernelGGL(dio_lowcut_kernel, dim3((g.z_len + 255) / 256, items), dim3(256), (size_t)(2 * g.c_lowcut + 1) * 8, s, g, lens_dev);
  int lmax = 0;
  for (int i = 0; i < g.n_bands; ++i) lmax = std::max(lmax, 4 * g.half[i]);
  hipLaunchKernelGGL(dio_band_kernel, dim3((g.y_max + 255) / 256, g.n_bands, items), dim3(256), (size_t)lmax * 8, s, g, lens_dev);
  hipLaunchKernelGGL(dio_events_kernel, dim3(4, g.n_bands, items), dim3(256), 0, s, g, lens_dev);
  hipLaunchKernelGGL(dio_candidates_kernel, dim3((g.frames_max + 255) / 256, g.n_bands, items), dim3(256), 0, s, g, lens_dev);
  hipLaunchKernelGGL(dio_contour_kernel, dim3((items + 63) / 64), dim3(64), 0, s, g, lens_dev);
  const size_t lds = (size_t)kStoneWaves * nwin * 8;
  if (lds > 64 * 1024) return fail(EVMI_ERR_UNSUPPORTED, "pitch_world: StoneMask window too long for this f0_floor / sample rate");
  hipLaunchKernelGGL(stonemask_kernel, dim3((g.frames_max + kStoneWaves - 1) / kStoneWaves, items), dim3(64 * kStoneWaves), lds, s, g, audio_dev, lens_dev,
                     f0_dev, nwin);
  EVMI_LAUNCH_CHECK("pitch_world");
  return EVMI_OK;
}

/* The decimation filter the device path designs for ratio r: a[3], b[2] (tests: against the tables WORLD prints for r = 11, 12). */
int evmi_pitch_world_decimator(int ratio, double* a3_host, double* b2_host) {
  DioGeom g;
  if (int rc = dio_geometry(g, 1, 4096, 22050, 256, ratio, 71.f, 800.f, 2.f, 0.1)) return rc;
  if (!a3_host || !b2_host) return fail(EVMI_ERR_INVALID_ARG, "pitch_world_decimator: null pointer");
  for (int i = 0; i < 3; ++i) a3_host[i] = g.a[i];
  for (int i = 0; i < 2; ++i) b2_host[i] = g.b[i];
  return EVMI_OK;
}
}
