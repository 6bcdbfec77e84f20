// iSTFTNet output head: reflection pad (1,0) -> conv_post (C -> n_fft+2, k7) -> exp / sin -> inverse STFT
// (n_fft 16, hop 4, periodic hann, centred) -> waveform.  One fused kernel on the bf16 path; small fp32
// kernels for the exact-arithmetic path.
//
//   z[f][c]   = b[c] + sum_{j<7} sum_ci W[c][ci][j] * xpad[f + j - 3][ci],   xpad[0] = x[1], xpad[q] = x[q-1]
//   mag[f][b] = exp(z[f][b]),  phi[f][b] = sin(z[f][9 + b]),   b in [0, 8]
//   y_f[n]    = w[n] / 16 * ( mag0 cos(phi0) + (-1)^n mag8 cos(phi8) + 2 sum_{b=1..7} mag_b cos(phi_b + 2 pi b n / 16) )
//   wav[t]    = sum_f y_f[t + 8 - 4 f] / sum_f w^2[t + 8 - 4 f]            (frames f in [0, L], 0 <= t+8-4f < 16)
// which is torch.istft(mag * exp(i*phi), 16, hop_length=4, win_length=16, window=hann, center=True).
#include "common.h"

namespace evmi {

constexpr int IS_NFFT = 16, IS_HOP = 4, IS_BINS = 9, IS_KS = 7;
constexpr int IS_FRAMES = 256;                       // frames per workgroup
constexpr int IS_SAMPLES = IS_HOP * (IS_FRAMES - 3);  // output samples per workgroup

__device__ __forceinline__ float hann16(int n) { return 0.5f - 0.5f * cosf(6.283185307179586f * n / IS_NFFT); }

template <int C>
__global__ __launch_bounds__(512) void istft_head_kernel(const bf16_t* __restrict__ x,   // [B][L][C], activation applied
                                                         const bf16_t* __restrict__ w,   // [7][32][C], rows >= 18 zero
                                                         const float* __restrict__ bias, // [32]
                                                         float* __restrict__ wav,        // [B][4 L]
                                                         int L) {
  constexpr int S = C + 8;
  constexpr int XROWS = IS_FRAMES + IS_KS - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* Xs = reinterpret_cast<bf16_t*>(smem);            // [XROWS][S]
  bf16_t* Ws = Xs + XROWS * S;                             // [7][32][S]
  float* Zs = reinterpret_cast<float*>(smem);              // after the MMA: [IS_FRAMES][20]  (m cos phi | m sin phi)
  float* Ys = Zs + IS_FRAMES * 20;                         // [IS_FRAMES][17] windowed time frames
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
  const int T0 = blockIdx.x * IS_SAMPLES;  // first output sample of this workgroup
  const int F0 = T0 / IS_HOP - 1;          // first frame it needs
  const int n_frames = L + 1;
  const bf16_t* xb = x + (long long)b * L * C;

  for (int v = tid; v < XROWS * (C / 8); v += 512) {
    const int i = v / (C / 8), c8 = v % (C / 8);
    const int q = F0 - 3 + i;  // index into the reflection-padded sequence, valid 0 .. L
    bf16x8 val;
#pragma unroll
    for (int e = 0; e < 8; ++e) val[e] = (bf16_t)0.f;
    if (q >= 0 && q <= L) {
      const int src = q == 0 ? (L > 1 ? 1 : 0) : q - 1;
      val = *reinterpret_cast<const bf16x8*>(xb + (long long)src * C + c8 * 8);
    }
    *reinterpret_cast<bf16x8*>(Xs + i * S + c8 * 8) = val;
  }
  for (int v = tid; v < IS_KS * 32 * (C / 8); v += 512) {
    const int row = v / (C / 8), c8 = v % (C / 8);
    *reinterpret_cast<bf16x8*>(Ws + row * S + c8 * 8) = *reinterpret_cast<const bf16x8*>(w + (long long)row * C + c8 * 8);
  }
  __syncthreads();

  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  mma_tap_group<1, 1, C / 16, IS_KS, 32 * S, 0, 0>(Ws + (lane & 31) * S + (lane >> 5) * 8,
                                                    Xs + (wn * 32 + (lane & 31)) * S + (lane >> 5) * 8, S, acc);
  __syncthreads();  // operand tiles are dead: Zs / Ys alias them

  {  // z -> (m cos phi, m sin phi) per bin; a frame's 18 channels are split over lanes l and l + 32
    const int n = wn * 32 + (lane & 31);
    const int f = F0 + n;
    const bool valid = f >= 0 && f < n_frames;
    float z[16];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 8 * q + 4 * (lane >> 5) + i;
        z[4 * q + i] = acc[0][0][4 * q + i] + bias[c];
      }
    // channel c of this lane's register (q, i) is 8q + 4*(lane>>5) + i: magnitudes are c in [0, 8], angles c in [9, 17];
    // stage raw z first, then combine (mag_b, phi_b) pairs that live in different registers / lanes
    float* zrow = Ys + n * 20;  // temporary use of the Ys region as [IS_FRAMES][20] raw z (17-stride Ys written later)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = 8 * q + 4 * (lane >> 5) + i;
        if (c < 18) zrow[c] = valid ? z[4 * q + i] : 0.f;
      }
    (void)valid;
  }
  __syncthreads();
  for (int v = tid; v < IS_FRAMES * IS_BINS; v += 512) {
    const int n = v / IS_BINS, bb = v % IS_BINS;
    const int f = F0 + n;
    const float* zrow = Ys + n * 20;
    float mc = 0.f, ms = 0.f;
    if (f >= 0 && f < n_frames) {
      const float m = expf(zrow[bb]);
      const float phi = sinf(zrow[IS_BINS + bb]);
      mc = m * cosf(phi);
      ms = m * sinf(phi);
    }
    Zs[n * 20 + bb] = mc;
    Zs[n * 20 + 10 + bb] = ms;
  }
  __syncthreads();
  // windowed inverse real DFT of every frame: 16 samples each (Ys re-used with stride 17 after the barrier below)
  float yv[(IS_FRAMES * IS_NFFT) / 512];
#pragma unroll
  for (int it = 0; it < (IS_FRAMES * IS_NFFT) / 512; ++it) {
    const int v = tid + it * 512;
    const int n = v / IS_NFFT, k = v % IS_NFFT;
    const float* zr = Zs + n * 20;
    float s = zr[0] + ((k & 1) ? -zr[8] : zr[8]);
#pragma unroll
    for (int bb = 1; bb < 8; ++bb) {
      const float th = 6.283185307179586f * ((bb * k) % IS_NFFT) / IS_NFFT;
      s += 2.f * (zr[bb] * cosf(th) - zr[10 + bb] * sinf(th));
    }
    yv[it] = s * hann16(k) * (1.f / IS_NFFT);
  }
  __syncthreads();  // raw z rows (in Ys) fully consumed before Ys is overwritten with stride 17
#pragma unroll
  for (int it = 0; it < (IS_FRAMES * IS_NFFT) / 512; ++it) {
    const int v = tid + it * 512;
    Ys[(v / IS_NFFT) * 17 + (v % IS_NFFT)] = yv[it];
  }
  __syncthreads();
  // overlap-add + envelope normalisation
  const int n_out = IS_HOP * L;
  for (int v = tid; v < IS_SAMPLES; v += 512) {
    const int t = T0 + v;
    if (t >= n_out) break;
    const int p = t + IS_NFFT / 2;
    float s = 0.f, env = 0.f;
#pragma unroll
    for (int d = 0; d < IS_NFFT / IS_HOP; ++d) {
      const int f = p / IS_HOP - d;
      const int k = p - IS_HOP * f;  // in [0, 16)
      if (f >= 0 && f < n_frames) {
        s += Ys[(f - F0) * 17 + k];
        const float wk = hann16(k);
        env += wk * wk;
      }
    }
    wav[(long long)b * n_out + t] = s / env;
  }
}

// ---- fp32 exact path pieces ---------------------------------------------------------------------
// xp[b][c][0] = lrelu(x[b][c][1]), xp[b][c][q] = lrelu(x[b][c][q-1])   ([B][C][L] -> [B][C][L+1])
__global__ void reflect_pad_left1_f32_kernel(const float* __restrict__ x, float* __restrict__ xp, int L, float slope,
                                             long long n_rows) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * (L + 1)) return;
  const long long row = idx / (L + 1);
  const int q = (int)(idx % (L + 1));
  const int src = q == 0 ? (L > 1 ? 1 : 0) : q - 1;
  const float v = x[row * L + src];
  xp[idx] = v > 0.f ? v : v * slope;
}

// z [B][18][F] (conv_post output) -> wav [B][4 (F-1)]
__global__ void istft_f32_kernel(const float* __restrict__ z, float* __restrict__ wav, int F) {
  const int n_out = IS_HOP * (F - 1);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int b = blockIdx.y;
  if (t >= n_out) return;
  const float* zb = z + (long long)b * 2 * IS_BINS * F;
  const int p = t + IS_NFFT / 2;
  float s = 0.f, env = 0.f;
  for (int d = 0; d < IS_NFFT / IS_HOP; ++d) {
    const int f = p / IS_HOP - d;
    const int k = p - IS_HOP * f;
    if (f < 0 || f >= F) continue;
    float acc = 0.f;
    for (int bb = 0; bb < IS_BINS; ++bb) {
      const float m = expf(zb[(long long)bb * F + f]);
      const float phi = sinf(zb[(long long)(IS_BINS + bb) * F + f]);
      const float th = 6.283185307179586f * ((bb * k) % IS_NFFT) / IS_NFFT;
      const float re = m * cosf(phi), im = m * sinf(phi);
      if (bb == 0) acc += re;
      else if (bb == IS_BINS - 1) acc += (k & 1) ? -re : re;
      else acc += 2.f * (re * cosf(th) - im * sinf(th));
    }
    const float wk = hann16(k);
    s += acc * wk * (1.f / IS_NFFT);
    env += wk * wk;
  }
  wav[(long long)b * n_out + t] = s / env;
}

int launch_istft_head(const bf16_t* x, const bf16_t* w, const float* bias, float* wav, int B, int L, int c_in,
                      hipStream_t s) {
  const int n_out = IS_HOP * L;
  dim3 grid((n_out + IS_SAMPLES - 1) / IS_SAMPLES, B);
  auto lds_of = [](int c) { return (size_t)((IS_FRAMES + IS_KS - 1) * (c + 8) + IS_KS * 32 * (c + 8)) * 2; };
  static thread_local bool configured_dev[kMaxDevices][3] = {};
  bool* configured = configured_dev[device_slot()];
#define EVMI_ISTFT_CASE(CC, IDX)                                                                                  \
  if (c_in == CC) {                                                                                               \
    const size_t lds = lds_of(CC) > (size_t)IS_FRAMES * 40 * 4 ? lds_of(CC) : (size_t)IS_FRAMES * 40 * 4;        \
    if (!configured[IDX]) {                                                                                       \
      EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)istft_head_kernel<CC>,                                      \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                  \
      configured[IDX] = true;                                                                                     \
    }                                                                                                             \
    hipLaunchKernelGGL((istft_head_kernel<CC>), grid, dim3(512), lds, s, x, w, bias, wav, L);                     \
    EVMI_LAUNCH_CHECK("istft_head");                                                                              \
    return EVMI_OK;                                                                                               \
  }
  EVMI_ISTFT_CASE(32, 0)
  EVMI_ISTFT_CASE(64, 1)
  EVMI_ISTFT_CASE(128, 2)
#undef EVMI_ISTFT_CASE
  return fail(EVMI_ERR_UNSUPPORTED, "istft_head: unsupported channel count");
}

int launch_reflect_pad_left1_f32(const float* x, float* xp, long long n_rows, int L, float slope, hipStream_t s) {
  const long long n = n_rows * (L + 1);
  hipLaunchKernelGGL(reflect_pad_left1_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, xp, L, slope, n_rows);
  EVMI_LAUNCH_CHECK("reflect_pad_left1_f32");
  return EVMI_OK;
}

int launch_istft_f32(const float* z, float* wav, int B, int F, hipStream_t s) {
  const int n_out = IS_HOP * (F - 1);
  hipLaunchKernelGGL(istft_f32_kernel, dim3((n_out + 255) / 256, B), dim3(256), 0, s, z, wav, F);
  EVMI_LAUNCH_CHECK("istft_f32");
  return EVMI_OK;
}

}  // namespace evmi
