// STFT -> magnitude -> mel -> log front-end of the reference's default "mel-librosa" spectrogram
// (everyvoice/utils/heavy.py:69-100 and :39-40), one kernel, fp32 throughout.
//
//   frames[f][k] = audio_reflect_padded[f*hop + k] * hann[k]          (center=True, pad_mode="reflect")
//   re/im[f][b]  = sum_k frames[f][k] * cos/sin(2*pi*b*k/n_fft)        b in [0, n_fft/2]
//   mag          = sqrt(re^2 + im^2 + 1e-9)
//   mel[m][f]    = sum_b basis[m][b] * mag[f][b]                        (librosa Slaney basis, given by the host)
//   out          = log(max(mel, 1e-5))                                 (optional)
//
// The DFT is a GEMM [32 frames x n_fft] x [n_fft x 2*(n_fft/2+1)] on the fp32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains, 157 TFLOP/s class) against a window-folded basis
// held in HBM/L2 (n_fft=1024: 4.2 MB).  The frame matrix is never materialised: the padded audio
// segment of a 32-frame block sits once in LDS with a skew of one word per hop (index s + s/hop),
// which makes the 32 lanes of an A-fragment read (stride hop) hit 32 distinct banks.
#include "common.h"

namespace evmi {

constexpr int MEL_FRAMES = 32;   // frames per workgroup (one MFMA M tile)
constexpr int MEL_THREADS = 512;

__device__ __forceinline__ int reflect_index(int i, int n) {
  // torch "reflect" padding (no edge repeat); valid for |overshoot| < n
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i;
}

// basis_ri: [n_fft][2*NB] fp32, column 2b = hann[k]*cos(2 pi b k / n_fft), column 2b+1 = -hann[k]*sin(...)
//           (NB = n_fft/2 + 1 padded up to a multiple of 16 -> 2*NB multiple of 32, extra columns zero)
// melb:     [n_mels][n_bins] fp32
// out:      [B][n_mels][n_frames] fp32 (torch layout)      energy: [B][n_frames] or nullptr
// The bins are processed in `chunk_tiles` column tiles (16 bins each) at a time so that the magnitude tile fits the LDS next to
// the audio segment at n_fft = 2048 (BASELINE config 5: 44.1 kHz, hop 512); the mel accumulators live in registers across the
// chunks and walk the bins in ascending order whatever the chunking, so the result does not depend on it.
constexpr int MEL_ROWS_PER_THREAD = 8;  // n_mels <= 16 * 8

__global__ __launch_bounds__(MEL_THREADS) void mel_frontend_kernel(
    const float* __restrict__ audio, const float* __restrict__ basis_ri, const float* __restrict__ melb,
    float* __restrict__ out, float* __restrict__ energy, float* __restrict__ mag_out, int n_samples_max, int n_frames,
    int n_fft, int hop, int nb_pad, int n_bins, int n_mels, int apply_log, int chunk_tiles, const int* __restrict__ lens) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int seg = (MEL_FRAMES - 1) * hop + n_fft;           // audio samples a block needs
  float* As = reinterpret_cast<float*>(smem);                // skewed: index s + s / hop
  const int as_words = seg + seg / hop + 1;
  float* Ls = As + ((as_words + 3) & ~3);                    // log-mel rows [MEL_FRAMES][n_mels + 1] for the energy reduction
  float* Ms = Ls + ((MEL_FRAMES * (n_mels + 1) + 3) & ~3);   // magnitudes of one bin chunk [MEL_FRAMES][16 * chunk_tiles + 1]
  const int ms_stride = 16 * chunk_tiles + 1;

  const int b = blockIdx.y;
  const int f0 = blockIdx.x * MEL_FRAMES;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // ragged batches: item b holds lens[b] samples of its n_samples_max-long row and reflects at ITS end; frames past its own
  // 1 + lens[b] / hop are computed from whatever follows (finite) and are the caller's to ignore
  const int n_samples = lens ? max(min(lens[b], n_samples_max), n_fft / 2 + 1) : n_samples_max;
  const float* ab = audio + (long long)b * n_samples_max;
  const int pad = n_fft / 2;

  for (int s = tid; s < seg; s += MEL_THREADS) {
    const int g = f0 * hop + s - pad;  // position in the un-padded signal
    // positions beyond the reflect-padded signal only feed frames >= n_frames (never stored)
    const float v = (g >= -pad && g <= n_samples - 1 + pad) ? ab[reflect_index(g, n_samples)] : 0.f;
    As[s + s / hop] = v;
  }
  __syncthreads();

  const int n_col_tiles = (2 * nb_pad) / 32;
  const int i = lane & 31, kh = lane >> 5;
  const int a_base = i * hop + i;  // skewed start of frame i: (i*hop) + (i*hop)/hop
  const int fr = tid & 31, mg = tid >> 5;  // mel projection: thread -> (frame, mel rows mg, mg + 16, ...)
  float macc[MEL_ROWS_PER_THREAD];
#pragma unroll
  for (int j = 0; j < MEL_ROWS_PER_THREAD; ++j) macc[j] = 0.f;

  for (int ct0 = 0; ct0 < n_col_tiles; ct0 += chunk_tiles) {
    const int ct1 = min(ct0 + chunk_tiles, n_col_tiles);
    const int bin0 = ct0 * 16, bin1 = min(ct1 * 16, n_bins);
    // ---- DFT on the fp32 matrix cores: wave w takes column tiles ct0 + w, ct0 + w + 8, ... of the (re, im) basis ------
    for (int ct = ct0 + wave; ct < ct1; ct += MEL_THREADS / 64) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const float* bcol = basis_ri + ct * 32 + i;  // column of this lane
      const int ld = 2 * nb_pad;
#pragma unroll 8
      for (int k = 0; k < n_fft; k += 2) {
        const int kk = k + kh;
        // frame i, sample kk: skewed index a_base + kk + kk / hop  (the window is folded into the basis)
        const float av = As[a_base + kk + kk / hop];
        const float bv = bcol[(long long)kk * ld];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
      }
      // D layout: lane holds column j = lane&31 (even = re, odd = im of bin ct*16 + j/2) for frames
      // (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float sq = acc[r] * acc[r];
        const float other = __shfl_xor(sq, 1, 64);
        const int frame = (r & 3) + 8 * (r >> 2) + 4 * kh;
        const int bin = ct * 16 + (i >> 1);
        if (!(i & 1) && bin < n_bins) Ms[frame * ms_stride + bin - bin0] = sqrtf(sq + other + 1e-9f);
      }
    }
    __syncthreads();

    if (mag_out) {  // optional linear-magnitude output [B][n_bins][n_frames]
      for (int v = tid; v < MEL_FRAMES * (bin1 - bin0); v += MEL_THREADS) {
        const int bin = bin0 + v / MEL_FRAMES, f = v % MEL_FRAMES;
        if (f0 + f < n_frames) mag_out[((long long)b * n_bins + bin) * n_frames + f0 + f] = Ms[f * ms_stride + bin - bin0];
      }
    }
    // ---- mel projection, this chunk's bins (ascending-bin fmaf chain continued from the previous chunk) ----------
    {
      const float* mrow = Ms + fr * ms_stride - bin0;
      const float* wbase = melb + (long long)mg * n_bins;
      const long long wstep = (long long)(MEL_THREADS / 32) * n_bins;
      const int rows = (n_mels - mg + MEL_THREADS / 32 - 1) / (MEL_THREADS / 32);  // mel rows of this thread
      for (int bin = bin0; bin < bin1; ++bin) {
        const float mv = mrow[bin];
#pragma unroll
        for (int j = 0; j < MEL_ROWS_PER_THREAD; ++j)
          if (j < rows) macc[j] = fmaf(wbase[j * wstep + bin], mv, macc[j]);
      }
    }
    __syncthreads();  // the next chunk overwrites Ms
  }
#pragma unroll
  for (int j = 0; j < MEL_ROWS_PER_THREAD; ++j) {
    const int m = mg + j * (MEL_THREADS / 32);
    if (m < n_mels) {
      float acc = macc[j];
      if (apply_log) acc = logf(fmaxf(acc, 1e-5f));
      if (f0 + fr < n_frames) out[((long long)b * n_mels + m) * n_frames + f0 + fr] = acc;
      Ls[fr * (n_mels + 1) + m] = acc;
    }
  }
  if (energy) {  // energy[f] = || mel[:, f] ||_2 over the (log-)mel bins (everyvoice/preprocessor/preprocessor.py:302-309)
    __syncthreads();
    if (tid < MEL_FRAMES && f0 + tid < n_frames) {
      float e = 0.f;
      for (int m = 0; m < n_mels; ++m) e = fmaf(Ls[tid * (n_mels + 1) + m], Ls[tid * (n_mels + 1) + m], e);
      energy[(long long)b * n_frames + f0 + tid] = sqrtf(e);
    }
  }
}

int launch_mel_frontend(const float* audio, const float* basis_ri, const float* melb, float* out, float* energy,
                        float* mag_out, int B, int n_samples, int n_frames, int n_fft, int hop, int nb_pad,
                        int n_bins, int n_mels, int apply_log, hipStream_t s, const int* lens) {
  if (n_fft % hop || n_fft % 2 || hop <= 0) return fail(EVMI_ERR_UNSUPPORTED, "mel: n_fft must be a multiple of hop");
  if (n_samples <= n_fft / 2) return fail(EVMI_ERR_INVALID_ARG, "mel: reflect padding needs n_samples > n_fft/2");
  if (n_mels > MEL_ROWS_PER_THREAD * (MEL_THREADS / 32)) return fail(EVMI_ERR_UNSUPPORTED, "mel: n_mels > 128");
  const int seg = (MEL_FRAMES - 1) * hop + n_fft;
  const size_t as_words = (seg + seg / hop + 1 + 3) & ~3;
  const size_t ls_words = (MEL_FRAMES * (n_mels + 1) + 3) & ~3;
  const size_t budget = 160 * 1024 / sizeof(float);
  const int n_col_tiles = (2 * nb_pad) / 32;
  if (as_words + ls_words + (size_t)MEL_FRAMES * 17 > budget) return fail(EVMI_ERR_UNSUPPORTED, "mel: n_fft / hop too large for the LDS tile");
  // the largest bin chunk that fits beside the audio segment, then the chunks evened out
  int max_tiles = (int)(((budget - as_words - ls_words) / MEL_FRAMES - 1) / 16);
  if (max_tiles > n_col_tiles) max_tiles = n_col_tiles;
  const int n_chunks = (n_col_tiles + max_tiles - 1) / max_tiles;
  const int chunk_tiles = (n_col_tiles + n_chunks - 1) / n_chunks;
  const size_t lds = (as_words + ls_words + (size_t)MEL_FRAMES * (16 * chunk_tiles + 1)) * sizeof(float);
  int dev = 0;
  EVMI_HIP_CHECK(hipGetDevice(&dev));
  static thread_local size_t configured[16] = {0};  // per device: the attribute belongs to the device's code object
  if (dev < 0 || dev >= 16 || lds > configured[dev]) {
    EVMI_HIP_CHECK(hipFuncSetAttribute((const void*)mel_frontend_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (dev >= 0 && dev < 16) configured[dev] = lds;
  }
  dim3 grid((n_frames + MEL_FRAMES - 1) / MEL_FRAMES, B);
  hipLaunchKernelGGL(mel_frontend_kernel, grid, dim3(MEL_THREADS), lds, s, audio, basis_ri, melb, out, energy, mag_out,
                     n_samples, n_frames, n_fft, hop, nb_pad, n_bins, n_mels, apply_log, chunk_tiles, lens);
  EVMI_LAUNCH_CHECK("mel_frontend");
  return EVMI_OK;
}

}  // namespace evmi
