"""TEST INFRASTRUCTURE (not shipped, not imported by the product): a numpy / float64 restatement of WORLD's DIO and StoneMask F0
estimators -- what the reference calls for its pitch targets:

    everyvoice/preprocessor/preprocessor.py:244-285  (extract_pitch)
        pitch, t = pyworld.dio(audio.f64, fs, frame_period = hop / fs * 1000, speed = 4)
        pitch    = pyworld.stonemask(audio.f64, pitch, t, fs)          # then 0 -> NaN -> linear interpolation

``pyworld-prebuilt 0.3.4.4`` (pyproject.toml of the reference) wraps M. Morise's WORLD (C++) and is NOT in this image: **parity unpinned**
against the binary.  What this file follows is the published algorithm -- Morise, Kawahara, Katayose, "Fast and reliable F0 estimation
method based on the period extraction of vocal fold vibration of singing voice and speech" (AES 35th Int. Conf., 2009) for DIO; Morise,
Kawahara, Nishiura, "Rapid F0 estimation for high-SNR speech based on fundamental component extraction" (IEICE Trans. 2010) for the
StoneMask refinement -- in the form of WORLD's own sources (src/dio.cpp, src/stonemask.cpp, src/matlabfunctions.cpp; every function
below names the routine it restates).  Pins that ARE available here:

  * the decimation filter: WORLD hard-codes one 3rd-order IIR per ratio; the two coefficient sets reproduced in ``_KNOWN_DECIMATORS``
    (ratio 11 and 12, as printed in matlabfunctions.cpp) equal ``scipy.signal.cheby1(3, 0.05, 0.8 / r)`` to 1e-12, which is the rule
    the other ratios are generated from (tests/test_oracle_golden.py checks it);
  * the reference's own fixture of another codebase's pyworld track: everyvoice/tests/data/ming024/eng-LJSpeech-pitch-LJ010-0008.npy
    (phone-level, standardised) against this estimator on everyvoice/tests/data/LJ010-0008.wav: Pearson r >= 0.99 (tests/test_pipeline.py).
"""

from __future__ import annotations

import math

import numpy as np

K_CUTOFF = 50.0            # world::kCutOff
K_MAXIMUM_VALUE = 100000.0  # world::kMaximumValue
K_SAFE_GUARD = 1e-12        # world::kMySafeGuardMinimum
K_FLOOR_F0_STONEMASK = 40.0  # world::kFloorF0StoneMask

# matlabfunctions.cpp: FilterForDecimate, `case 11` and `case 12` (a[0..2], b[0..1]; numerator b0 (1 + 3 z^-1 + 3 z^-2 + z^-3))
_KNOWN_DECIMATORS = {
    11: ((2.450743295230728, -2.06794904601978, 0.59574774438332101), (0.0026822508007163792, 0.0080467524021491377)),
    12: ((2.4981398605924205, -2.1368928194784025, 0.62187513816221485), (0.0021097275904709001, 0.0063291827714127002)),
}


def matlab_round(x: float) -> int:
    """matlabfunctions.cpp: matlab_round (half away from zero)."""
    return int(x + 0.5) if x > 0 else int(x - 0.5)


def decimator_coefficients(r: int):
    """(a[3], b[2]) of FilterForDecimate: the hard-coded tables are Chebyshev type I, order 3, 0.05 dB, cut-off 0.8 / r (bilinear)."""
    if r in _KNOWN_DECIMATORS:
        return _KNOWN_DECIMATORS[r]
    from scipy import signal

    b, a = signal.cheby1(3, 0.05, 0.8 / r)
    return (-a[1], -a[2], -a[3]), (b[0], b[1])


def _filter_for_decimate(x: np.ndarray, r: int) -> np.ndarray:
    """matlabfunctions.cpp: FilterForDecimate (direct form II, state w[3])."""
    from scipy import signal

    (a0, a1, a2), (b0, b1) = decimator_coefficients(r)
    return signal.lfilter([b0, b1, b1, b0], [1.0, -a0, -a1, -a2], x)


def decimate(x: np.ndarray, r: int) -> np.ndarray:
    """matlabfunctions.cpp: decimate -- 9 reflected samples on both sides, the filter forward and backward, every r-th sample."""
    n_fact = 9
    n = len(x)
    head = 2 * x[0] - x[n_fact:0:-1]
    tail = 2 * x[-1] - x[n - 2 : n - 2 - n_fact : -1]
    tmp = np.concatenate([head, x, tail])
    tmp = _filter_for_decimate(tmp, r)[::-1]
    tmp = _filter_for_decimate(tmp, r)[::-1]
    nout = (n - 1) // r + 1
    nbeg = r - r * nout + n
    idx = np.arange(nbeg, n + n_fact, r) + n_fact - 1
    return tmp[idx]


def get_suitable_fft_size(sample: int) -> int:
    """common.cpp: GetSuitableFFTSize."""
    return int(2.0 ** (int(math.log(sample) / math.log(2.0)) + 1.0))


def nuttall_window(n: int) -> np.ndarray:
    """common.cpp: NuttallWindow."""
    t = np.arange(n) / (n - 1.0)
    return 0.355768 - 0.487396 * np.cos(2 * np.pi * t) + 0.144232 * np.cos(4 * np.pi * t) - 0.012604 * np.cos(6 * np.pi * t)


def _design_low_cut_filter(n: int, fft_size: int) -> np.ndarray:
    """dio.cpp: DesignLowCutFilter (delta minus a normalised Hann low-pass, circularly centred on sample 0)."""
    f = np.zeros(fft_size)
    i = np.arange(1, n + 1)
    f[:n] = 0.5 - 0.5 * np.cos(i * 2.0 * np.pi / (n + 1))
    f[:n] = -f[:n] / f[:n].sum()
    half = (n - 1) // 2
    f[fft_size - half :] = f[:half]
    f[:n] = np.concatenate([f[half:n], f[n : n + half]])
    f[0] += 1.0
    return f


def _spectrum_for_estimation(x: np.ndarray, y_length: int, actual_fs: float, fft_size: int, ratio: int) -> np.ndarray:
    """dio.cpp: GetSpectrumForEstimation (decimation, DC removal, 50 Hz low cut)."""
    y = np.zeros(fft_size)
    d = decimate(x, ratio) if ratio != 1 else x
    y[: len(d)] = d
    y[:y_length] -= y[:y_length].mean()
    y[y_length:] = 0.0
    cutoff = matlab_round(actual_fs / K_CUTOFF)
    return np.fft.rfft(y) * np.fft.rfft(_design_low_cut_filter(cutoff * 2 + 1, fft_size))


def _zero_crossing_engine(s: np.ndarray, fs: float):
    """dio.cpp: ZeroCrossingEngine -- negative-going crossings, their sub-sample positions, the intervals between neighbours."""
    n = len(s)
    edges = np.nonzero((s[: n - 1] > 0.0) & (s[1:] <= 0.0))[0] + 1  # (1-based sample numbers, as in the source)
    if len(edges) < 2:
        return np.zeros(0), np.zeros(0)
    fine = edges - s[edges - 1] / (s[edges] - s[edges - 1])
    return (fine[:-1] + fine[1:]) / 2.0 / fs, fs / (fine[1:] - fine[:-1])


def _four_zero_crossing_intervals(filtered: np.ndarray, fs: float):
    """dio.cpp: GetFourZeroCrossingIntervals: negative-going, positive-going zero crossings, peaks, dips."""
    s = filtered.copy()
    out = [_zero_crossing_engine(s, fs), _zero_crossing_engine(-s, fs)]
    d = s[:-1] - s[1:]
    out += [_zero_crossing_engine(d, fs), _zero_crossing_engine(-d, fs)]
    return out


def interp1(x: np.ndarray, y: np.ndarray, xi: np.ndarray) -> np.ndarray:
    """matlabfunctions.cpp: interp1 + histc -- linear, the first / last segment extended beyond the ends."""
    k = np.clip(np.searchsorted(x, xi, side="right"), 1, len(x) - 1)
    s = (xi - x[k - 1]) / (x[k] - x[k - 1])
    return y[k - 1] + s * (y[k] - y[k - 1])


def _candidate_from_raw_event(boundary_f0, fs, y_spectrum, y_length, fft_size, f0_floor, f0_ceil, temporal_positions):
    """dio.cpp: GetF0CandidateFromRawEvent = GetFilteredSignal + GetFourZeroCrossingIntervals + GetF0CandidateContour(Sub)."""
    half = matlab_round(fs / boundary_f0 / 2.0)
    lp = np.zeros(fft_size)
    lp[: half * 4] = nuttall_window(half * 4)
    filtered = np.fft.irfft(y_spectrum * np.fft.rfft(lp), fft_size)
    filtered = filtered[half * 2 : half * 2 + y_length]  # compensation of the delay
    events = _four_zero_crossing_intervals(filtered, fs)
    n = len(temporal_positions)
    if any(len(loc) - 2 <= 0 for loc, _ in events):  # CheckEvent(number - 2) of every kind
        return np.zeros(n), np.full(n, K_MAXIMUM_VALUE)
    sets = np.stack([interp1(loc, itv, temporal_positions) for loc, itv in events])
    cand = sets.sum(0) / 4.0
    score = np.sqrt(((sets - cand) ** 2).sum(0) / 3.0)
    bad = (cand > boundary_f0) | (cand < boundary_f0 / 2.0) | (cand > f0_ceil) | (cand < f0_floor)
    cand[bad] = 0.0
    score[bad] = K_MAXIMUM_VALUE
    return cand, score


def _select_best_f0(current, past, candidates, target, allowed_range):
    """dio.cpp: SelectBestF0."""
    ref = (current * 3.0 - past) / 2.0
    col = candidates[:, target]
    best = col[np.argmin(np.abs(ref - col))]  # (first minimum, as the `<` of the source)
    if abs(1.0 - best / ref) > allowed_range:
        return 0.0
    return best


def _fix_f0_contour(frame_period, candidates, best, f0_floor, allowed_range):
    """dio.cpp: FixF0Contour = FixStep1 .. FixStep4."""
    n = len(best)
    vrm = int(0.5 + 1000.0 / frame_period / f0_floor) * 2 + 1  # voice_range_minimum
    if n <= vrm:
        return np.zeros(n)
    # step 1: no jumps
    base = best.copy()
    base[:vrm] = 0.0
    base[n - vrm :] = 0.0
    s1 = np.zeros(n)
    i = np.arange(vrm, n)
    ok = np.abs((base[i] - base[i - 1]) / (K_SAFE_GUARD + base[i])) < allowed_range
    s1[i] = np.where(ok, base[i], 0.0)
    # step 2: no short voiced runs (every frame within +- centre must be voiced)
    s2 = s1.copy()
    c = (vrm - 1) // 2
    for i in range(c, n - c):
        if np.any(s1[i - c : i + c + 1] == 0):
            s2[i] = 0.0
    # voiced sections
    pos = [i for i in range(1, n) if s2[i - 1] == 0 and s2[i] != 0]
    neg = [i - 1 for i in range(1, n) if s2[i] == 0 and s2[i - 1] != 0]
    # step 3: extend every section forward
    s3 = s2.copy()
    for k, start in enumerate(neg):
        limit = n - 1 if k == len(neg) - 1 else neg[k + 1]
        for j in range(start, limit):
            s3[j + 1] = _select_best_f0(s3[j], s3[j - 1], candidates, j + 1, allowed_range)
            if s3[j + 1] == 0:
                break
    # step 4: and backward
    s4 = s3.copy()
    for k in range(len(pos) - 1, -1, -1):
        limit = 1 if k == 0 else pos[k - 1]
        for j in range(pos[k], limit, -1):
            s4[j - 1] = _select_best_f0(s4[j], s4[j + 1], candidates, j - 1, allowed_range)
            if s4[j - 1] == 0:
                break
    return s4


def dio(x: np.ndarray, fs: int, f0_floor: float = 71.0, f0_ceil: float = 800.0, channels_in_octave: float = 2.0, frame_period: float = 5.0,
        speed: int = 1, allowed_range: float = 0.1, return_candidates: bool = False):
    """dio.cpp: Dio / DioGeneralBody -> (f0 [frames], temporal_positions [frames]); the signature of ``pyworld.dio``."""
    x = np.asarray(x, dtype=np.float64)
    n_bands = 1 + int(math.log(f0_ceil / f0_floor) / math.log(2.0) * channels_in_octave)
    boundary = f0_floor * 2.0 ** ((np.arange(n_bands) + 1) / channels_in_octave)
    ratio = max(min(int(speed), 12), 1)
    y_length = 1 + len(x) // ratio
    actual_fs = fs / ratio
    fft_size = get_suitable_fft_size(y_length + matlab_round(actual_fs / K_CUTOFF) * 2 + 1 + 4 * int(1.0 + actual_fs / boundary[0] / 2.0))
    y_spectrum = _spectrum_for_estimation(x, y_length, actual_fs, fft_size, ratio)
    f0_length = int(1000.0 * len(x) / fs / frame_period) + 1  # GetSamplesForDIO
    t = np.arange(f0_length) * frame_period / 1000.0
    cands, scores = [], []
    for b in boundary:
        c, s = _candidate_from_raw_event(b, actual_fs, y_spectrum, y_length, fft_size, f0_floor, f0_ceil, t)
        cands.append(c)
        scores.append(s / (c + K_SAFE_GUARD))
    cands, scores = np.stack(cands), np.stack(scores)
    best = cands[np.argmin(scores, axis=0), np.arange(f0_length)]  # GetBestF0Contour (first minimum)
    f0 = _fix_f0_contour(frame_period, cands, best, f0_floor, allowed_range)
    if return_candidates:
        return f0, t, cands, scores, best
    return f0, t


def _fix_f0(power, numerator, fft_size, fs, initial_f0, n_harmonics):
    """stonemask.cpp: FixF0 -- amplitude-weighted mean of the harmonics' instantaneous frequencies."""
    num = den = 0.0
    for i in range(n_harmonics):
        idx = min(matlab_round(initial_f0 * fft_size / fs * (i + 1)), fft_size // 2)
        inst = 0.0 if power[idx] == 0.0 else idx * fs / fft_size + numerator[idx] / power[idx] * fs / 2.0 / np.pi
        amp = math.sqrt(power[idx])
        num += amp * inst
        den += amp * (i + 1)
    return num / (den + K_SAFE_GUARD)


def _refined_f0(x, fs, position, initial_f0):
    """stonemask.cpp: GetRefinedF0 / GetMeanF0."""
    if initial_f0 <= K_FLOOR_F0_STONEMASK or initial_f0 > fs / 12.0:
        return 0.0
    half = int(1.5 * fs / initial_f0 + 1.0)
    window_time = (2.0 * half + 1.0) / fs
    n = half * 2 + 1
    fft_size = int(2.0 ** (2.0 + int(math.log(half * 2.0 + 1.0) / math.log(2.0))))
    base_index = matlab_round((position - half / fs) * fs + 0.001) + np.arange(n)  # GetBaseIndex
    tmp = (base_index - 1.0) / fs - position
    main = 0.42 + 0.5 * np.cos(2.0 * np.pi * tmp / window_time) + 0.08 * np.cos(4.0 * np.pi * tmp / window_time)  # GetMainWindow
    diff = np.empty(n)                                                                                              # GetDiffWindow
    diff[0] = -main[1] / 2.0
    diff[1:-1] = -(main[2:] - main[:-2]) / 2.0
    diff[-1] = main[-2] / 2.0
    seg = x[np.clip(base_index - 1, 0, len(x) - 1)]
    ms, ds = np.fft.rfft(seg * main, fft_size), np.fft.rfft(seg * diff, fft_size)
    numerator = ms.real * ds.imag - ms.imag * ds.real
    power = ms.real ** 2 + ms.imag ** 2
    n_harm = min(int(fs / 2.0 / initial_f0), 6)
    tentative = _fix_f0(power, numerator, fft_size, fs, initial_f0, 2)
    if tentative <= 0.0 or tentative > initial_f0 * 2:  # "if the fixed value is too large, the result will be rejected"
        mean_f0 = 0.0
    else:
        mean_f0 = _fix_f0(power, numerator, fft_size, fs, tentative, n_harm)
    if abs(mean_f0 - initial_f0) > initial_f0 * 0.2:  # a correction above 20 %: the initial value stands
        mean_f0 = initial_f0
    return mean_f0


def stonemask(x: np.ndarray, f0: np.ndarray, temporal_positions: np.ndarray, fs: int) -> np.ndarray:
    """stonemask.cpp: StoneMask; the signature of ``pyworld.stonemask``."""
    x = np.asarray(x, dtype=np.float64)
    return np.array([_refined_f0(x, fs, t, f) for t, f in zip(temporal_positions, f0)])


def extract_pitch_ref(audio: np.ndarray, fs: int, hop: int, speed: int = 4) -> np.ndarray:
    """Preprocessor.extract_pitch (preprocessor.py:244-285): dio(speed = 4) -> stonemask -> unvoiced frames linearly interpolated
    (an utterance without a voiced frame: zeros)."""
    x = np.asarray(audio, dtype=np.float64)
    f0, t = dio(x, fs, frame_period=hop / fs * 1000.0, speed=speed)
    f0 = stonemask(x, f0, t, fs)
    voiced = f0 > 0
    if not voiced.any():
        return np.zeros_like(f0)
    out = f0.copy()
    out[~voiced] = np.interp(np.nonzero(~voiced)[0], np.nonzero(voiced)[0], f0[voiced])
    return out
