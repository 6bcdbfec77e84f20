"""numpy restatement of the in-tree tensor utilities of the hot path.  TEST INFRASTRUCTURE.

PINNED: every function here is checked in ``tests/test_oracle_golden.py`` against golden
vectors produced by importing the reference's own ``everyvoice.utils.heavy`` /
``everyvoice.preprocessor`` (``tests/golden/make_golden.py``).

Reference lines each function follows (relative to /root/reference):
  expand_ref            everyvoice/utils/heavy.py:12-21
  collate_ref           everyvoice/utils/heavy.py:24-36 (+ _flatten, everyvoice/utils/__init__.py:121-133)
  drc_ref / drd_ref     everyvoice/utils/heavy.py:39-44
  get_segments_ref      everyvoice/utils/heavy.py:122-148
  energy_ref            everyvoice/preprocessor/preprocessor.py:302-309
  average_by_durations_ref  everyvoice/preprocessor/preprocessor.py:287-300
  scaler_stats_ref      everyvoice/preprocessor/helpers.py:47-106
"""

from __future__ import annotations

import numpy as np


def expand_ref(values: np.ndarray, durations) -> np.ndarray:
    """Length regulator primitive: row i of ``values`` repeated max(0, int(d_i)) times.

    ``int(d)`` truncates toward zero, so fractional durations floor for d>0 and negative
    durations give no rows.  (The reference raises on an all-zero duration vector for tensors —
    ``torch.stack([])``; here that is an empty [0, ...] array and the caller decides.)
    """
    values = np.asarray(values)
    reps = np.array([max(0, int(d)) for d in np.asarray(durations).tolist()], dtype=np.int64)
    n = min(len(values), len(reps))  # zip() semantics
    return np.repeat(values[:n], reps[:n], axis=0)


def expand_index_ref(durations) -> np.ndarray:
    """The gather index expand() implies: out[t] = values[idx[t]]."""
    reps = np.array([max(0, int(d)) for d in np.asarray(durations).tolist()], dtype=np.int64)
    return np.repeat(np.arange(len(reps), dtype=np.int64), reps)


def length_regulate_batch_ref(values: np.ndarray, durations: np.ndarray, max_len: int | None = None):
    """expand() per batch item, zero-padded to the batch max (how FastSpeech2's length regulator
    uses the primitive).  values [B, L, D], durations [B, L] int ->
    (out [B, Tmax, D], lengths [B])."""
    outs = [expand_ref(v, d) for v, d in zip(values, durations)]
    lens = np.array([o.shape[0] for o in outs], dtype=np.int64)
    tmax = int(lens.max()) if max_len is None else int(max_len)
    out = np.zeros((len(outs), tmax) + values.shape[2:], dtype=values.dtype)
    for b, o in enumerate(outs):
        n = min(tmax, o.shape[0])
        out[b, :n] = o[:n]
    return out, np.minimum(lens, tmax)


def _flatten(structure, key="", path="", flattened=None):
    if flattened is None:
        flattened = {}
    if not isinstance(structure, dict):
        flattened[(f"{path}_" if path else "") + key] = structure
    else:
        for new_key, value in structure.items():
            _flatten(value, new_key, (f"{path}_" if path else "") + key, flattened)
    return flattened


def collate_ref(data: list[dict]) -> dict:
    """Zero-pad batcher.  Arrays are padded along dim 0 to the longest (pad_sequence,
    batch_first, padding_value 0); python ints become an int32 vector; anything else stays a list."""
    data = [_flatten(x) for x in data]
    out = {k: [d[k] for d in data] for k in data[0]}
    for key, items in out.items():
        if isinstance(items[0], np.ndarray):
            longest = max(x.shape[0] for x in items)
            padded = np.zeros((len(items), longest) + items[0].shape[1:], dtype=items[0].dtype)
            for i, x in enumerate(items):
                padded[i, : x.shape[0]] = x
            out[key] = padded
        elif isinstance(items[0], int) and not isinstance(items[0], bool):
            out[key] = np.asarray(items, dtype=np.int32)
    return out


def drc_ref(x: np.ndarray, C: float = 1.0, clip_val: float = 1e-5) -> np.ndarray:
    x = np.asarray(x, dtype=np.float32)
    return np.log(np.maximum(x, np.float32(clip_val)) * np.float32(C)).astype(np.float32)


def drd_ref(x: np.ndarray, C: float = 1.0) -> np.ndarray:
    return (np.exp(np.asarray(x, dtype=np.float32)) / np.float32(C)).astype(np.float32)


def get_segments_ref(t: np.ndarray, segment_size: int, start: int):
    """Deterministic branch of get_segments (explicit ``start``); the random branch draws
    ``random.randint(0, len - seg - 1)`` and then does the same slice."""
    t_len = t.shape[1]
    if t_len >= segment_size:
        max_start = t_len - segment_size - 1
        assert start <= max_start
        return t[:, start : start + segment_size], start
    pad = [(0, 0), (0, segment_size - t_len)] + [(0, 0)] * (t.ndim - 2)
    return np.pad(t, pad), 0


def energy_ref(logmel: np.ndarray) -> np.ndarray:
    """L2 norm over the mel bins of the LOG-mel: [n_mels, T] -> [T]."""
    x = np.asarray(logmel, dtype=np.float32)
    return np.sqrt((x.astype(np.float64) ** 2).sum(axis=0)).astype(np.float32)


def average_by_durations_ref(data: np.ndarray, durations) -> np.ndarray:
    out, pos = [], 0
    for d in np.asarray(durations).tolist():
        d = int(d)
        if d > 0:
            out.append(np.float32(np.mean(np.asarray(data[pos : pos + d], dtype=np.float32), dtype=np.float32)))
        else:
            out.append(np.float32(1e-7))
        pos += d
    return np.asarray(out, dtype=np.float32)


def scaler_stats_ref(chunks: list[np.ndarray]) -> dict:
    data = np.concatenate([np.asarray(c, dtype=np.float32) for c in chunks])
    ok = data[~np.isnan(data)]
    mean = np.float32(ok.astype(np.float64).mean())
    std = np.float32(ok.astype(np.float64).std(ddof=1))
    mn, mx = ok.min(), ok.max()
    return {
        "sample_size": len(chunks),
        "min": float(mn),
        "max": float(mx),
        "mean": float(mean),
        "std": float(std),
        "norm_min": float((mn - mean) / std),
        "norm_max": float((mx - mean) / std),
    }
