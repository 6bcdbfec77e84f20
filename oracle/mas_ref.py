"""Monotonic alignment search, restated.  TEST INFRASTRUCTURE ONLY.

The reference takes hard alignments from the third-party package ``ilt-monotonic-align`` 1.2.1 (``pyproject.toml:79``;
absent here), a packaging of the Glow-TTS ``monotonic_align.maximum_path`` Cython kernel (Kim et al. 2020, Algorithm 1).
PARITY UNPINNED against that package; this file restates the published dynamic programme:

  Q[y, x] = value[y, x] + max(Q[y-1, x], Q[y-1, x-1])       over mel frames y < t_y and tokens x < t_x, monotonic, every
  token visited (x <= y and the path must end at (t_y - 1, t_x - 1)); the path is read back from the end, moving to x - 1
  when x == y or Q[y-1, x] < Q[y-1, x-1].
"""

from __future__ import annotations

import numpy as np

MAX_NEG = -1e9


def maximum_path_ref(value: np.ndarray, t_y: int, t_x: int) -> np.ndarray:
    """value [T, L] float32 (log-likelihood of frame y under token x) -> path [T, L] int32 (one token per frame)."""
    v = np.array(value, dtype=np.float32, copy=True)
    path = np.zeros(v.shape, dtype=np.int32)
    for y in range(t_y):
        for x in range(max(0, t_x + y - t_y), min(t_x, y + 1)):
            v_cur = np.float32(MAX_NEG) if x == y else v[y - 1, x]
            if x == 0:
                v_prev = np.float32(0.0) if y == 0 else np.float32(MAX_NEG)
            else:
                v_prev = v[y - 1, x - 1]
            v[y, x] = v[y, x] + max(v_prev, v_cur)
    index = t_x - 1
    for y in range(t_y - 1, -1, -1):
        path[y, index] = 1
        if index != 0 and (index == y or v[y - 1, index] < v[y - 1, index - 1]):
            index -= 1
    return path


def maximum_path_batch_ref(values: np.ndarray, mel_lens, text_lens):
    """values [B, T, L] -> (paths [B, T, L] int32, durations [B, L] int64 = frames per token)."""
    paths = np.stack([maximum_path_ref(values[b], int(mel_lens[b]), int(text_lens[b])) for b in range(values.shape[0])])
    return paths, paths.sum(axis=1).astype(np.int64)
