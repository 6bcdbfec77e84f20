"""GPU mirrors of the tensor utilities in the reference's ``everyvoice/utils/heavy.py`` that sit on
the hot path.  Same names and argument meaning; tensors must live on the GPU — there is no CPU
fallback (the reference's own functions remain the CPU implementation).

  expand(values, durations)              everyvoice/utils/heavy.py:12-21
  length_regulate(values, durations)     the batched, zero-padded use FastSpeech2 makes of expand
"""

from __future__ import annotations

import torch

from . import _lib


def _as_int64_durations(durations, device) -> torch.Tensor:
    d = torch.as_tensor(durations, device=device)
    if d.is_floating_point():
        d = d.trunc()  # the reference applies int(d): truncation toward zero
    return d.to(torch.int64).contiguous()


def length_regulate(values: torch.Tensor, durations, max_len: int | None = None, return_index: bool = False):
    """values [B, L, D] (fp32 / bf16 / fp16 / int32 / int16, copied bit-for-bit), durations [B, L] ->
    (out [B, T, D] zero padded, lengths [B] int64[, index [B, T] int32 with -1 padding]).
    T = ``max_len`` or the longest expanded item (needs one device->host read)."""
    if not values.is_cuda:
        raise RuntimeError("everyvoice_amd.heavy computes on the GPU only (no CPU fallback)")
    if values.dim() != 3:
        raise ValueError(f"values must be [B, L, D], got {tuple(values.shape)}")
    if values.element_size() not in (2, 4):
        raise ValueError(f"unsupported element size {values.element_size()} (2 or 4 bytes)")
    lib = _lib.load()
    values = values.contiguous()
    B, L, D = values.shape
    dur = _as_int64_durations(durations, values.device)
    if dur.shape != (B, L):
        raise ValueError(f"durations must be [{B}, {L}], got {tuple(dur.shape)}")
    if max_len is None:
        max_len = int(dur.clamp(min=0).sum(dim=1).max().item()) if B * L > 0 else 0
    out = torch.empty(B, max_len, D, device=values.device, dtype=values.dtype)
    lens = torch.empty(B, device=values.device, dtype=torch.int64)
    index = torch.empty(B, max_len, device=values.device, dtype=torch.int32) if return_index else None
    with torch.cuda.device(values.device):
        _lib.check(
            lib.evmi_length_regulate(values.data_ptr(), dur.data_ptr(), out.data_ptr(), lens.data_ptr(), _lib.ptr(index),
                                     B, L, D, max_len, values.element_size(), _lib.current_stream_ptr(values.device)),
            "evmi_length_regulate",
        )
    return (out, lens, index) if return_index else (out, lens)


def expand(values: torch.Tensor, durations) -> torch.Tensor:
    """Row i of ``values`` repeated max(0, int(d_i)) times (1-D values are treated as [L, 1])."""
    if not isinstance(values, torch.Tensor) or not values.is_cuda:
        raise RuntimeError("everyvoice_amd.heavy.expand takes CUDA tensors (no CPU fallback); "
                           "the reference's everyvoice.utils.heavy.expand covers lists / numpy / CPU tensors")
    squeeze = values.dim() == 1
    v = values.reshape(values.shape[0], -1).unsqueeze(0)
    d = _as_int64_durations(durations, values.device)
    n = min(v.shape[1], d.numel())  # zip() semantics of the reference
    out, _ = length_regulate(v[:, :n], d[:n].unsqueeze(0))
    if out.shape[1] == 0:
        raise RuntimeError("stack expects a non-empty TensorList")  # what torch.stack([]) raises in the reference
    out = out[0]
    return out[:, 0] if squeeze else out.reshape((out.shape[0],) + tuple(values.shape[1:]))


def length_regulate_backward(grad_out: torch.Tensor, durations, L: int | None = None) -> torch.Tensor:
    """grad wrt values of :func:`length_regulate` (fp32): sums each token's frames."""
    if not grad_out.is_cuda:
        raise RuntimeError("everyvoice_amd.heavy computes on the GPU only (no CPU fallback)")
    lib = _lib.load()
    go = grad_out.to(torch.float32).contiguous()
    B, T, D = go.shape
    dur = _as_int64_durations(durations, go.device)
    L = dur.shape[1]
    gv = torch.zeros(B, L, D, device=go.device, dtype=torch.float32)
    with torch.cuda.device(go.device):
        _lib.check(
            lib.evmi_length_regulate_bwd_f32(go.data_ptr(), dur.data_ptr(), gv.data_ptr(), B, L, D, T,
                                             _lib.current_stream_ptr(go.device)),
            "evmi_length_regulate_bwd_f32",
        )
    return gv


# ---- batch assembly and small tensor utilities of everyvoice/utils/heavy.py --------------------------------------------
def _flatten(tree: dict, prefix: str = "") -> dict:
    """Nested dict -> flat dict whose keys join the path with ``_`` (the reference's ``_flatten``,
    ``everyvoice/utils/__init__.py:121-133``: ``{"a": {"b": 2, "c": {"d": "e"}}, "g": 5} -> {"a_b": 2, "a_c_d": "e", "g": 5}``)."""
    flat = {}
    for key, value in tree.items():
        name = f"{prefix}_{key}" if prefix else key
        if isinstance(value, dict):
            flat.update(_flatten(value, name))
        else:
            flat[name] = value
    return flat


def collate_fn(data: list[dict]) -> dict:
    """Batch a list of (nested) sample dicts the way ``everyvoice/utils/heavy.py:24-36`` does: one entry per flattened key;
    numpy arrays and tensors are zero padded along dim 0 to the longest sample (``pad_sequence(batch_first=True)``; tensors
    already on the GPU are padded there), python ints become an ``IntTensor``, anything else stays a list."""
    import numpy as np
    from torch.nn.utils.rnn import pad_sequence

    samples = [_flatten(sample) for sample in data]
    batch = {}
    for key in samples[0]:
        column = [sample[key] for sample in samples]
        if isinstance(column[0], np.ndarray):
            column = [torch.tensor(item) for item in column]
        if torch.is_tensor(column[0]):
            batch[key] = pad_sequence(column, batch_first=True, padding_value=0)
        elif isinstance(column[0], int):
            batch[key] = torch.IntTensor(column)
        else:
            batch[key] = column
    return batch


def get_segments(t: torch.Tensor, segment_size: int, start=None) -> tuple[torch.Tensor, int]:
    """A window of ``segment_size`` along dim 1 and where it starts (``everyvoice/utils/heavy.py:122-148``).  Inputs at least
    as long as the window are cropped at ``start`` (which must not exceed ``len - segment_size - 1``; drawn with python's
    ``random.randint(0, len - segment_size - 1)`` when not given); shorter inputs are zero padded on the right, start 0."""
    import random

    length = t.size(1)
    if length < segment_size:
        return torch.nn.functional.pad(t, (0, segment_size - length)), 0
    last_start = length - segment_size - 1
    if start is None:
        start = random.randint(0, last_start)
    else:
        assert start <= last_start, f"segment start {start} is past the last admissible start {last_start}"
    return t[:, start : start + segment_size], start


def vocoder_training_batch(specs: list[torch.Tensor], audios: list[torch.Tensor], segment_frames: int, hop: int, starts=None):
    """The HiFiGAN training batch from device-resident utterances (what hfgl's SpecDataset + collate_fn deliver,
    everyvoice/tests/test_dataloader.py:55-65): spec [B, n_mels, segment_frames] and audio [B, segment_frames * hop] cropped at
    the same position (``start`` frames / ``start * hop`` samples)."""
    segs, wavs, used = [], [], []
    for i, (spec, audio) in enumerate(zip(specs, audios)):
        seg, st = get_segments(spec, segment_frames, None if starts is None else starts[i])
        wav, _ = get_segments(audio.reshape(1, -1), segment_frames * hop, st * hop if spec.size(1) >= segment_frames else None)
        segs.append(seg)
        wavs.append(wav.squeeze(0))
        used.append(st)
    return torch.stack(segs), torch.stack(wavs), used


def dynamic_range_compression_torch(x: torch.Tensor, C=1, clip_val=1e-5) -> torch.Tensor:
    """log(clamp(x, min=clip_val) * C) on the GPU (everyvoice/utils/heavy.py:39-40)."""
    if not x.is_cuda:
        raise RuntimeError("everyvoice_amd.heavy computes on the GPU only (no CPU fallback)")
    x = x.to(torch.float32).contiguous()
    y = torch.empty_like(x)
    # log(max(x, clip) * C) = log(max(x * C, clip * C)): one fused elementwise kernel (op 9: log(max(a, p0)))
    src = x if C == 1 else x * float(C)
    _lib.check(_lib.load().evmi_elementwise_f32(9, src.data_ptr(), 0, 0, y.data_ptr(), x.numel(), float(clip_val) * float(C), 0.0,
                                                _lib.current_stream_ptr(x.device)), "evmi_elementwise_f32")
    return y


def dynamic_range_decompression_torch(x: torch.Tensor, C=1) -> torch.Tensor:
    """exp(x) / C (everyvoice/utils/heavy.py:43-44); plumbing, left to torch."""
    return torch.exp(x) / C


class BetaBinomialInterpolator:
    """everyvoice/preprocessor/attention_prior.py:34-67: ``interp(w, h)`` -> float64 prior [w, h] (mel frames x text tokens),
    the beta-binomial table on a grid rounded to (100, 20) zoomed with order-1 interpolation.  Computed by
    ``evmi_attention_prior_f64`` on the device, cached per shape like the reference's lru_cache."""

    def __init__(self, round_mel_len_to=100, round_text_len_to=20, device="cuda:0"):
        self.round_mel_len_to, self.round_text_len_to = round_mel_len_to, round_text_len_to
        self.device = torch.device(device)
        self._cache: dict = {}

    @staticmethod
    def round(val, to):
        return max(1, int(round((val + 1) / to))) * to  # python round = numpy round: half to even

    def __call__(self, w: int, h: int) -> torch.Tensor:
        key = (int(w), int(h))
        if key not in self._cache:
            bw, bh = self.round(w, self.round_mel_len_to), self.round(h, self.round_text_len_to)
            out = torch.empty(key, device=self.device, dtype=torch.float64)
            with torch.cuda.device(self.device):
                _lib.check(_lib.load().evmi_attention_prior_f64(out.data_ptr(), key[0], key[1], bw, bh, _lib.current_stream_ptr(self.device)),
                           "evmi_attention_prior_f64")
            self._cache[key] = out
        return self._cache[key]


def maximum_path(value: torch.Tensor, mel_lens: torch.Tensor, text_lens: torch.Tensor):
    """Monotonic alignment search on the device (the reference's ``monotonic_align.maximum_path``, third-party
    ilt-monotonic-align): value [B, T, L] float32 log-likelihoods -> (path [B, T, L] int32, durations [B, L] int64)."""
    if not value.is_cuda:
        raise RuntimeError("everyvoice_amd.heavy computes on the GPU only (no CPU fallback)")
    v = value.to(torch.float32).contiguous()
    B, T, L = v.shape
    dev = v.device
    ml, tl = mel_lens.to(dev, torch.int32).contiguous(), text_lens.to(dev, torch.int32).contiguous()
    path = torch.empty(B, T, L, device=dev, dtype=torch.int32)
    dur = torch.empty(B, L, device=dev, dtype=torch.int32)
    scratch = torch.empty(B * T * L, device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().evmi_monotonic_align_f32(v.data_ptr(), ml.data_ptr(), tl.data_ptr(), path.data_ptr(), dur.data_ptr(),
                                                        scratch.data_ptr(), B, T, L, _lib.current_stream_ptr(dev)), "evmi_monotonic_align_f32")
    return path, dur.to(torch.int64)


# ---- alignment learning: attention over (mel, text), forward-sum loss, binarisation loss --------------
def alignment_attention(q_enc: torch.Tensor, k_enc: torch.Tensor, text_lens: torch.Tensor, prior: torch.Tensor | None = None,
                        temperature: float = 1.0):
    """q_enc [B, A, T], k_enc [B, A, L] (the projected mel / text of the aligner), prior [B, T, L] (float64, A9) ->
    (soft attention [B, T, L], log-probabilities [B, T, L])."""
    if not q_enc.is_cuda:
        raise RuntimeError("everyvoice_amd.heavy computes on the GPU only (no CPU fallback)")
    B, A, T = q_enc.shape
    L = k_enc.shape[2]
    dev = q_enc.device
    q = q_enc.to(torch.float32).permute(1, 0, 2).contiguous()
    k = k_enc.to(torch.float32).permute(1, 0, 2).contiguous()
    tl = text_lens.to(dev, torch.int32).contiguous()
    pr = None if prior is None else prior.to(dev, torch.float64).contiguous()
    soft = torch.empty(B, T, L, device=dev, dtype=torch.float32)
    logprob = torch.empty_like(soft)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().evmi_align_attention_f32(q.data_ptr(), k.data_ptr(), _lib.ptr(pr), tl.data_ptr(), soft.data_ptr(),
                                                        logprob.data_ptr(), A, B, T, L, float(temperature), _lib.current_stream_ptr(dev)),
                   "evmi_align_attention_f32")
    return soft, logprob


def forward_sum_loss(attn_logprob: torch.Tensor, text_lens: torch.Tensor, mel_lens: torch.Tensor, blank_logprob: float = -1.0) -> torch.Tensor:
    """Mean over the batch of the CTC forward-sum loss of every item (attn_logprob [B, T, L])."""
    B, T, L = attn_logprob.shape
    dev = attn_logprob.device
    lp = attn_logprob.to(torch.float32).contiguous()
    tl, ml = text_lens.to(dev, torch.int32).contiguous(), mel_lens.to(dev, torch.int32).contiguous()
    per_item = torch.empty(B, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().evmi_forward_sum_loss_f32(lp.data_ptr(), tl.data_ptr(), ml.data_ptr(), per_item.data_ptr(), B, T, L,
                                                         float(blank_logprob), _lib.current_stream_ptr(dev)), "evmi_forward_sum_loss_f32")
    return per_item.mean()


def binarization_loss(hard: torch.Tensor, soft: torch.Tensor) -> torch.Tensor:
    """-sum log(clamp(soft, 1e-12)) over the hard alignment's cells / number of cells."""
    dev = soft.device
    h, s = hard.to(dev, torch.int32).contiguous(), soft.to(torch.float32).contiguous()
    n = s.numel()
    nb = max(1, min(256, (n + 2047) // 2048))
    part = torch.empty(nb, 2, device=dev, dtype=torch.float64)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().evmi_binarization_partials_f64(h.data_ptr(), s.data_ptr(), part.data_ptr(), nb, n, _lib.current_stream_ptr(dev)),
                   "evmi_binarization_partials_f64")
    tot = part.sum(0)
    return (-tot[0] / tot[1]).to(torch.float32)
