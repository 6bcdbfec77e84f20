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
