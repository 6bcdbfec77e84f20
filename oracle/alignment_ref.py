"""Alignment learning of the feature-prediction model, restated -- TEST INFRASTRUCTURE ONLY.

The reference's implementation lives in the absent submodule FastSpeech2_lightning; its configuration names the method
(``learn_alignment``: Badlani et al. 2021, arXiv 2108.10447) and the loss weights (``attn_ctc_loss_weight``,
``attn_bin_loss_weight``, ``everyvoice/.schema/everyvoice-text-to-spec-0.5.json`` FastSpeech2TrainingConfig).  PARITY UNPINNED;
this follows the public implementation of that paper (NVIDIA FastPitch ``ConvAttention`` / ``AttentionCTCLoss`` /
``AttentionBinarizationLoss``) with the beta-binomial prior of ``everyvoice/preprocessor/attention_prior.py`` (A9, pinned).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F


def alignment_attention_ref(q_enc, k_enc, text_lens, prior=None, temperature: float = 1.0):
    """q_enc [B, A, T] (projected mel), k_enc [B, A, L] (projected text), prior [B, T, L] or None ->
    (attn soft [B, T, L] with padded keys at probability 0, attn_logprob [B, T, L] before masking)."""
    attn = -temperature * ((q_enc[:, :, :, None] - k_enc[:, :, None, :]) ** 2).sum(1)  # [B, T, L]
    if prior is not None:
        attn = F.log_softmax(attn, dim=2) + torch.log(prior.to(attn.dtype) + 1e-8)
    logprob = attn.clone()
    L = k_enc.shape[2]
    mask = torch.arange(L)[None, None, :] >= text_lens[:, None, None]
    soft = F.softmax(attn.masked_fill(mask, float("-inf")), dim=2)
    return soft, logprob


def forward_sum_loss_ref(attn_logprob, text_lens, mel_lens, blank_logprob: float = -1.0):
    """CTC forward-sum loss over the alignment log-probabilities [B, T, L] (blank prepended at index 0)."""
    padded = F.pad(attn_logprob, (1, 0), value=blank_logprob)  # [B, T, L + 1]
    total = attn_logprob.new_zeros(())
    for b in range(attn_logprob.shape[0]):
        Lb, Tb = int(text_lens[b]), int(mel_lens[b])
        target = torch.arange(1, Lb + 1).unsqueeze(0)
        cur = F.log_softmax(padded[b, :Tb, : Lb + 1], dim=1).unsqueeze(1)  # [T, 1, L + 1]
        total = total + F.ctc_loss(cur, target, input_lengths=torch.tensor([Tb]), target_lengths=torch.tensor([Lb]), blank=0,
                                   zero_infinity=True)
    return total / attn_logprob.shape[0]


def binarization_loss_ref(hard, soft):
    """-mean log soft attention on the cells of the hard (monotonic) alignment."""
    return -torch.log(torch.clamp(soft[hard == 1], min=1e-12)).sum() / hard.sum()


class AlignerRef(torch.nn.Module):
    """FastPitch ``ConvAttention``: key / query projections in front of ``alignment_attention_ref`` (parameter names as there:
    ``key_proj.{0,2}.conv``, ``query_proj.{0,2,4}.conv``)."""

    def __init__(self, d_text: int, n_mels: int, n_att: int = 80, temperature: float = 0.0005):
        super().__init__()
        conv = lambda cin, cout, k: _ConvNormRef(cin, cout, k)
        self.key_proj = torch.nn.Sequential(conv(d_text, 2 * d_text, 3), torch.nn.ReLU(), conv(2 * d_text, n_att, 1))
        self.query_proj = torch.nn.Sequential(conv(n_mels, 2 * n_mels, 3), torch.nn.ReLU(), conv(2 * n_mels, n_mels, 1), torch.nn.ReLU(),
                                              conv(n_mels, n_att, 1))
        self.temperature = temperature

    def forward(self, mel, text_emb, text_lens, prior=None):
        """mel [B, n_mels, T], text_emb [B, D, L] -> (soft, logprob) [B, T, L]"""
        return alignment_attention_ref(self.query_proj(mel), self.key_proj(text_emb), text_lens, prior, self.temperature)


class _ConvNormRef(torch.nn.Module):
    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv = torch.nn.Conv1d(cin, cout, k, padding=(k - 1) // 2)

    def forward(self, x):
        return self.conv(x)
