"""Length regulator on the GPU (through the C ABI) vs the oracle: bit-exact."""

import numpy as np
import pytest
import torch

from oracle import heavy_ref

pytestmark = pytest.mark.gpu


def test_expand_golden_bit_exact(golden_dir, cuda_device):
    from everyvoice_amd.heavy import expand

    g = np.load(golden_dir / "expand.npz")
    for i in range(5):
        v = torch.from_numpy(g[f"lj{i}_val"]).to(cuda_device)
        d = torch.from_numpy(g[f"lj{i}_dur"]).to(cuda_device)
        out = expand(v, d)
        assert out.dtype == torch.float32
        assert np.array_equal(out.cpu().numpy().view(np.uint32), g[f"lj{i}_out"].view(np.uint32))
    out = expand(torch.from_numpy(g["edge_val"]).to(cuda_device), torch.from_numpy(g["edge_dur"]).to(cuda_device))
    assert np.array_equal(out.cpu().numpy(), g["edge_out"])  # zero and negative durations
    out = expand(torch.from_numpy(g["frac_val"]).to(cuda_device), torch.from_numpy(g["frac_dur"]).to(cuda_device))
    assert np.array_equal(out.cpu().numpy(), g["frac_out"])  # float durations truncate like int()
    out = expand(torch.arange(5, dtype=torch.float32, device=cuda_device), [1, 2, 0, 1, 3])
    assert np.array_equal(out.cpu().numpy(), g["np1d_out"])
    with pytest.raises(RuntimeError):  # the reference's torch.stack([]) error on an all-zero duration vector
        expand(torch.ones(3, 4, device=cuda_device), [0, 0, -1])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16, torch.int32])
@pytest.mark.parametrize("D", [256, 80, 7])
def test_batched_vs_oracle(cuda_device, dtype, D):
    from everyvoice_amd.heavy import length_regulate

    rng = np.random.default_rng(1234)
    B, L = 9, 61
    durs = rng.integers(-1, 9, size=(B, L)).astype(np.int64)
    durs[3] = 0  # one empty item
    if dtype == torch.int32:
        vals = torch.from_numpy(rng.integers(-2**31, 2**31 - 1, size=(B, L, D)).astype(np.int32))
    else:
        vals = torch.from_numpy(rng.standard_normal((B, L, D)).astype(np.float32)).to(dtype)
    want, want_lens = heavy_ref.length_regulate_batch_ref(vals.view(torch.int16 if vals.element_size() == 2 else torch.int32).numpy(), durs)
    out, lens, idx = length_regulate(vals.to(cuda_device), torch.from_numpy(durs).to(cuda_device), return_index=True)
    got = out.cpu().view(torch.int16 if vals.element_size() == 2 else torch.int32).numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    assert np.array_equal(lens.cpu().numpy(), want_lens)
    idx = idx.cpu().numpy()
    for b in range(B):
        ref_idx = heavy_ref.expand_index_ref(durs[b])
        assert np.array_equal(idx[b, : len(ref_idx)], ref_idx) and (idx[b, len(ref_idx) :] == -1).all()
    # explicit max_len: truncation and extra padding
    for max_len in (5, want.shape[1] + 13):
        o2, l2 = length_regulate(vals.to(cuda_device), torch.from_numpy(durs).to(cuda_device), max_len=max_len)
        w2, wl2 = heavy_ref.length_regulate_batch_ref(vals.view(torch.int16 if vals.element_size() == 2 else torch.int32).numpy(), durs, max_len)
        assert np.array_equal(o2.cpu().view(torch.int16 if vals.element_size() == 2 else torch.int32).numpy(), w2)
        assert np.array_equal(l2.cpu().numpy(), wl2)


def test_empty_and_degenerate(cuda_device):
    from everyvoice_amd.heavy import length_regulate

    out, lens = length_regulate(torch.zeros(2, 0, 8, device=cuda_device), torch.zeros(2, 0, dtype=torch.int64, device=cuda_device), max_len=4)
    assert out.shape == (2, 4, 8) and not out.any() and not lens.any()
    out, lens = length_regulate(torch.ones(2, 3, 8, device=cuda_device), torch.zeros(2, 3, dtype=torch.int64, device=cuda_device))
    assert out.shape == (2, 0, 8) and not lens.any()


def test_full_size_properties(cuda_device):
    """BASELINE config 3 shape (B=32, L=187, D=256, T<=947): size-independent checks."""
    from everyvoice_amd.heavy import length_regulate

    g = torch.Generator().manual_seed(1234)
    B, L, D = 32, 187, 256
    durs = torch.randint(0, 11, (B, L), generator=g)
    vals = torch.randn(B, L, D, generator=g)
    out, lens, idx = length_regulate(vals.to(cuda_device), durs.to(cuda_device), return_index=True)
    out, lens, idx = out.cpu(), lens.cpu(), idx.cpu().long()
    assert torch.equal(lens, durs.sum(1))
    for b in range(B):
        n = int(lens[b])
        assert (idx[b, :n].diff() >= 0).all() and (idx[b, n:] == -1).all()  # sortedness
        assert torch.equal(torch.bincount(idx[b, :n], minlength=L), durs[b])  # each token d_i times
        assert torch.equal(out[b, :n], vals[b][idx[b, :n]]) and not out[b, n:].any()  # pure gather, bitwise
    # checksum of checksums on the raw bits
    bits = vals.view(torch.int32).long().sum(-1)
    assert int((bits * durs).sum()) == int(out.view(torch.int32).long().sum())


def test_backward_is_segment_sum(cuda_device):
    from everyvoice_amd.heavy import length_regulate, length_regulate_backward

    g = torch.Generator().manual_seed(7)
    B, L, D = 4, 33, 80
    durs = torch.randint(0, 6, (B, L), generator=g)
    T = int(durs.sum(1).max())
    go = torch.randn(B, T, D, generator=g)
    gv = length_regulate_backward(go.to(cuda_device), durs.to(cuda_device)).cpu()
    vals = torch.randn(B, L, D, generator=g, dtype=torch.float64).requires_grad_()
    # autograd of the oracle formulation
    outs = []
    for b in range(B):
        e = vals[b][torch.from_numpy(heavy_ref.expand_index_ref(durs[b].numpy()))]
        outs.append(torch.nn.functional.pad(e, (0, 0, 0, T - e.shape[0])))
    torch.stack(outs).backward(go.double())
    torch.testing.assert_close(gv.double(), vals.grad, rtol=1e-5, atol=1e-5)


def test_attention_prior_matches_reference_golden(cuda_device, golden_dir):
    """evmi_attention_prior_f64 (A9) against vectors from the reference's BetaBinomialInterpolator (scipy betabinom + zoom)."""
    import numpy as np

    from everyvoice_amd.heavy import BetaBinomialInterpolator

    g = np.load(golden_dir / "attn_prior.npz")
    interp = BetaBinomialInterpolator(device=cuda_device)
    for i, (T, L) in enumerate(g["shapes"]):
        got = interp(int(T), int(L))
        assert got.dtype == torch.float64 and tuple(got.shape) == (T, L)
        np.testing.assert_allclose(got.cpu().numpy(), g[f"p{i}"], rtol=1e-9, atol=1e-13)
    assert interp(32, 7) is interp(32, 7)  # cached per shape


def test_dynamic_range_compression_matches_reference_golden(cuda_device, golden_dir):
    import numpy as np

    from everyvoice_amd.heavy import dynamic_range_compression_torch, dynamic_range_decompression_torch

    g = np.load(golden_dir / "drc.npz")
    x = torch.from_numpy(g["x"]).to(cuda_device)
    got = dynamic_range_compression_torch(x)
    np.testing.assert_allclose(got.cpu().numpy(), g["drc"], rtol=4e-7, atol=1e-6)  # the device logf is within 2 ulp of torch's
    np.testing.assert_allclose(dynamic_range_decompression_torch(got).cpu().numpy(), g["drd"], rtol=4e-6)
    np.testing.assert_allclose(dynamic_range_compression_torch(x, C=2.0).cpu().numpy(), np.log(np.maximum(g["x"], 1e-5) * 2.0), rtol=4e-7, atol=2e-6)


def test_monotonic_alignment_search_matches_oracle(cuda_device):
    """evmi_monotonic_align_f32 vs the restated Glow-TTS dynamic programme: paths and durations are integers -> exact,
    including ties (quantised scores), ragged lengths, t_x == t_y (pure diagonal) and a single token."""
    import numpy as np

    from everyvoice_amd.heavy import maximum_path
    from oracle.mas_ref import maximum_path_batch_ref

    g = torch.Generator().manual_seed(6)
    for B, T, L, quant in ((4, 60, 17, False), (3, 200, 45, True), (2, 9, 9, False), (2, 30, 1, False), (5, 947, 187, False)):
        v = torch.randn(B, T, L, generator=g) * 3
        if quant:
            v = (v * 2).round() / 2  # many exact ties
        mel_lens = torch.randint(max(L, T // 2), T + 1, (B,), generator=g)
        text_lens = torch.minimum(torch.randint(max(1, L // 2), L + 1, (B,), generator=g), mel_lens)
        mel_lens[0], text_lens[0] = T, L
        if T == L:
            mel_lens[:], text_lens[:] = T, L
        want_path, want_dur = maximum_path_batch_ref(v.numpy(), mel_lens.numpy(), text_lens.numpy())
        path, dur = maximum_path(v.to(cuda_device), mel_lens, text_lens)
        assert np.array_equal(path.cpu().numpy(), want_path), (B, T, L)
        assert np.array_equal(dur.cpu().numpy(), want_dur)
        assert torch.equal(dur.sum(1).cpu(), mel_lens)  # every frame belongs to exactly one token


def test_alignment_learning_forward_matches_oracle(cuda_device):
    """F5 forward: attention (with the A9 prior), CTC forward-sum loss, MAS hard alignment, binarisation loss vs torch CPU."""
    from everyvoice_amd.heavy import (BetaBinomialInterpolator, alignment_attention, binarization_loss, forward_sum_loss, maximum_path)
    from oracle.alignment_ref import alignment_attention_ref, binarization_loss_ref, forward_sum_loss_ref
    from oracle.attention_prior_ref import attention_prior_ref

    g = torch.Generator().manual_seed(21)
    B, A, T, L = 3, 80, 57, 19
    text_lens, mel_lens = torch.tensor([19, 11, 15]), torch.tensor([57, 40, 52])
    q, k = torch.randn(B, A, T, generator=g) * 0.3, torch.randn(B, A, L, generator=g) * 0.3
    prior = torch.zeros(B, T, L, dtype=torch.float64)
    interp = BetaBinomialInterpolator(device=cuda_device)
    for b in range(B):
        prior[b, : mel_lens[b], : text_lens[b]] = torch.from_numpy(attention_prior_ref(int(mel_lens[b]), int(text_lens[b])))
        torch.testing.assert_close(interp(int(mel_lens[b]), int(text_lens[b])).cpu(), prior[b, : mel_lens[b], : text_lens[b]], rtol=1e-9, atol=1e-13)
    for pr in (prior, None):
        want_soft, want_lp = alignment_attention_ref(q, k, text_lens, pr)
        soft, lp = alignment_attention(q.to(cuda_device), k.to(cuda_device), text_lens, pr)
        torch.testing.assert_close(lp.cpu(), want_lp, rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(soft.cpu(), want_soft, rtol=2e-5, atol=1e-7)
        want_ctc = forward_sum_loss_ref(want_lp, text_lens, mel_lens)
        got_ctc = forward_sum_loss(lp, text_lens, mel_lens)
        assert float(got_ctc) == pytest.approx(float(want_ctc), rel=2e-5)
    # hard alignment from the soft one (log domain, as the reference's binarisation does), then the binarisation loss
    path, dur = maximum_path(torch.log(soft.clamp_min(1e-12)), mel_lens, text_lens)
    hard = path.cpu()
    assert float(binarization_loss(path, soft)) == pytest.approx(float(binarization_loss_ref(hard, want_soft)), rel=2e-5)
    assert torch.equal(dur.sum(1).cpu(), mel_lens)
