"""FastSpeech2 forward path (csrc/fs2_ops.hip + the fp32 matrix-core GEMM) against the torch-CPU oracle.

Tolerances: fp32 everywhere; the only differences are summation order and expf / sinf implementations, so activations agree
to ~1e-5 relative; 2e-4 (relative to the tensor's largest entry) leaves room for 8 conformer layers of accumulation.
Durations are integers: with durations given they must be bit-exact; predicted ones are compared where exp(log_d) - 1 is
not within 1e-3 of a rounding boundary."""

import pytest
import torch

from oracle.fs2_ref import FastSpeech2ConfigRef, FastSpeech2Ref, randomize_norm_stats_

pytestmark = pytest.mark.gpu


def _close(got, want, rel=2e-4):
    scale = float(want.abs().max()) + 1e-12
    err = float((got - want).abs().max())
    assert err <= rel * scale, f"err {err:.3e} vs scale {scale:.3e}"


def _product_config(ref_cfg):
    from everyvoice_amd.fs2 import ConformerConfig, FastSpeech2ModelConfig, VariancePredictorConfig, VariancePredictors

    conf = lambda c: ConformerConfig(layers=c.layers, heads=c.heads, input_dim=c.input_dim, feedforward_dim=c.feedforward_dim,
                                     conv_kernel_size=c.conv_kernel_size, dropout=c.dropout)
    vpc = lambda v: VariancePredictorConfig(n_layers=v.n_layers, kernel_size=v.kernel_size, input_dim=v.input_dim, n_bins=v.n_bins,
                                            depthwise=v.depthwise, level=v.level)
    return FastSpeech2ModelConfig(encoder=conf(ref_cfg.encoder), decoder=conf(ref_cfg.decoder),
                                  variance_predictors=VariancePredictors(energy=vpc(ref_cfg.energy), duration=vpc(ref_cfg.duration), pitch=vpc(ref_cfg.pitch)),
                                  n_symbols=ref_cfg.n_symbols, n_mels=ref_cfg.n_mels, use_postnet=ref_cfg.use_postnet,
                                  postnet_channels=ref_cfg.postnet_channels, postnet_kernel=ref_cfg.postnet_kernel,
                                  postnet_layers=ref_cfg.postnet_layers,
                                  target_text_representation_level=ref_cfg.target_text_representation_level)


def _models(ref_cfg, cuda_device, seed):
    from everyvoice_amd.fs2 import FastSpeech2

    torch.manual_seed(seed)
    ref = FastSpeech2Ref(ref_cfg).eval()
    g = torch.Generator().manual_seed(seed + 1)
    randomize_norm_stats_(ref, g)
    with torch.no_grad():  # livelier than the default init: biases and embeddings that matter
        for n, p in ref.named_parameters():
            if n.endswith("bias"):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    model = FastSpeech2(_product_config(ref_cfg), device=cuda_device).load_state_dict(ref.state_dict())
    return ref, model


def _batch(n_symbols, B, L, seed, lens=None):
    g = torch.Generator().manual_seed(seed)
    lens = torch.tensor(lens) if lens is not None else torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lens[0] = L
    ids = torch.randint(1, n_symbols, (B, L), generator=g)
    ids = ids.masked_fill(torch.arange(L)[None] >= lens[:, None], 0)
    return ids, lens, g


def test_fs2_phonological_features_input(cuda_device):
    """target_text_representation_level = "phonological_features" (everyvoice-text-to-spec-0.5.json:265-272): 43-dim multi-hot
    vectors (text/features.py:7) through a bias-free Linear instead of symbol ids through the embedding table."""
    ref_cfg = FastSpeech2ConfigRef.small()
    ref_cfg.target_text_representation_level = "phonological_features"
    ref, model = _models(ref_cfg, cuda_device, seed=77)
    assert tuple(ref.state_dict()["text_input_layer.weight"].shape) == (ref_cfg.encoder.input_dim, 43)
    B, L = 3, 14
    _, lens, g = _batch(20, B, L, seed=5)
    feats = (torch.rand(B, L, 43, generator=g) < 0.3).float()
    durs = torch.randint(0, 6, (B, L), generator=g)
    durs[:, 0] += 1
    want = ref(feats, lens, durations=durs)
    got = model(feats, lens, durations=durs)
    assert torch.equal(got[2].cpu(), want[2]) and torch.equal(got[5].cpu(), want[5])
    for i in (0, 1, 3, 4):
        _close(got[i].cpu(), want[i])
    with pytest.raises(ValueError, match="phonological features"):
        model(torch.zeros(B, L, dtype=torch.long), lens, durations=durs)


@pytest.mark.parametrize("B,L", [(3, 12), (1, 5), (4, 33)])
def test_fs2_small_given_durations(cuda_device, B, L):
    ref, model = _models(FastSpeech2ConfigRef.small(), cuda_device, seed=B * 100 + L)
    ids, lens, g = _batch(20, B, L, seed=7)
    durs = torch.randint(0, 6, (B, L), generator=g)
    durs[:, 0] += 1
    want = ref(ids, lens, durations=durs)
    got = model(ids, lens, durations=durs)
    assert torch.equal(got[2].cpu(), want[2])          # durations: integers, bit-exact
    assert torch.equal(got[5].cpu(), want[5])          # mel lengths
    _close(got[3].cpu(), want[3])                      # pitch
    _close(got[4].cpu(), want[4])                      # energy
    _close(got[0].cpu(), want[0])                      # decoder mel
    _close(got[1].cpu(), want[1])                      # postnet mel
    fpad = torch.arange(got[1].shape[1])[None, :] >= want[5][:, None]  # padded frames are exactly zero
    assert float(got[1].cpu()[fpad].abs().sum()) == 0.0 and float(got[0].cpu()[fpad].abs().sum()) == 0.0


def test_fs2_small_predicted_durations_and_controls(cuda_device):
    ref, model = _models(FastSpeech2ConfigRef.small(), cuda_device, seed=5)
    with torch.no_grad():
        ref.duration_predictor.linear.bias.fill_(1.2)  # durations of a few frames instead of 0
    model.load_state_dict(ref.state_dict())
    ids, lens, _ = _batch(20, 3, 14, seed=9)
    kw = dict(duration_control=1.0, pitch_control=1.3, energy_control=0.8)
    want = ref(ids, lens, **kw)
    # the oracle's log-durations, to know which tokens sit on a rounding boundary
    pad = torch.arange(14)[None] >= lens[:, None]
    x = ref.text_input_layer(ids) + ref.position_embedding(14)[None]
    x, _ = ref.encoder(x.masked_fill(pad[..., None], 0.0), lens)
    raw = torch.exp(ref.duration_predictor(x, pad)) - 1.0
    safe = ((raw - torch.floor(raw) - 0.5).abs() > 1e-3) | pad
    got = model(ids, lens, **kw)
    assert safe.all(), "test input sits on a rounding boundary: pick another seed"
    assert torch.equal(got[2].cpu(), want[2])
    _close(got[3].cpu(), want[3])
    _close(got[4].cpu(), want[4])
    _close(got[1].cpu(), want[1])


def test_fs2_default_config(cuda_device):
    """The reference's default sizes (conformer 4 x 256 / 1024 / k9, 2 heads of 128; predictors 5 x k3; postnet 5 x 512)."""
    ref, model = _models(FastSpeech2ConfigRef(), cuda_device, seed=3)
    ids, lens, g = _batch(80, 2, 40, seed=11, lens=[40, 23])
    durs = torch.randint(2, 9, (2, 40), generator=g)
    want = ref(ids, lens, durations=durs)
    got = model(ids, lens, durations=durs)
    assert torch.equal(got[5].cpu(), want[5])
    _close(got[0].cpu(), want[0], rel=5e-4)
    _close(got[1].cpu(), want[1], rel=5e-4)


def test_fs2_default_config_bf16_operands(cuda_device):
    """precision="bf16": the dense layers take bf16 operands (fp32 accumulation, fp32 LayerNorm / attention / activations).
    Against the fp32 oracle with durations given (so the lengths are exact).  The pitch / energy embeddings are made a SMOOTH
    function of the bin (as trained ones are): a predicted value moved by the operand rounding (1 % of its range = a few of the
    256 bins) then moves the embedding a little; with the default independent random rows it would swap it for an unrelated
    unit-variance vector (measured: pitch off by 1e-2, everything behind the embedding by 0.7 -- a property of the random
    table, not of the arithmetic, which tests/test_gpu_train_ops.py pins per operator).  Tolerance: 5e-2 of each tensor's scale."""
    from everyvoice_amd.fs2 import FastSpeech2

    ref_cfg = FastSpeech2ConfigRef()
    torch.manual_seed(3)
    ref = FastSpeech2Ref(ref_cfg).eval()
    g = torch.Generator().manual_seed(4)
    randomize_norm_stats_(ref, g)
    with torch.no_grad():
        ramp = torch.linspace(-1.0, 1.0, 256)[:, None]
        ref.pitch_embedding.weight.copy_(ramp * torch.randn(1, 256, generator=g))
        ref.energy_embedding.weight.copy_(ramp * torch.randn(1, 256, generator=g))
    model = FastSpeech2(_product_config(ref_cfg), device=cuda_device, precision="bf16").load_state_dict(ref.state_dict())
    ids, lens, g = _batch(80, 2, 40, seed=11, lens=[40, 23])
    durs = torch.randint(2, 9, (2, 40), generator=g)
    want = ref(ids, lens, durations=durs)
    got = model(ids, lens, durations=durs)
    assert torch.equal(got[5].cpu(), want[5])
    for i in (3, 4, 0, 1):
        _close(got[i].cpu(), want[i], rel=5e-2)
    model.precision = "f32"  # the switch is per object and per call: the fp32 path is untouched
    got32 = model(ids, lens, durations=durs)
    _close(got32[1].cpu(), want[1], rel=5e-4)
    assert float((got32[1] - got[1]).abs().max()) > 0.0


def test_attention_kernel_alone(cuda_device):
    """evmi_attention_cbt_f32 vs torch scaled-dot-product attention with a key padding mask, ragged lengths, T not a multiple of 32."""
    from everyvoice_amd import _lib

    g = torch.Generator().manual_seed(2)
    for (B, T, D, H) in ((2, 50, 64, 2), (3, 200, 256, 2), (1, 33, 128, 2)):
        qkv = torch.randn(3 * D, B, T, generator=g)
        lens = torch.randint(1, T + 1, (B,), generator=g)
        lens[0] = T
        q, k, v = [t.view(H, D // H, B, T).permute(2, 0, 3, 1) for t in qkv.split(D)]  # [B, H, T, dh]
        mask = (torch.arange(T)[None] >= lens[:, None])[:, None, None, :]
        s = (q @ k.transpose(-1, -2)) / (D // H) ** 0.5
        want = (s.masked_fill(mask, float("-inf")).softmax(-1) @ v).permute(1, 3, 0, 2).reshape(D, B, T)
        qd, ld = qkv.to(cuda_device), lens.to(cuda_device, torch.int32)
        out = torch.empty(D, B, T, device=cuda_device)
        _lib.check(_lib.load().evmi_attention_cbt_f32(qd.data_ptr(), ld.data_ptr(), out.data_ptr(), B, T, D, H,
                                                      torch.cuda.current_stream().cuda_stream), "attention")
        torch.testing.assert_close(out.cpu(), want, rtol=2e-5, atol=2e-5)


def test_attention_kernel_bf16_operands(cuda_device):
    """evmi_attention_cbt_bf16 vs torch attention with the SAME roundings restated: q (pre-scaled), k, v and the un-normalised
    probabilities exp(s - max) rounded to bf16, everything else fp32 (the running maximum of the kernel's online softmax only
    rescales: the restatement uses the row maximum, so the probabilities may differ by their bf16 rounding -> 1e-2 of the
    output's scale); and within 2e-2 of the exact attention."""
    from everyvoice_amd import _lib

    def bf(t):
        return t.to(torch.bfloat16).to(torch.float32)

    g = torch.Generator().manual_seed(3)
    for (B, T, D, H) in ((2, 50, 64, 2), (3, 200, 256, 2), (1, 33, 128, 2), (2, 70, 128, 4)):
        qkv = torch.randn(3 * D, B, T, generator=g)
        lens = torch.randint(1, T + 1, (B,), generator=g)
        lens[0] = T
        q, k, v = [t.view(H, D // H, B, T).permute(2, 0, 3, 1) for t in qkv.split(D)]  # [B, H, T, dh]
        mask = (torch.arange(T)[None] >= lens[:, None])[:, None, None, :]
        scale = 1.0 / (D // H) ** 0.5
        exact = ((q @ k.transpose(-1, -2)) * scale).masked_fill(mask, float("-inf")).softmax(-1) @ v
        s = (bf(q * scale) @ bf(k).transpose(-1, -2)).masked_fill(mask, float("-inf"))
        p = torch.exp(s - s.amax(-1, keepdim=True))
        rounded = (bf(p) @ bf(v)) / p.sum(-1, keepdim=True)
        qd, ld = qkv.to(cuda_device), lens.to(cuda_device, torch.int32)
        out = torch.empty(D, B, T, device=cuda_device)
        _lib.check(_lib.load().evmi_attention_cbt_bf16(qd.data_ptr(), ld.data_ptr(), out.data_ptr(), B, T, D, H,
                                                       torch.cuda.current_stream().cuda_stream), "attention")
        got = out.cpu().view(H, D // H, B, T).permute(2, 0, 3, 1)
        sc = float(exact.abs().max())
        assert float((got - rounded).abs().max()) <= 1e-2 * sc, (B, T, D, H)
        assert float((got - exact).abs().max()) <= 2e-2 * sc, (B, T, D, H)


def test_fs2_against_committed_golden(cuda_device):
    """The committed fixture (inputs, parameters and the oracle's outputs) through the HIP path."""
    import numpy as np

    from pathlib import Path

    from everyvoice_amd.fs2 import FastSpeech2

    z = np.load(Path(__file__).parent / "golden" / "fs2_small.npz")
    sd = {k[len("param:"):]: torch.from_numpy(z[k].astype(np.float32) if z[k].dtype == np.float16 else z[k]) for k in z.files if k.startswith("param:")}
    model = FastSpeech2(_product_config(FastSpeech2ConfigRef.small()), device=cuda_device).load_state_dict(sd)
    ids, lens = torch.from_numpy(z["ids"]), torch.from_numpy(z["lens"])
    tf = model(ids, lens, durations=torch.from_numpy(z["given_durations"]))
    free = model(ids, lens, duration_control=1.0, pitch_control=1.2, energy_control=0.9)
    for tag, o in (("tf", tf), ("free", free)):
        assert torch.equal(o[2].cpu(), torch.from_numpy(z[f"{tag}_durations"]))
        assert torch.equal(o[5].cpu(), torch.from_numpy(z[f"{tag}_mel_lens"]))
        _close(o[1].cpu(), torch.from_numpy(z[f"{tag}_post"]))
        _close(o[3].cpu(), torch.from_numpy(z[f"{tag}_pitch"]))


def test_synthesize_from_text_writes_reference_named_files(cuda_device, tmp_path):
    """ids -> FastSpeech2 -> HiFiGAN -> wav + spec files (file counts and names, as tests/test_cli.py:106-169 checks)."""
    import wave

    from everyvoice_amd.config import HiFiGANConfig
    from everyvoice_amd.fs2 import FastSpeech2
    from everyvoice_amd.pipeline import synthesize_from_text
    from everyvoice_amd.vocoder import HiFiGANGenerator

    fs2 = FastSpeech2(device=cuda_device).init_random(3)
    # durations of a few frames per token instead of the zeros a random duration predictor gives
    fs2.duration_predictor.b_lin.fill_(1.0)
    voc = HiFiGANGenerator(HiFiGANConfig(), precision="bf16").to(cuda_device).eval()
    ids, lens, _ = _batch(80, 2, 9, seed=1, lens=[9, 6])
    recs = synthesize_from_text(ids, lens, fs2, voc, tmp_path, ["utt-a", "utt-b"])
    assert sorted(p.name for p in (tmp_path / "wav").iterdir()) == ["utt-a--default--default--pred.wav", "utt-b--default--default--pred.wav"]
    assert len(list((tmp_path / "synthesized_spec").iterdir())) == 2
    for r in recs:
        spec = torch.load(r["spec"])
        assert spec.shape == (80, r["frames"]) and r["frames"] == int(r["durations"].sum())
        with wave.open(str(r["wav"])) as w:
            assert w.getframerate() == 22050 and w.getnframes() == r["frames"] * 256 and w.getsampwidth() == 2


def test_multispeaker_multilingual_and_checkpoint_roundtrip(cuda_device):
    """Speaker / language embeddings (BASELINE config 5: multi-speaker FastSpeech2) and the checkpoint conventions."""
    import json

    from everyvoice_amd.fs2 import FastSpeech2

    cfg = FastSpeech2ConfigRef.small()
    cfg.n_speakers, cfg.n_languages = 4, 2
    torch.manual_seed(12)
    ref = FastSpeech2Ref(cfg).eval()
    randomize_norm_stats_(ref, torch.Generator().manual_seed(13))
    pc = _product_config(cfg)
    pc.multispeaker, pc.multilingual, pc.n_speakers, pc.n_languages = True, True, 4, 2
    model = FastSpeech2(pc, device=cuda_device, speaker2id={"a": 0, "b": 1, "c": 2, "d": 3}, lang2id={"x": 0, "y": 1}).load_state_dict(ref.state_dict())
    ids, lens, g = _batch(20, 3, 11, seed=2)
    durs = torch.randint(1, 5, (3, 11), generator=g)
    spk, lang = torch.tensor([3, 0, 2]), torch.tensor([1, 1, 0])
    want = ref(ids, lens, durations=durs, speakers=spk, languages=lang)
    got = model(ids, lens, durations=durs, speakers=spk, languages=lang)
    _close(got[1].cpu(), want[1])
    with pytest.raises(ValueError, match="speakers"):
        model(ids, lens, durations=durs)

    ckpt = model.to_checkpoint(ref.state_dict())
    json.dumps(ckpt["hyper_parameters"])  # JSON-only, like the reference's on_save_checkpoint
    assert ckpt["model_info"] == {"name": "FastSpeech2", "version": "1.0"}
    again = FastSpeech2.from_checkpoint(ckpt, device=cuda_device)
    assert again.speaker2id == {"a": 0, "b": 1, "c": 2, "d": 3}
    _close(again(ids, lens, durations=durs, speakers=spk, languages=lang)[1].cpu(), want[1])
    with pytest.raises(TypeError, match="Wrong model type"):
        FastSpeech2.from_checkpoint({**ckpt, "model_info": {"name": "HiFiGAN", "version": "1.0"}}, device=cuda_device)
    with pytest.raises(ValueError, match="newer version"):
        FastSpeech2.from_checkpoint({**ckpt, "model_info": {"name": "FastSpeech2", "version": "9.0"}}, device=cuda_device)


def test_config5_multispeaker_fs2_into_hop512_istft_vocoder(cuda_device):
    """BASELINE config 5 end to end at its own model size: multi-speaker FastSpeech2 (default 256-dim conformers, 4 speakers)
    -> iSTFTNet vocoder with upsampling 8 x 8 x 2 and iSTFT(16, 4) = hop 512 (44.1 kHz), against the oracle chain in fp32;
    the bf16 path within its relative-L2 bound."""
    from everyvoice_amd.fs2 import FastSpeech2
    from oracle.hifigan_ref import HiFiGANModelConfigRef
    from tests.helpers import make_ref_generator, rel_l2
    from tests.test_gpu_generator import _product_from_ref

    cfg = FastSpeech2ConfigRef()
    cfg.n_speakers = 4
    torch.manual_seed(21)
    ref = FastSpeech2Ref(cfg).eval()
    randomize_norm_stats_(ref, torch.Generator().manual_seed(22))
    pc = _product_config(cfg)
    pc.multispeaker, pc.n_speakers = True, 4
    fs2 = FastSpeech2(pc, device=cuda_device, speaker2id={f"s{i}": i for i in range(4)}).load_state_dict(ref.state_dict())
    ids, lens, g = _batch(cfg.n_symbols, 3, 37, seed=9)
    durs = torch.randint(1, 5, (3, 37), generator=g)
    spk = torch.tensor([2, 0, 3])
    want = ref(ids, lens, durations=durs, speakers=spk)
    got = fs2(ids, lens, durations=durs, speakers=spk)
    assert torch.equal(got[5].cpu(), want[5])
    _close(got[1].cpu(), want[1])
    voc_ref = make_ref_generator(HiFiGANModelConfigRef(istft_layer=True, upsample_rates=[8, 8, 2], upsample_kernel_sizes=[16, 16, 4]), seed=99)
    mel_ref_bct = want[1].detach().transpose(1, 2).contiguous()  # [B, 80, T]
    with torch.no_grad():
        wav_want = voc_ref(mel_ref_bct)
    assert wav_want.shape[-1] == 512 * mel_ref_bct.shape[-1]
    mel_got = got[1].transpose(1, 2).contiguous()
    wav32 = _product_from_ref(voc_ref, cuda_device, "f32")(mel_got).cpu()
    assert wav32.shape == wav_want.shape
    assert float((wav32 - wav_want).abs().max()) <= 1e-3 * max(1.0, float(wav_want.abs().max()))  # mel 2e-4 of its scale, through the vocoder
    wav16 = _product_from_ref(voc_ref, cuda_device, "bf16")(mel_got).cpu()
    assert torch.isfinite(wav16).all() and rel_l2(wav16, wav_want) <= 3e-2


def test_fs2_edge_cases(cuda_device):
    """One token, zero durations for some tokens, an all-zero item next to a normal one, max_length guard, duration_control."""
    from everyvoice_amd.fs2 import FastSpeech2

    ref, model = _models(FastSpeech2ConfigRef.small(), cuda_device, seed=77)
    # a single token, single item
    ids, lens = torch.tensor([[5]]), torch.tensor([1])
    durs = torch.tensor([[3]])
    want, got = ref(ids, lens, durations=durs), model(ids, lens, durations=durs)
    assert got[1].shape == (1, 3, 16)
    _close(got[1].cpu(), want[1])
    # tokens with zero duration vanish from the frames; item 1 is much shorter than item 0
    ids, lens, g = _batch(20, 2, 9, seed=5, lens=[9, 2])
    durs = torch.tensor([[2, 0, 0, 3, 1, 0, 4, 0, 1], [0, 1, 7, 7, 7, 7, 7, 7, 7]])  # entries past lens are ignored
    want, got = ref(ids, lens, durations=durs), model(ids, lens, durations=durs)
    assert torch.equal(got[5].cpu(), torch.tensor([11, 1])) and torch.equal(got[2].cpu(), want[2])
    _close(got[1].cpu(), want[1])
    # all durations zero everywhere: nothing to synthesise
    with pytest.raises(ValueError, match="zero"):
        model(ids, lens, durations=torch.zeros(2, 9, dtype=torch.long))
    # max_length guard of FastSpeech2ModelConfig
    model.config.max_length = 4
    with pytest.raises(ValueError, match="max_length"):
        model(ids, lens, durations=durs)
    model.config.max_length = 1000
    # duration_control scales the rounded predictions (here: 2x) -- integer-exact against the oracle
    with torch.no_grad():
        ref.duration_predictor.linear.bias.fill_(1.0)
    model.load_state_dict(ref.state_dict())
    a = model(ids, lens, duration_control=1.0)[2]
    b = model(ids, lens, duration_control=2.0)[2]
    assert torch.equal(b, 2 * a) and torch.equal(b.cpu(), ref(ids, lens, duration_control=2.0)[2])
