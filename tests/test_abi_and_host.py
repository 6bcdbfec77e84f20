"""CPU-side checks: the C-ABI library loads and exports every symbol include/evmi.h declares
(no compute calls without a GPU), and the host-side mirror (config, checkpoints, weight-norm
folding) behaves like the reference's interface."""

import ctypes as C
import re
from pathlib import Path

import pytest
import torch

from everyvoice_amd import _lib
from everyvoice_amd.config import HiFiGANConfig
from everyvoice_amd.vocoder import Generator, HiFiGANGenerator, fold_weight_norm_, load_hifigan_from_checkpoint
from oracle.hifigan_ref import GeneratorRef, HiFiGANModelConfigRef, count_params

ROOT = Path(__file__).resolve().parent.parent


def declared_symbols():
    text = (ROOT / "include" / "evmi.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(evmi_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()  # raises if the .so is missing: there is no fallback
    names = declared_symbols()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/evmi.h but not exported"
    assert set(names) == set(_lib.SYMBOLS), "ctypes table and header disagree"
    assert lib.evmi_abi_version() == 2


def test_generator_object_without_gpu():
    """create / weight_info / macs are host-only: safe without a device."""
    g = Generator(HiFiGANConfig())
    assert g.macs_per_sample() == 1_199_424.0  # HiFi-GAN V1 MAC per output sample
    assert g._handle is None  # asking for the MAC count does not pin the module to a device
    g._new_handle(0)  # (creating the native object is host-only too)
    lib = _lib.load()
    n = lib.evmi_generator_num_weights(g._handle)
    names = []
    for i in range(n):
        buf = C.create_string_buffer(96)
        numel = C.c_int64()
        _lib.check(lib.evmi_generator_weight_info(g._handle, i, buf, 96, C.byref(numel)))
        names.append((buf.value.decode(), numel.value))
    mine = {k: v.numel() for k, v in g.state_dict().items()}
    assert dict(names) == mine
    assert sum(mine.values()) == 13_926_017
    istft = Generator(HiFiGANConfig(model=dict(istft_layer=True, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16])))
    assert istft.macs_per_sample() == 803_872.0 and istft.hop == 256
    assert sum(p.numel() for p in istft.parameters()) == 13_254_034  # everyvoice/tests/test_cli.py:363


def test_errors_are_reported_not_swallowed():
    lib = _lib.load()
    bad = _lib.GeneratorConfig()
    h = C.c_void_p()
    rc = lib.evmi_generator_create(C.byref(bad), 0, C.byref(h))
    assert rc != 0 and b"config" in lib.evmi_last_error()
    with pytest.raises(_lib.EvmiError):
        _lib.check(rc, "evmi_generator_create")
    g = Generator(HiFiGANConfig())
    g.macs_per_sample()
    x = torch.zeros(3)
    assert lib.evmi_generator_set_weight(g._handle, b"nope.weight", x.data_ptr(), 3) != 0
    assert lib.evmi_generator_set_weight(g._handle, b"conv_post.bias", x.data_ptr(), 3) != 0  # wrong size
    assert lib.evmi_generator_forward(g._handle, 1, 1, 1, 1, 0, None) != 0  # not finalised
    with pytest.raises(RuntimeError, match="GPU only"):
        g(torch.zeros(1, 80, 4))


def test_weight_norm_checkpoint_loads_folded():
    torch.manual_seed(0)
    ref = GeneratorRef()
    sd = {"generator." + k: v for k, v in ref.state_dict().items()}
    model = HiFiGANGenerator(HiFiGANConfig())
    model.load_state_dict(sd)
    ref.remove_weight_norm()
    want = ref.state_dict()
    got = model.generator.state_dict()
    assert sorted(got.keys()) == sorted(want.keys())
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=1e-6, atol=1e-7)
    assert count_params(model) == 13_926_017


def test_fold_matches_torch_weight_norm_for_transposed_conv():
    conv = torch.nn.utils.weight_norm(torch.nn.ConvTranspose1d(6, 4, 4, 2, padding=1))
    sd = dict(conv.state_dict())
    fold_weight_norm_(sd)
    torch.testing.assert_close(sd["weight"], conv.weight.detach())


def test_checkpoint_round_trip_and_errors(tmp_path):
    cfg = HiFiGANConfig(model=dict(istft_layer=True, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16]))
    model = HiFiGANGenerator(cfg)
    ckpt = model.to_checkpoint()
    assert ckpt["model_info"] == {"name": "HiFiGANGenerator", "version": "1.0"}
    import json

    json.dumps(ckpt["hyper_parameters"])  # JSON-only, path-free (everyvoice/tests/test_model.py:85-151)
    torch.save(ckpt, tmp_path / "g.ckpt")
    loaded, cfg2 = load_hifigan_from_checkpoint(torch.load(tmp_path / "g.ckpt", weights_only=False), "cpu")
    assert cfg2.model.istft_layer and isinstance(loaded, HiFiGANGenerator)
    for (k1, v1), (k2, v2) in zip(model.state_dict().items(), loaded.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)
    bad = dict(ckpt, model_info={"name": "FastSpeech2", "version": "1.0"})
    with pytest.raises(TypeError, match="Wrong model type"):
        load_hifigan_from_checkpoint(bad, "cpu")
    with pytest.raises(TypeError, match="Unable to load config"):
        load_hifigan_from_checkpoint({"state_dict": {}, "hyper_parameters": {"config": {"model": {"bogus": 1}}}}, "cpu")
    with pytest.raises(TypeError):
        load_hifigan_from_checkpoint({"state_dict": {"x": torch.zeros(1)}, "hyper_parameters": {"config": {}}}, "cpu")


def test_reference_test_config_ref_matches_host_config():
    ref = HiFiGANModelConfigRef.test_config()
    mine = HiFiGANConfig(model=dict(istft_layer=True, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16])).model
    for f in ("upsample_rates", "upsample_kernel_sizes", "resblock_kernel_sizes", "resblock_dilation_sizes", "mpd_layers"):
        assert getattr(ref, f) == getattr(mine, f)


def test_trainer_state_dict_loads_strictly_into_the_upstream_modules():
    """A checkpoint trained here must resume upstream: every exported tensor has upstream's name AND shape -- the period
    discriminators are Conv2d((k, 1)) there (weight_v [c_out, c_in, k, 1], weight_g [c_out, 1, 1, 1]), Conv1d on the period
    view here.  Host-only: constructing the trainer and exporting / importing its state runs no kernel."""
    from everyvoice_amd.train.hifigan import HiFiGANTrainer
    from oracle.hifigan_ref import MultiPeriodDiscriminatorRef, MultiScaleDiscriminatorRef

    tr = HiFiGANTrainer(device="cpu")
    sd = tr.state_dict()
    strip = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}  # noqa: E731
    g, mpd, msd = GeneratorRef(), MultiPeriodDiscriminatorRef(), MultiScaleDiscriminatorRef()
    for mod, pre in ((g, "generator."), (mpd, "mpd."), (msd, "msd.")):
        res = mod.load_state_dict(strip(pre), strict=True)
        assert not res.missing_keys and not res.unexpected_keys
    assert sd["mpd.discriminators.0.convs.1.weight_v"].shape == (128, 32, 5, 1)
    assert sd["mpd.discriminators.0.convs.1.weight_g"].shape == (128, 1, 1, 1)
    # and back: the upstream-shaped tensors load into a fresh trainer bit for bit
    tr2 = HiFiGANTrainer(device="cpu", seed=7)
    tr2.load_checkpoint(tr.checkpoint())
    assert torch.equal(tr2.d_params.flat, tr.d_params.flat) and torch.equal(tr2.g_params.flat, tr.g_params.flat)


def test_a_branch_keeps_every_tensor_its_tape_operators_can_reach():
    """train/fs2.py: _held_tensors -- what a side branch holds until the main chain has joined it: tensors behind Vars, dicts (the
    packed operands an operator clears when it has launched), lists and nested closures."""
    import torch

    from everyvoice_amd.train.autograd import Var
    from everyvoice_amd.train.fs2 import _held_tensors

    a, b, c, d, e = (torch.zeros(i + 1) for i in range(5))
    v = Var(a)
    v.grad = b
    packed = {"x_packed": c}
    inner = lambda: d  # noqa: E731

    def op():
        return v, packed, inner, [e]

    held = _held_tensors([op])
    assert {t.numel() for t in held} == {1, 2, 3, 4, 5}
    packed.clear()  # (what the operator does when it has launched its kernels)
    assert any(t is c for t in held)


def test_the_row_pitch_of_a_shared_packed_operand():
    """train/ops.py: pk_pitch mirrors csrc/conv_pk_common.h: pk_shared_pitch -- tight items, or one item rounded up to 64 units."""
    from everyvoice_amd.train import ops

    assert ops.pk_pitch(32, 814) == 32 * 814 and ops.pk_pitch(1, 4513) == 4544 and ops.pk_pitch(1, 64) == 64 and ops.pk_pitch(1, 1) == 64
