"""HiFiGAN generator on the GPU (through the C ABI) vs the fp32 CPU oracle.

Tolerances (stated per BASELINE.json north_star, "within a stated fp tolerance"):
  * EVMI_PREC_F32 (fp32 fmaf chains):  max |diff| <= 2e-4 on wav in [-1, 1]
  * EVMI_PREC_BF16 (bf16 operands / activations in HBM, fp32 accumulate on MFMA):
        relative L2 error <= 1e-2 and max |diff| <= 5e-2 on wav in [-1, 1]   (measured 4.8e-3 relative L2: a kernel change that
        doubles the error fails)
"""

import numpy as np
import pytest
import torch

from helpers import make_ref_generator, rel_l2, synthetic_mel

pytestmark = pytest.mark.gpu

F32_ATOL = 2e-4
BF16_REL_L2 = 1e-2
BF16_ATOL = 5e-2  # (a single-sample metric: measured 0.5e-2 ... 3.7e-2 over the input sizes and kernel variants)
# the iSTFTNet head puts exp() behind conv_post: its output error is the logit's error times the magnitude (measured 1.33e-2 on the
# hop-512 configuration, 6e-3 on C8C8I)
BF16_REL_L2_ISTFT = 2.5e-2


def _product_from_ref(ref, device, precision):
    from everyvoice_amd.config import HiFiGANConfig
    from everyvoice_amd.vocoder import HiFiGANGenerator

    c = ref.cfg
    cfg = HiFiGANConfig(model=dict(resblock=c.resblock, upsample_rates=c.upsample_rates,
                                   upsample_kernel_sizes=c.upsample_kernel_sizes,
                                   upsample_initial_channel=c.upsample_initial_channel,
                                   resblock_kernel_sizes=c.resblock_kernel_sizes,
                                   resblock_dilation_sizes=c.resblock_dilation_sizes, istft_layer=c.istft_layer))
    model = HiFiGANGenerator(cfg, precision=precision)
    model.load_state_dict({"generator." + k: v for k, v in ref.state_dict().items()})
    return model.to(device).eval()


@pytest.fixture(scope="module")
def ref_gen():
    torch.set_num_threads(8)
    return make_ref_generator(seed=1234)


@pytest.mark.parametrize("conv", [
    dict(cin=8, cout=12, k=3, stride=1, pad=1, dil=1, groups=1),
    dict(cin=16, cout=16, k=11, stride=1, pad=25, dil=5, groups=1),
    dict(cin=16, cout=32, k=41, stride=4, pad=20, dil=1, groups=4),  # MSD-style grouped strided conv
    dict(cin=1, cout=8, k=15, stride=1, pad=7, dil=1, groups=1),
])
def test_conv1d_f32_vs_torch(cuda_device, conv):
    import ctypes as C

    from everyvoice_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    B, T = 3, 97
    x = torch.randn(B, conv["cin"], T, generator=g)
    w = torch.randn(conv["cout"], conv["cin"] // conv["groups"], conv["k"], generator=g) * 0.2
    b = torch.randn(conv["cout"], generator=g)
    want = torch.nn.functional.conv1d(torch.nn.functional.leaky_relu(x, 0.1), w, b, conv["stride"], conv["pad"], conv["dil"], conv["groups"])
    res = torch.randn(want.shape, generator=g)
    y0 = torch.randn(want.shape, generator=g)
    y = y0.clone().to(cuda_device)
    xd, wd, bd, rd = (t.to(cuda_device) for t in (x, w, b, res))
    _lib.check(lib.evmi_conv1d_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), rd.data_ptr(), y.data_ptr(), B,
                                   conv["cin"], T, conv["cout"], conv["k"], conv["stride"], conv["pad"], conv["dil"],
                                   conv["groups"], 0.1, 0.5, 1, _lib.current_stream_ptr()))
    torch.testing.assert_close(y.cpu(), y0 + 0.5 * (want + res), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("u,k", [(8, 16), (2, 4), (3, 7)])
def test_conv_transpose1d_f32_vs_torch(cuda_device, u, k):
    from everyvoice_amd import _lib

    lib = _lib.load()
    g = torch.Generator().manual_seed(6)
    B, T, cin, cout = 2, 37, 12, 6
    p = (k - u) // 2
    x = torch.randn(B, cin, T, generator=g)
    w = torch.randn(cin, cout, k, generator=g) * 0.2
    b = torch.randn(cout, generator=g)
    want = torch.nn.functional.conv_transpose1d(torch.nn.functional.leaky_relu(x, 0.1), w, b, u, p)
    y = torch.empty(want.shape, device=cuda_device)
    xd, wd, bd = (t.to(cuda_device) for t in (x, w, b))
    _lib.check(lib.evmi_conv_transpose1d_f32(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), y.data_ptr(), B, cin, T, cout,
                                             k, u, p, 0.1, _lib.current_stream_ptr()))
    torch.testing.assert_close(y.cpu(), want, rtol=1e-5, atol=1e-5)


F32_MODES = ["f32", "f32-direct"]  # fp32 matrix-core kernels (channel-major) / direct vector-ALU kernels of the native object


@pytest.mark.parametrize("prec", F32_MODES)
@pytest.mark.parametrize("B,T", [(1, 1), (2, 6), (3, 37), (1, 130)])
def test_generator_f32_vs_oracle(cuda_device, ref_gen, B, T, prec):
    model = _product_from_ref(ref_gen, cuda_device, prec)
    mel = synthetic_mel(B, T, seed=100 + T)
    with torch.no_grad():
        want = ref_gen(mel)
    got = model(mel.to(cuda_device)).cpu()
    assert got.shape == want.shape == (B, 1, T * 256)
    assert float((got - want).abs().max()) <= F32_ATOL


@pytest.mark.parametrize("B,T", [(1, 1), (2, 6), (3, 37), (1, 130), (2, 257)])
def test_generator_bf16_vs_oracle(cuda_device, ref_gen, B, T):
    model = _product_from_ref(ref_gen, cuda_device, "bf16")
    mel = synthetic_mel(B, T, seed=100 + T)
    with torch.no_grad():
        want = ref_gen(mel)
    got = model(mel.to(cuda_device)).cpu()
    assert got.shape == want.shape
    assert torch.isfinite(got).all()
    err, mx = rel_l2(got, want), float((got - want).abs().max())
    print(f"bf16 B={B} T={T}: rel_l2={err:.3e} max_abs={mx:.3e}")
    assert err <= BF16_REL_L2 and mx <= BF16_ATOL


def test_generator_vs_committed_fixture(cuda_device, ref_gen, golden_dir):
    g = np.load(golden_dir / "hifigan_v1_small.npz")
    mel, want = torch.from_numpy(g["mel"]), torch.from_numpy(g["wav"])
    for prec in F32_MODES:
        got32 = _product_from_ref(ref_gen, cuda_device, prec)(mel.to(cuda_device)).cpu()
        assert float((got32 - want).abs().max()) <= F32_ATOL, prec
    got16 = _product_from_ref(ref_gen, cuda_device, "bf16")(mel.to(cuda_device)).cpu()
    assert rel_l2(got16, want) <= BF16_REL_L2


def test_upstream_init_weights_and_weight_norm_checkpoint(cuda_device):
    """The upstream N(0, 0.01) init (tiny activations) through a weight_g / weight_v checkpoint."""
    from everyvoice_amd.config import HiFiGANConfig
    from everyvoice_amd.vocoder import HiFiGANGenerator, load_hifigan_from_checkpoint
    from oracle.hifigan_ref import GeneratorRef

    torch.manual_seed(1234)
    ref = GeneratorRef().eval()
    ckpt = {"state_dict": {"generator." + k: v for k, v in ref.state_dict().items()},
            "hyper_parameters": {"config": HiFiGANConfig().model_dump(mode="json")},
            "model_info": {"name": "HiFiGAN", "version": "1.0"}}
    model, _ = load_hifigan_from_checkpoint(ckpt, cuda_device, precision="f32")
    mel = synthetic_mel(2, 9, seed=3)
    with torch.no_grad():
        want = ref(mel)
    got = model(mel.to(cuda_device)).cpu()
    assert float((got - want).abs().max()) <= 1e-5
    model.generator.precision = "bf16"
    got = model(mel.to(cuda_device)).cpu()
    assert float((got - want).abs().max()) <= 2e-3


def test_full_size_properties_bf16(cuda_device, ref_gen):
    """BASELINE config 2 (B=32, T=768): checks that do not need the oracle at full size.
    (a) items are independent: item b of the batch == the same item run alone, bitwise;
    (b) the interior of a 96-frame window run alone == the same samples of the long run within the
        bf16 tolerance (receptive field < 24 frames per side), and that window matches the oracle."""
    model = _product_from_ref(ref_gen, cuda_device, "bf16")
    mel = synthetic_mel(32, 768, seed=1234).to(cuda_device)
    wav = model(mel)
    assert wav.shape == (32, 1, 768 * 256) and torch.isfinite(wav).all()
    for b in (0, 17, 31):
        alone = model(mel[b : b + 1].contiguous())
        assert torch.equal(alone[0], wav[b])
    lo, hi, guard = 300, 396, 24
    win = mel[5:6, :, lo:hi].contiguous()
    wav_win = model(win)[0, 0, guard * 256 : (hi - lo - guard) * 256].cpu()
    wav_long = wav[5, 0, (lo + guard) * 256 : (hi - guard) * 256].cpu()
    assert rel_l2(wav_win, wav_long) <= BF16_REL_L2
    with torch.no_grad():
        want = ref_gen(win.cpu())[0, 0, guard * 256 : (hi - lo - guard) * 256]
    assert rel_l2(wav_long, want) <= BF16_REL_L2


def test_no_cpu_fallback():
    from everyvoice_amd.config import HiFiGANConfig
    from everyvoice_amd.vocoder import HiFiGANGenerator

    with pytest.raises(RuntimeError, match="GPU only"):
        HiFiGANGenerator(HiFiGANConfig())(torch.zeros(1, 80, 4))


ISTFT_CONFIGS = {
    # the reference's own test config (everyvoice/tests/data/relative/config/everyvoice-spec-to-wav.yaml): C8C8I
    "test_config_c8c8i": dict(istft_layer=True, upsample_rates=[8, 8], upsample_kernel_sizes=[16, 16]),
    # BASELINE config 5 style: hop 512 = 8*8*2 * istft hop 4
    "hop512_c8c8c2i": dict(istft_layer=True, upsample_rates=[8, 8, 2], upsample_kernel_sizes=[16, 16, 4]),
}


@pytest.mark.parametrize("name", sorted(ISTFT_CONFIGS))
@pytest.mark.parametrize("B,T", [(1, 1), (2, 9), (1, 70)])
def test_istft_generator_vs_oracle(cuda_device, name, B, T):
    from oracle.hifigan_ref import HiFiGANModelConfigRef

    cfg = HiFiGANModelConfigRef(**ISTFT_CONFIGS[name])
    ref = make_ref_generator(cfg, seed=4321)
    mel = synthetic_mel(B, T, seed=7 + T)
    with torch.no_grad():
        want = ref(mel)
    assert want.shape == (B, 1, T * ref.hop)
    scale = float(want.abs().max())
    for prec in F32_MODES:
        got32 = _product_from_ref(ref, cuda_device, prec)(mel.to(cuda_device)).cpu()
        assert got32.shape == want.shape
        assert float((got32 - want).abs().max()) <= 2e-4 * max(1.0, scale), prec
    got16 = _product_from_ref(ref, cuda_device, "bf16")(mel.to(cuda_device)).cpu()
    err = rel_l2(got16, want)
    print(f"istft {name} B={B} T={T}: bf16 rel_l2={err:.3e} (|wav| max {scale:.2f})")
    assert torch.isfinite(got16).all() and err <= BF16_REL_L2_ISTFT


_GEN_SWITCHES = ["EVMI_CONV_DMA=0", "EVMI_PAIR_C128=0", "EVMI_BRANCH=0"]


@pytest.mark.gpu
@pytest.mark.parametrize("switch", _GEN_SWITCHES)
def test_kernel_variants_behind_switches_match_the_oracle(switch):
    """The library picks its inference kernels once per process: the variants behind the A/B switches (register-staged convolutions,
    unfused 128-channel pairs, pair kernels instead of the whole-branch kernels) run the oracle
    comparisons of this file in a child process each (started together, tests/helpers.py), so a switch that is off by default cannot rot."""
    from helpers import child_pytest_results

    jobs = {}
    for sw in _GEN_SWITCHES:
        key, val = sw.split("=")
        jobs[sw] = ([__file__, "-q", "-x", "-k", "bf16_vs_oracle or committed_fixture or full_size_properties or istft_generator"], {key: val})
    rc, out = child_pytest_results("generator_switches", jobs, parallel=2, timeout=600)[switch]
    assert rc == 0, out


_BRANCH_CHILD = """
import sys, torch
sys.path.insert(0, {root!r})
import bench
torch.manual_seed(0)
model = bench.upstream_init_generator("bf16").to("cuda:0").eval()
for B, T in ((3, 40), (2, 301), (8, 768)):
    wav = model.generator(bench.synthetic_mel(B, T, 99 + T).to("cuda:0"))
    torch.save(wav.cpu(), {out!r} + f"/wav_{{B}}_{{T}}.pt")
"""


@pytest.mark.gpu
def test_whole_branch_kernel_gives_the_bits_of_the_pair_kernels(tmp_path, cuda_device):
    """resblock_branch_kernel.h keeps a branch's running value in LDS over its three pairs; its rounding points are the pair kernel's
    (bf16 where that one stores), so the waveform must be IDENTICAL to the one of a process with EVMI_BRANCH=0 -- at a tile-ragged
    length, at a length shorter than one tile's halo, and at the bench's 768 frames (every tile position of the persistent walk)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parents[1])
    outs = {}
    for flag in ("1", "0"):
        d = tmp_path / f"branch{flag}"
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", _BRANCH_CHILD.format(root=root, out=str(d))], env=dict(os.environ, EVMI_BRANCH=flag),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[flag] = {p.name: torch.load(p) for p in sorted(d.iterdir())}
    assert sorted(outs["1"]) == sorted(outs["0"]) and len(outs["1"]) == 3
    for name, a in outs["1"].items():
        b = outs["0"][name]
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


_NARROW_CHILD = """
import sys, torch
sys.path.insert(0, {root!r})
import bench
torch.manual_seed(0)
model = bench.upstream_init_generator("bf16").to("cuda:0").eval()
for B, T in ((1, 8), (2, 40), (3, 130), (16, 32)):
    wav = model.generator(bench.synthetic_mel(B, T, 99 + T).to("cuda:0"))
    torch.save(wav.cpu(), {out!r} + f"/wav_{{B}}_{{T}}.pt")
"""


@pytest.mark.gpu
def test_narrow_row_tiles_give_the_bits_of_the_256_row_tiles(tmp_path, cuda_device):
    """Round 6: where the 256-row tiles of the wide residual-stack convolutions (conv_tc_dma_kernel.h) would leave most of the chip idle
    -- one utterance at a time, the GAN step's generator at 16 x 32 frames: 32 .. 128 workgroups on 256 CUs -- launch_conv_tc picks the
    same kernel on 128-row tiles.  An output element's K order (channel chunk, tap) does not depend on the tile, so the waveform is
    IDENTICAL to a process with EVMI_CONV_NARROW=0, at lengths whose stages take narrow tiles, wide tiles, and a mix (the bench's
    training shape, 16 x 32 frames, among them)."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = str(Path(__file__).resolve().parents[1])
    outs = {}
    for flag in ("1", "0"):
        d = tmp_path / f"narrow{flag}"
        d.mkdir()
        r = subprocess.run([sys.executable, "-c", _NARROW_CHILD.format(root=root, out=str(d))], env=dict(os.environ, EVMI_CONV_NARROW=flag),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs[flag] = {p.name: torch.load(p) for p in sorted(d.iterdir())}
    assert sorted(outs["1"]) == sorted(outs["0"]) and len(outs["1"]) == 4
    for name, a in outs["0"].items():
        assert torch.isfinite(a).all() and float(a.abs().max()) > 0
        b = outs["1"][name]
        assert torch.equal(a, b), (name, float((a - b).abs().max()))
