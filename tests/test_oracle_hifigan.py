"""Structural pins of the vocoder oracle: the exact parameter counts the reference's tests assert
and the defaults frozen in its published schema."""

import json
from pathlib import Path

import pytest
import torch

from oracle.hifigan_ref import (
    GeneratorRef,
    HiFiGANModelConfigRef,
    MultiPeriodDiscriminatorRef,
    MultiScaleDiscriminatorRef,
    count_params,
)


def test_v1_generator_param_count():
    g = GeneratorRef()
    g.remove_weight_norm()
    assert count_params(g) == 13_926_017  # HiFi-GAN V1, the paper's 13.92 M


def test_reference_test_config_param_counts():
    cfg = HiFiGANModelConfigRef.test_config()
    g = GeneratorRef(cfg)
    mpd = MultiPeriodDiscriminatorRef(cfg.mpd_layers)
    msd = MultiScaleDiscriminatorRef(cfg.msd_layers)
    # everyvoice/tests/test_cli.py:340 — trainable params of the full HiFiGAN module
    assert count_params(g) + count_params(mpd) + count_params(msd) == 83_986_835
    g.remove_weight_norm()
    # everyvoice/tests/test_cli.py:363 — exported generator, weight norm folded
    assert count_params(g) == 13_254_034


def test_state_dict_uses_upstream_names():
    keys = set(GeneratorRef().state_dict().keys())
    for k in ("conv_pre.weight_g", "conv_pre.weight_v", "ups.3.bias", "resblocks.11.convs2.2.weight_v", "conv_post.bias"):
        assert k in keys


def test_shapes_v1_and_istft():
    mel = torch.randn(2, 80, 5)
    with torch.no_grad():
        assert GeneratorRef().eval()(mel).shape == (2, 1, 5 * 256)
        g = GeneratorRef(HiFiGANModelConfigRef.test_config()).eval()
        assert g.hop == 256 and g(mel).shape == (2, 1, 5 * 256)


@pytest.mark.skipif(not Path("/root/reference").exists(), reason="reference tree only exists in the build container")
def test_defaults_match_reference_schema():
    schema = json.loads(Path("/root/reference/everyvoice/.schema/everyvoice-spec-to-wav-0.5.json").read_text())
    props = schema["$defs"]["HiFiGANModelConfig"]["properties"]
    cfg = HiFiGANModelConfigRef()
    for field in ("upsample_rates", "upsample_kernel_sizes", "upsample_initial_channel", "resblock_kernel_sizes",
                  "resblock_dilation_sizes", "istft_layer", "msd_layers", "mpd_layers"):
        assert getattr(cfg, field) == props[field]["default"], field
    from everyvoice_amd.config import AudioConfig, HiFiGANModelConfig

    mine = HiFiGANModelConfig()
    for field, spec in props.items():
        if "default" in spec:
            assert getattr(mine, field) == spec["default"], field
    audio = AudioConfig()
    for field, spec in schema["$defs"]["AudioConfig"]["properties"].items():
        if "default" in spec:
            assert getattr(audio, field) == spec["default"], field
