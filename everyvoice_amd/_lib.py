"""ctypes binding of libevmi_hip.so (the C ABI declared in include/evmi.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing the
compute modules fails loudly — the product never silently runs a CPU or eager-PyTorch path.
"""

from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_NAME = "libevmi_hip.so"

EVMI_OK = 0
EVMI_PREC_BF16 = 0
EVMI_PREC_F32 = 1
EVMI_MAX_UPSAMPLES = 8
EVMI_MAX_RESBLOCK_KERNELS = 8
EVMI_MAX_DILATIONS = 8


class EvmiError(RuntimeError):
    """A libevmi_hip call returned a non-zero status."""


class GeneratorConfig(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int),
        ("upsample_initial_channel", C.c_int),
        ("num_upsamples", C.c_int),
        ("upsample_rates", C.c_int * EVMI_MAX_UPSAMPLES),
        ("upsample_kernel_sizes", C.c_int * EVMI_MAX_UPSAMPLES),
        ("resblock_type", C.c_int),
        ("num_kernels", C.c_int),
        ("resblock_kernel_sizes", C.c_int * EVMI_MAX_RESBLOCK_KERNELS),
        ("num_dilations", C.c_int * EVMI_MAX_RESBLOCK_KERNELS),
        ("resblock_dilations", (C.c_int * EVMI_MAX_DILATIONS) * EVMI_MAX_RESBLOCK_KERNELS),
        ("lrelu_slope", C.c_float),
        ("post_lrelu_slope", C.c_float),
        ("istft_layer", C.c_int),
        ("istft_n_fft", C.c_int),
        ("istft_hop", C.c_int),
    ]


class LaunchRecord(C.Structure):
    _fields_ = [
        ("kernel", C.c_char * 48),
        ("layer", C.c_char * 48),
        ("ms", C.c_float),
        ("flops", C.c_double),
        ("bytes", C.c_double),
    ]


class PkFlatJob(C.Structure):  # evmi_pkflat_job
    _fields_ = [("mode", C.c_int), ("c_in", C.c_int), ("c_out", C.c_int), ("k", C.c_int), ("stride", C.c_int), ("groups", C.c_int),
                ("w", C.c_void_p), ("wf", C.c_void_p), ("wf_elems", C.c_longlong)]


class PkFlatPair(C.Structure):  # evmi_pkflat_pair
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("units", C.c_longlong), ("plane", C.c_longlong), ("rows", C.c_int), ("scale", C.c_float)]


class PkFlatRows(C.Structure):  # evmi_pkflat_rows
    _fields_ = [("dy", C.c_void_p), ("plane", C.c_longlong), ("units", C.c_longlong), ("C", C.c_int), ("db", C.c_void_p)]


class LnPartials(C.Structure):  # evmi_ln_partials
    _fields_ = [("ws", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("C", C.c_int), ("n_cols", C.c_longlong)]


# name -> (restype, argtypes); every symbol include/evmi.h declares
SYMBOLS = {
    "evmi_tm_colsum_batch_bf16": (C.c_int, [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p, C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    "evmi_conv_pkflat_ws_elems": (C.c_longlong, [C.c_int] * 10),
    "evmi_conv_pkflat_plan": (C.c_int, [C.c_int] * 10),
    "evmi_conv_pkflat_tab": (C.c_int, [C.c_int] * 10 + [C.c_void_p, C.c_longlong, C.c_void_p]),
    "evmi_conv_pkflat_frag_elems": (C.c_longlong, [C.c_int] * 6),
    "evmi_conv_pkflat_fragments": (C.c_int, [C.c_int, C.POINTER(PkFlatJob), C.c_void_p]),
    "evmi_conv_pkflat_fwd": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong]
                             + [C.c_int] * 12 + [C.c_float, C.c_void_p]),
    "evmi_conv_pkflat_dgrad": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong] + [C.c_int] * 11
                               + [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "evmi_conv_pkflat_wgrad_ws_elems": (C.c_longlong, [C.c_int] * 8),
    "evmi_conv_pkflat_wgrad": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 10
                               + [C.c_void_p]),
    "evmi_pkflat_zero": (C.c_int, [C.c_void_p, C.c_longlong, C.c_void_p]),
    "evmi_disc_first_fwd": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 6
                            + [C.c_float, C.c_void_p]),
    "evmi_disc_first_wgrad_ws_elems": (C.c_longlong, [C.c_int] * 4),
    "evmi_disc_first_wgrad": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_longlong] + [C.c_int] * 5 + [C.c_void_p]),
    "evmi_disc_first_dgrad": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "evmi_disc_post_fwd_ws_elems": (C.c_longlong, [C.c_int] * 3),
    "evmi_disc_post_fwd": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                     C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "evmi_disc_post_dgrad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 6
                             + [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_float, C.c_float, C.c_void_p]),
    "evmi_disc_post_wgrad_ws_elems": (C.c_longlong, [C.c_int] * 4),
    "evmi_disc_post_wgrad": (C.c_int, [C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong]
                             + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_pkflat_absdiff_ws_elems": (C.c_longlong, [C.c_int]),
    "evmi_pkflat_absdiff": (C.c_int, [C.c_int, C.POINTER(PkFlatPair), C.c_void_p, C.c_void_p, C.c_longlong, C.c_void_p]),
    "evmi_pkflat_rowsum_ws_elems": (C.c_longlong, [C.c_int, C.POINTER(PkFlatRows)]),
    "evmi_pkflat_rowsum": (C.c_int, [C.c_int, C.POINTER(PkFlatRows), C.c_void_p, C.c_longlong, C.c_void_p]),
    "evmi_abi_version": (C.c_int, []),
    "evmi_last_error": (C.c_char_p, []),
    "evmi_device_info": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "evmi_length_regulate": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "evmi_length_regulate_bwd_f32": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
    ),
    "evmi_conv1d_f32": (
        C.c_int,
        [C.c_void_p] * 5 + [C.c_int] * 9 + [C.c_float, C.c_float, C.c_int, C.c_void_p],
    ),
    "evmi_conv_transpose1d_f32": (
        C.c_int,
        [C.c_void_p] * 4 + [C.c_int] * 7 + [C.c_float, C.c_void_p],
    ),
    "evmi_mel_spectrogram_f32": (
        C.c_int,
        [C.c_void_p] * 6 + [C.c_int] * 7 + [C.c_void_p],
    ),
    "evmi_mel_spectrogram_ragged_f32": (
        C.c_int,
        [C.c_void_p] * 7 + [C.c_int] * 7 + [C.c_void_p],
    ),
    "evmi_spectrogram_layout_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "evmi_gemm_f32": (C.c_int, [C.c_int] * 5 + [C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p]),
    "evmi_conv1d_cbt_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong] + [C.c_int] * 15 + [C.c_float, C.c_void_p]),
    "evmi_conv1d_cbt_bf16": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong] + [C.c_int] * 15 + [C.c_float, C.c_void_p]),
    "evmi_conv1d_cbt_bf16pk": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong] + [C.c_int] * 15 + [C.c_float, C.c_void_p]),
    "evmi_conv1d_cbt_bf16pk_ws_elems": (C.c_longlong, [C.c_int] * 10),
    "evmi_conv1d_cbt_bf16pk_plan": (C.c_int, [C.c_int] * 10),
    "evmi_conv1d_cbt_bf16_rounds": (C.c_int, [C.c_int] * 10),
    "evmi_conv1d_dgrad_cbt_bf16pk_plan": (C.c_int, [C.c_int] * 10),
    "evmi_conv1d_wgrad_cbt_bf16pk_plan": (C.c_int, [C.c_int] * 10),
    "evmi_conv1d_dgrad_cbt_bf16pk": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 10 + [C.c_void_p]),
    "evmi_conv1d_dgrad_cbt_bf16pk_ws_elems": (C.c_longlong, [C.c_int] * 10),
    "evmi_conv1d_cbt_f32_ws_elems": (C.c_longlong, [C.c_int] * 6),
    "evmi_conv1d_cbt_f32_supported": (C.c_int, [C.c_int] * 9),
    "evmi_conv1d_dgrad_cbt_f32_ws_elems": (C.c_longlong, [C.c_int] * 10),
    "evmi_conv1d_dgrad_cbt_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 10 + [C.c_void_p]),
    "evmi_conv1d_dgrad_cbt_bf16": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 10 + [C.c_void_p]),
    "evmi_weight_norm_fwd_batched_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_longlong, C.c_longlong, C.c_void_p]),
    "evmi_weight_norm_bwd_batched_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.c_longlong, C.c_longlong, C.c_void_p]),
    "evmi_istft_polar_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]),
    "evmi_istft_polar_bwd_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]),
    "evmi_reflect_pad_left1_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    "evmi_conv1d_wgrad_cbt_bf16pk_ws_elems": (C.c_longlong, [C.c_int] * 10),
    "evmi_conv1d_wgrad_cbt_bf16pk": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 11 + [C.c_void_p]),
    "evmi_conv1d_cbt_bf16pk_fused": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong] + [C.c_int] * 12 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "evmi_conv1d_dgrad_cbt_bf16pk_fused": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 10 + [C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "evmi_conv1d_dgrad_cbt_bf16pk_staged": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 10 + [C.c_void_p]),
    "evmi_layernorm_pack_bf16pk": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "evmi_layernorm_pack_bf16pk_w": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 4 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong,
                                                                                                   C.c_int, C.c_void_p]),
    "evmi_conv1d_cbt_bf16pk_prepacked": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_void_p]),
    "evmi_conv1d_cbt_bf16pk_silu_dropout": (C.c_int, [C.c_void_p] * 5 + [C.c_longlong] + [C.c_int] * 6 + [C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "evmi_conv1d_cbt_bf16pk_resdrop": (C.c_int, [C.c_int] + [C.c_void_p] * 6 + [C.c_longlong] + [C.c_int] * 4 + [C.c_float, C.c_ulonglong, C.c_float, C.c_ulonglong,
                                                  C.c_float, C.c_void_p, C.c_void_p]),
    "evmi_conv1d_dgrad_cbt_bf16pk_staged_dropout": (C.c_int, [C.c_int, C.c_void_p, C.c_float, C.c_ulonglong, C.c_void_p, C.c_float] + [C.c_void_p] * 3
                                                    + [C.c_longlong] + [C.c_int] * 10 + [C.c_void_p]),
    "evmi_conv1d_dgrad_cbt_bf16pk_staged_silu_dropout": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_ulonglong, C.c_void_p] + [C.c_void_p] * 3
                                                         + [C.c_longlong] + [C.c_int] * 10 + [C.c_void_p]),
    "evmi_conv1d_cbt_bf16pk_ffn_up": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 5
                                      + [C.c_float, C.c_ulonglong, C.c_void_p, C.c_int, C.c_void_p]),
    "evmi_conv1d_dgrad_cbt_bf16pk_ffn_down": (C.c_int, [C.c_void_p] * 2 + [C.c_longlong, C.c_void_p, C.c_void_p, C.c_longlong] + [C.c_int] * 5
                                              + [C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "evmi_conv1d_wgrad_cbt_bf16pk_prepacked": (C.c_int, [C.c_void_p] * 6 + [C.c_longlong] + [C.c_int] * 11 + [C.c_void_p]),
    "evmi_conv1d_bf16pk_shares_packed": (C.c_int, [C.c_int] * 7),
    "evmi_conv1d_wgrad_cbt_bf16pk_fused": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 11 + [C.c_float, C.c_void_p, C.c_float, C.c_void_p]),
    "evmi_conv1d_wgrad_cbt_f32_ws_elems": (C.c_longlong, [C.c_int] * 10),
    "evmi_conv1d_wgrad_cbt_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 11 + [C.c_void_p]),
    "evmi_fs2_embed_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 3 + [C.c_void_p]),
    "evmi_fs2_add_posemb_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "evmi_mask_cols_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_int] * 3 + [C.c_void_p]),
    "evmi_layernorm_cbt_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    "evmi_dwconv1d_cbt_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 6 + [C.c_void_p]),
    "evmi_fs2_bucket_embed_add_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "evmi_fs2_durations_i32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 2 + [C.c_float, C.c_void_p]),
    "evmi_length_regulate_cbt_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_attention_cbt_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_attention_cbt_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_fs2_add_item_embedding_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p]),
    "evmi_attention_prior_f64": (C.c_int, [C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_align_attention_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "evmi_forward_sum_loss_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_float, C.c_void_p]),
    "evmi_binarization_partials_f64": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_longlong, C.c_void_p]),
    "evmi_monotonic_align_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p]),
    "evmi_layernorm_bwd_cbt_f32_ws_elems": (C.c_longlong, [C.c_int, C.c_longlong]),
    "evmi_layernorm_bwd_cbt_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_longlong, C.c_int, C.c_longlong, C.c_float, C.c_int, C.c_void_p]),
    "evmi_layernorm_bwd_partials_reduce": (C.c_int, [C.c_int, C.POINTER(LnPartials), C.c_void_p]),
    "evmi_batchnorm_fwd_cbt_f32": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_longlong, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "evmi_batchnorm_bwd_cbt_f32": (C.c_int, [C.c_void_p] * 9 + [C.c_int, C.c_longlong, C.c_int, C.c_void_p]),
    "evmi_dwconv1d_bwd_cbt_f32_ws_elems": (C.c_longlong, [C.c_int] * 3),
    "evmi_dwconv1d_bwd_cbt_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_longlong] + [C.c_int] * 5 + [C.c_void_p]),
    "evmi_mha_fwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "evmi_mha_fwd_bf16": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "evmi_mha_bwd_bf16": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "evmi_mha_bwd_f32": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 4 + [C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "evmi_softmax_rows_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_float, C.c_ulonglong, C.c_void_p]),
    "evmi_softmax_bwd_rows_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_longlong, C.c_int, C.c_float, C.c_float, C.c_ulonglong, C.c_void_p]),
    "evmi_glu_bwd_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_void_p]),
    "evmi_dropout_f32": (C.c_int, [C.c_void_p] * 2 + [C.c_longlong, C.c_float, C.c_ulonglong, C.c_void_p, C.c_void_p]),
    "evmi_dropout_fused_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 3 + [C.c_longlong, C.c_float, C.c_ulonglong, C.c_void_p, C.c_float, C.c_void_p]),
    "evmi_fs2_embed_bwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_void_p]),
    "evmi_fs2_bucket_embed_bwd_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_float, C.c_void_p]),
    "evmi_fs2_item_embedding_bwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_length_regulate_bwd_cbt_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_forward_sum_grad_f32_ws_elems": (C.c_longlong, [C.c_int] * 3),
    "evmi_forward_sum_grad_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_longlong] + [C.c_int] * 3 + [C.c_float, C.c_float, C.c_void_p]),
    "evmi_align_attention_bwd_f32": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 3 + [C.c_float, C.c_void_p, C.c_void_p]),
    "evmi_align_qk_grad_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_longlong, C.c_float, C.c_void_p]),
    "evmi_dgrad_weights_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]),
    "evmi_gemm_batched_f32": (C.c_int, [C.c_int] * 5 + [C.c_float, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_void_p, C.c_int, C.c_longlong, C.c_int, C.c_void_p]),
    "evmi_unfold_cbt_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 8 + [C.c_void_p]),
    "evmi_fold_cbt_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 9 + [C.c_void_p]),
    "evmi_bias_add_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_void_p]),
    "evmi_lrelu_bwd_rowsum_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_longlong, C.c_float, C.c_int, C.c_void_p]),
    "evmi_row_reduce_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_int, C.c_void_p]),
    "evmi_elementwise_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float, C.c_void_p]),
    "evmi_scalar_reduce_f32": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_float, C.c_int, C.c_void_p]),
    "evmi_avgpool4s2_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    "evmi_period_view_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "evmi_stft_frames_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "evmi_weight_norm_fwd_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p]),
    "evmi_weight_norm_bwd_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    "evmi_normalize_vec_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "evmi_conv_tc_supported": (C.c_int, [C.c_int] * 4),
    "evmi_conv_tc_relayout_f32": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_void_p]),
    "evmi_conv_tc_tile_layout": (C.c_int, [C.c_int] * 4 + [C.POINTER(C.c_int)] * 3),
    "evmi_conv_tc_relayout_batched_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 3 + [C.c_void_p]),
    "evmi_conv_tc_tm_bf16": (C.c_int, [C.c_void_p] * 6 + [C.c_int] * 8 + [C.c_float] * 4 + [C.c_void_p]),
    "evmi_conv1d_wgrad_tm_bf16_ws_elems": (C.c_longlong, [C.c_longlong] + [C.c_int] * 4),
    "evmi_conv1d_wgrad_tm_bf16": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong, C.c_longlong] + [C.c_int] * 6 + [C.c_void_p]),
    "evmi_tm_colsum_bf16_ws_elems": (C.c_longlong, [C.c_longlong, C.c_int]),
    "evmi_tm_colsum_bf16": (C.c_int, [C.c_void_p] * 3 + [C.c_longlong, C.c_longlong, C.c_int, C.c_int, C.c_void_p]),
    "evmi_cbt_f32_to_tm_bf16": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int] * 5 + [C.c_float, C.c_float, C.c_void_p]),
    "evmi_tm_bf16_to_cbt_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_int] * 5 + [C.c_float, C.c_void_p]),
    "evmi_tm_lrelu_bf16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_void_p]),
    "evmi_optimizer_step_lrdev_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_longlong, C.c_void_p] + [C.c_float] * 4 + [C.c_void_p, C.c_float, C.c_void_p]),
    "evmi_store_f32": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "evmi_store_u64": (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_void_p]),
    "evmi_optimizer_step_f32": (C.c_int, [C.c_int] + [C.c_void_p] * 4 + [C.c_longlong] + [C.c_float] * 5 + [C.c_int, C.c_void_p, C.c_float, C.c_void_p]),
    "evmi_comm_unique_id": (C.c_int, [C.c_void_p]),
    "evmi_comm_init_rank": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_void_p, C.c_int]),
    "evmi_comm_destroy": (C.c_int, [C.c_void_p]),
    "evmi_allreduce_bucket": (C.c_int, [C.c_void_p, C.c_void_p, C.c_longlong, C.c_float, C.c_void_p]),
    "evmi_loudness_lkfs_f32": (C.c_int, [C.c_void_p] * 5 + [C.c_int] * 4 + [C.c_void_p]),
    "evmi_loudness_scratch_elems": (C.c_longlong, [C.c_int] * 4),
    "evmi_pitch_acf_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_float] * 3 + [C.c_void_p]),
    "evmi_pitch_world_ws_elems": (C.c_longlong, [C.c_int] * 5 + [C.c_float] * 3),
    "evmi_pitch_world_f64": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_int] * 5 + [C.c_float] * 4 + [C.c_void_p]),
    "evmi_pitch_world_decimator": (C.c_int, [C.c_int, C.c_void_p, C.c_void_p]),
    "evmi_peak_normalize_f32": (C.c_int, [C.c_void_p] * 3 + [C.c_int, C.c_int, C.c_float, C.c_void_p]),
    "evmi_transpose_bct_cbt_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "evmi_counter_add_i32": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "evmi_spectral_norm_grad_f32": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p]),
    "evmi_ratio_accumulate_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "evmi_adamw_f32": (C.c_int, [C.c_void_p] * 4 + [C.c_longlong] + [C.c_float] * 5 + [C.c_int, C.c_void_p]),
    "evmi_generator_create": (C.c_int, [C.POINTER(GeneratorConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "evmi_generator_destroy": (None, [C.c_void_p]),
    "evmi_generator_set_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "evmi_generator_num_weights": (C.c_int, [C.c_void_p]),
    "evmi_generator_weight_info": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int64)]),
    "evmi_generator_finalize": (C.c_int, [C.c_void_p]),
    "evmi_generator_hop": (C.c_int, [C.c_void_p]),
    "evmi_generator_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "evmi_generator_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "evmi_generator_forward_profiled": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(LaunchRecord), C.c_int, C.POINTER(C.c_int)],
    ),
    "evmi_generator_macs_per_sample": (C.c_double, [C.c_void_p]),
}

_lib = None


def lib_path() -> Path:
    override = os.environ.get("EVMI_LIB")
    return Path(override) if override else _HERE / LIB_NAME


def load() -> C.CDLL:
    """Load the library once and bind every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not path.exists():
        raise ImportError(
            f"{path} not found: build it with `make` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "everyvoice_amd has no CPU fallback."
        )
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.evmi_abi_version() != 2:
        raise ImportError(f"{path}: ABI version {lib.evmi_abi_version()} != 2")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != EVMI_OK:
        msg = load().evmi_last_error()
        raise EvmiError(f"{what or 'libevmi_hip'} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


_raw_stream = None


def current_stream_ptr(device=None) -> int:
    """hipStream_t of torch's current stream on `device` (a torch.device, an index or None = the current device).  Every
    operator wrapper calls this once per launch: torch's raw accessor (no Stream object) costs a fraction of a microsecond
    where ``torch.cuda.current_stream(device).cuda_stream`` costs two to three."""
    global _raw_stream
    import torch

    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", False)
    if not _raw_stream:
        return torch.cuda.current_stream(device).cuda_stream
    idx = device if isinstance(device, int) else (None if device is None else device.index)
    if idx is None:
        idx = torch.cuda.current_device()
    return _raw_stream(idx)
