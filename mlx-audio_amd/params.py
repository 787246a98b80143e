"""Kokoro-82M configuration, parameter inventory and random-init checkpoints.

The hyper-parameters are the ones the reference's own test pins
(mlx_audio/tts/tests/test_models.py:92-122); the parameter names/shapes are the MLX-side
(post-`sanitize`) module tree built at kokoro.py:83-113, modules.py:22-39,289-342,381-387 and
istftnet.py:350-375,709-767,853-861,917-945.  Conv weights are [C_out, K, C_in/groups]
(MLX layout), linear weights [out, in].

`synth_checkpoint` makes a seeded random-init checkpoint of exactly this architecture.  There
is no network here, so no real checkpoint exists; benchmarks and parity tests run on these.
"""

from __future__ import annotations

import json
import os
from typing import Dict, List, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def load_vocab() -> Dict[str, int]:
    """Phoneme -> id table (data from the reference's vendored tokenizer,
    mlx_audio_swift/tts/Swift-TTS/Kokoro/TextProcessing/Tokenizer.swift:28-44; the same
    table ships in the checkpoint's config.json as `vocab`, kokoro.py:62,87)."""
    with open(os.path.join(_HERE, "data", "kokoro_vocab.json"), encoding="utf-8") as f:
        return json.load(f)


def kokoro_config(with_vocab: bool = True) -> dict:
    cfg = {
        "istftnet": {
            "upsample_kernel_sizes": [20, 12],
            "upsample_rates": [10, 6],
            "gen_istft_hop_size": 5,
            "gen_istft_n_fft": 20,
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]],
            "resblock_kernel_sizes": [3, 7, 11],
            "upsample_initial_channel": 512,
        },
        "dim_in": 64,
        "dropout": 0.2,
        "hidden_dim": 512,
        "max_conv_dim": 512,
        "max_dur": 50,
        "multispeaker": True,
        "n_layer": 3,
        "n_mels": 80,
        "n_token": 178,
        "style_dim": 128,
        "text_encoder_kernel_size": 5,
        "plbert": {
            "hidden_size": 768,
            "num_attention_heads": 12,
            "intermediate_size": 2048,
            "max_position_embeddings": 512,
            "num_hidden_layers": 12,
            "dropout": 0.1,
        },
        "vocab": load_vocab() if with_vocab else {},
        "sample_rate": 24000,
        "model_type": "kokoro",
    }
    return cfg


def tiny_config() -> dict:
    """A structurally identical but small configuration for fast CPU/GPU unit tests."""
    cfg = kokoro_config(with_vocab=False)
    cfg["hidden_dim"] = 64
    cfg["style_dim"] = 128  # ref_s split at 128 is hard-coded (kokoro.py:145,165)
    cfg["n_token"] = 178
    cfg["plbert"] = {
        "hidden_size": 128,
        "num_attention_heads": 2,
        "intermediate_size": 256,
        "max_position_embeddings": 512,
        "num_hidden_layers": 2,
        "dropout": 0.1,
    }
    cfg["istftnet"] = dict(cfg["istftnet"], upsample_initial_channel=64)
    cfg["decoder_hidden"] = 128
    return cfg


# ----------------------------------------------------------------------------------------
# parameter inventory
# ----------------------------------------------------------------------------------------


def _lstm(prefix: str, inp: int, hid: int) -> List[Tuple[str, Tuple[int, ...], str]]:
    out = []
    for d in ("forward", "backward"):
        out += [
            (f"{prefix}.Wx_{d}", (4 * hid, inp), "lstm"),
            (f"{prefix}.Wh_{d}", (4 * hid, hid), "lstm"),
            (f"{prefix}.bias_ih_{d}", (4 * hid,), "lstm_b"),
            (f"{prefix}.bias_hh_{d}", (4 * hid,), "lstm_b"),
        ]
    return out


def _convw(prefix: str, o: int, k: int, i: int, bias=None):
    """ConvWeighted parameters (istftnet.py:118-126).  `bias` = size or None."""
    out = [(f"{prefix}.weight_g", (o, 1, 1), "g"), (f"{prefix}.weight_v", (o, k, i), "v")]
    if bias is not None:
        out.append((f"{prefix}.bias", (bias,), "bias"))
    return out


def _linear(prefix: str, o: int, i: int, bias=True):
    out = [(f"{prefix}.weight", (o, i), "mat")]
    if bias:
        out.append((f"{prefix}.bias", (o,), "bias"))
    return out


def _ln(prefix: str, c: int):
    return [(f"{prefix}.weight", (c,), "ln_w"), (f"{prefix}.bias", (c,), "ln_b")]


def _adain_resblk1d(prefix: str, cin: int, cout: int, style: int, upsample: bool):
    out = []
    out += _convw(f"{prefix}.conv1", cout, 3, cin, cout)
    out += _convw(f"{prefix}.conv2", cout, 3, cout, cout)
    out += _linear(f"{prefix}.norm1.fc", 2 * cin, style)
    out += _linear(f"{prefix}.norm2.fc", 2 * cout, style)
    if cin != cout:
        out += _convw(f"{prefix}.conv1x1", cout, 1, cin, None)
    if upsample:
        out += _convw(f"{prefix}.pool", cin, 3, 1, cin)
    return out


def _adain_resblock1(prefix: str, ch: int, k: int, style: int):
    out = []
    for j in range(3):
        out += _convw(f"{prefix}.convs1.{j}", ch, k, ch, ch)
        out += _convw(f"{prefix}.convs2.{j}", ch, k, ch, ch)
        out += _linear(f"{prefix}.adain1.{j}.fc", 2 * ch, style)
        out += _linear(f"{prefix}.adain2.{j}.fc", 2 * ch, style)
        out += [(f"{prefix}.alpha1.{j}", (1, ch, 1), "alpha"), (f"{prefix}.alpha2.{j}", (1, ch, 1), "alpha")]
    return out


def param_inventory(cfg: dict, include_unused: bool = True) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) for every parameter of kokoro.py:Model, MLX-side names."""
    H = cfg["hidden_dim"]
    S = cfg["style_dim"]
    pb = cfg["plbert"]
    hs, inter, E = pb["hidden_size"], pb["intermediate_size"], pb.get("embedding_size", 128)
    ist = cfg["istftnet"]
    DH = cfg.get("decoder_hidden", 1024)  # hard-coded 1024 at istftnet.py:917-932
    inv: List[Tuple[str, Tuple[int, ...], str]] = []
    # --- CustomAlbert (modules.py:438-649)
    inv += [
        ("bert.embeddings.word_embeddings.weight", (cfg["n_token"], E), "emb"),
        ("bert.embeddings.position_embeddings.weight", (pb["max_position_embeddings"], E), "emb"),
        ("bert.embeddings.token_type_embeddings.weight", (2, E), "emb"),
    ]
    inv += _ln("bert.embeddings.LayerNorm", E)
    inv += _linear("bert.encoder.embedding_hidden_mapping_in", hs, E)
    lp = "bert.encoder.albert_layer_groups.0.albert_layers.0"
    for n in ("query", "key", "value", "dense"):
        inv += _linear(f"{lp}.attention.{n}", hs, hs)
    inv += _ln(f"{lp}.attention.LayerNorm", hs)
    inv += _ln(f"{lp}.full_layer_layer_norm", hs)
    inv += _linear(f"{lp}.ffn", inter, hs)
    inv += _linear(f"{lp}.ffn_output", hs, inter)
    if include_unused:
        inv += _linear("bert.pooler", hs, hs)  # computed but unused (kokoro.py:142)
    inv += _linear("bert_encoder", H, hs)
    # --- ProsodyPredictor (modules.py:288-342,380-387)
    for i in range(cfg["n_layer"]):
        inv += _lstm(f"predictor.text_encoder.lstms.{2 * i}", H + S, H // 2)
        inv += _linear(f"predictor.text_encoder.lstms.{2 * i + 1}.fc", 2 * H, S)
    inv += _lstm("predictor.lstm", H + S, H // 2)
    inv += _linear("predictor.duration_proj.linear_layer", cfg["max_dur"], H)
    inv += _lstm("predictor.shared", H + S, H // 2)
    for name in ("F0", "N"):
        inv += _adain_resblk1d(f"predictor.{name}.0", H, H, S, False)
        inv += _adain_resblk1d(f"predictor.{name}.1", H, H // 2, S, True)
        inv += _adain_resblk1d(f"predictor.{name}.2", H // 2, H // 2, S, False)
        inv += [(f"predictor.{name}_proj.weight", (1, 1, H // 2), "conv"), (f"predictor.{name}_proj.bias", (1,), "bias")]
    # --- TextEncoder (modules.py:21-39)
    inv += [("text_encoder.embedding.weight", (cfg["n_token"], H), "emb")]
    k = cfg["text_encoder_kernel_size"]
    for i in range(cfg["n_layer"]):
        inv += _convw(f"text_encoder.cnn.{i}.0", H, k, H, H)
        inv += _ln(f"text_encoder.cnn.{i}.1", H)
    inv += _lstm("text_encoder.lstm", H, H // 2)
    # --- Decoder (istftnet.py:902-945)
    inv += _adain_resblk1d("decoder.encode", H + 2, DH, S, False)
    for i in range(3):
        inv += _adain_resblk1d(f"decoder.decode.{i}", DH + 2 + 64, DH, S, False)
    inv += _adain_resblk1d("decoder.decode.3", DH + 2 + 64, H, S, True)
    inv += _convw("decoder.F0_conv", 1, 3, 1, 1)
    inv += _convw("decoder.N_conv", 1, 3, 1, 1)
    inv += _convw("decoder.asr_res.0", 64, 1, H, 64)
    # --- Generator (istftnet.py:696-767)
    g = "decoder.generator"
    C0 = ist["upsample_initial_channel"]
    nfft = ist["gen_istft_n_fft"]
    inv += _linear(f"{g}.m_source.l_linear", 1, 9)
    rates, ks = ist["upsample_rates"], ist["upsample_kernel_sizes"]
    nk = len(ist["resblock_kernel_sizes"])
    for i, (u, kk) in enumerate(zip(rates, ks)):
        cin, cout = C0 // (2**i), C0 // (2 ** (i + 1))
        # ConvWeighted(out//.., in.., encode=True): weight_v [in_ch_of_convT, K, out_ch], bias [out_ch]
        inv += [
            (f"{g}.ups.{i}.weight_g", (cin, 1, 1), "g"),
            (f"{g}.ups.{i}.weight_v", (cin, kk, cout), "v"),
            (f"{g}.ups.{i}.bias", (cout,), "bias"),
        ]
        if i + 1 < len(rates):
            sf0 = int(np.prod(rates[i + 1 :]))
            inv += [(f"{g}.noise_convs.{i}.weight", (cout, sf0 * 2, nfft + 2), "conv"), (f"{g}.noise_convs.{i}.bias", (cout,), "bias")]
            inv += _adain_resblock1(f"{g}.noise_res.{i}", cout, 7, S)
        else:
            inv += [(f"{g}.noise_convs.{i}.weight", (cout, 1, nfft + 2), "conv"), (f"{g}.noise_convs.{i}.bias", (cout,), "bias")]
            inv += _adain_resblock1(f"{g}.noise_res.{i}", cout, 11, S)
        for j, kr in enumerate(ist["resblock_kernel_sizes"]):
            inv += _adain_resblock1(f"{g}.resblocks.{i * nk + j}", cout, kr, S)
    inv += _convw(f"{g}.conv_post", nfft + 2, 7, C0 // (2 ** len(rates)), nfft + 2)
    return inv


def param_count(cfg: dict) -> int:
    return int(sum(int(np.prod(s)) for _, s, _ in param_inventory(cfg)))


def synth_checkpoint(cfg: dict, seed: int = 0, f0_mean: float = 120.0, f0_std: float = 80.0) -> Dict[str, np.ndarray]:
    """Seeded random-init checkpoint (float32 numpy, MLX-side layout).

    Scales are chosen so that every stage carries O(1) signal (fan-in scaled matrices, random
    biases, weight_g != ||v||, alpha != 1) instead of the degenerate all-zero/all-one defaults:
    a layout or indexing bug anywhere then shows up in the waveform.  The F0 projection is
    biased to ~N(f0_mean, f0_std) Hz so utterances contain voiced (f0 > 10, istftnet.py:716)
    and unvoiced frames; the duration head is biased to ~5 frames per token.
    """
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for name, shape, kind in param_inventory(cfg):
        if kind == "mat":
            a = rng.standard_normal(shape) / np.sqrt(shape[1])
        elif kind in ("v", "conv"):
            fan = shape[1] * shape[2]
            a = rng.standard_normal(shape) / np.sqrt(fan)
        elif kind == "g":
            a = None  # filled from v below
        elif kind == "emb":
            a = rng.standard_normal(shape) * 0.5
        elif kind == "lstm":
            s = 1.0 / np.sqrt(shape[0] // 4)
            a = rng.uniform(-s, s, shape)
        elif kind == "lstm_b":
            a = rng.uniform(-0.06, 0.06, shape)
        elif kind == "bias":
            a = rng.standard_normal(shape) * 0.1
        elif kind == "ln_w":
            a = 1.0 + 0.1 * rng.standard_normal(shape)
        elif kind == "ln_b":
            a = 0.1 * rng.standard_normal(shape)
        elif kind == "alpha":
            a = rng.uniform(0.5, 1.5, shape)
        else:
            raise ValueError(kind)
        if a is not None:
            w[name] = a.astype(np.float32)
    # weight_g: magnitude relative to ||v|| per leading index (so that w = g*v/||v|| has fan-in scaled rows)
    for name, shape, kind in param_inventory(cfg):
        if kind == "g":
            v = w[name[: -len("weight_g")] + "weight_v"]
            nrm = np.sqrt((v.astype(np.float64) ** 2).sum(axis=(1, 2), keepdims=True))
            w[name] = (nrm * (1.0 + 0.1 * rng.standard_normal(shape))).astype(np.float32)
    # AdaIN / AdaLN style projections: keep gamma/beta moderate
    for name in list(w):
        if name.endswith(".fc.weight"):
            w[name] = (w[name] * 0.5).astype(np.float32)
    # heads: F0/N in Hz-like range, duration ~5 frames/token
    for nm, mean, std in (("F0", f0_mean, f0_std), ("N", 0.0, 1.0)):
        w[f"predictor.{nm}_proj.weight"] = (w[f"predictor.{nm}_proj.weight"] * std).astype(np.float32)
        w[f"predictor.{nm}_proj.bias"] = np.full((1,), mean, np.float32)
    w["predictor.duration_proj.linear_layer.bias"] = (
        -2.2 + 0.2 * rng.standard_normal(w["predictor.duration_proj.linear_layer.bias"].shape)
    ).astype(np.float32)
    # conv_post drives exp(): keep log-magnitudes small
    w["decoder.generator.conv_post.weight_g"] = (w["decoder.generator.conv_post.weight_g"] * 0.5).astype(np.float32)
    return w


def to_torch_layout(w: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Inverse of the reference's `sanitize` (kokoro.py:172-252, istftnet.py:965-979): produce the
    PyTorch-side names/layouts a non-quantised Hugging Face checkpoint carries.  Used to test
    that the loader accepts both layouts."""
    out = {}
    lstm_map = {
        "Wx_forward": "weight_ih_l0", "Wh_forward": "weight_hh_l0",
        "bias_ih_forward": "bias_ih_l0", "bias_hh_forward": "bias_hh_l0",
        "Wx_backward": "weight_ih_l0_reverse", "Wh_backward": "weight_hh_l0_reverse",
        "bias_ih_backward": "bias_ih_l0_reverse", "bias_hh_backward": "bias_hh_l0_reverse",
    }
    for k, v in w.items():
        base, _, leaf = k.rpartition(".")
        if leaf in lstm_map:
            out[f"{base}.{lstm_map[leaf]}"] = v
        elif leaf == "weight_v" or (("noise_convs" in k or "F0_proj" in k or "N_proj" in k) and leaf == "weight"):
            out[k] = np.ascontiguousarray(v.transpose(0, 2, 1))
        elif k.startswith("text_encoder.cnn.") and base.endswith(".1"):
            out[f"{base}.{'gamma' if leaf == 'weight' else 'beta'}"] = v
        else:
            out[k] = v
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Mimi codec (decode path): configuration mirrors mimi_202407 (mlx_audio/codec/models/mimi/mimi.py:41-101)
# ---------------------------------------------------------------------------------------------------------------------
def mimi_config(num_codebooks: int = 32) -> dict:
    return dict(dim=512, nq=num_codebooks, bins=2048, qdim=256, num_heads=8, num_layers=8, dim_feedforward=2048, nfilters=64,
                ratios=[8, 6, 5, 4], ksize=7, residual_ksize=3, last_ksize=3, upsample_stride=2, rope_base=10000, compress=2,
                sample_rate=24000, frame_rate=12.5)


def mimi_tiny_config() -> dict:
    """Same topology, small widths (head_dim stays 64): unit tests."""
    return dict(dim=128, nq=4, bins=64, qdim=32, num_heads=2, num_layers=2, dim_feedforward=256, nfilters=8,
                ratios=[8, 6, 5, 4], ksize=7, residual_ksize=3, last_ksize=3, upsample_stride=2, rope_base=10000, compress=2,
                sample_rate=24000, frame_rate=12.5)


def mimi_param_inventory(cfg: dict, encode: bool = False) -> dict:
    """MLX-side names and shapes of everything Mimi.decode (and, with encode=True, Mimi.encode) reads (after load_pytorch_weights'
    remap, mimi.py:184-249)."""
    D, Q = cfg["dim"], cfg["qdim"]
    inv = {}
    if encode:
        nf = cfg["nfilters"]
        inv["encoder.init_conv1d.conv.conv.weight"] = (nf, cfg["ksize"], 1)
        inv["encoder.init_conv1d.conv.conv.bias"] = (nf,)
        mult = 1
        for l, r in enumerate(reversed(cfg["ratios"])):
            dim = mult * nf
            hid = dim // cfg["compress"]
            p = f"encoder.layers.{l}"
            inv[f"{p}.residuals.0.block.0.conv.conv.weight"] = (hid, cfg["residual_ksize"], dim)
            inv[f"{p}.residuals.0.block.0.conv.conv.bias"] = (hid,)
            inv[f"{p}.residuals.0.block.1.conv.conv.weight"] = (dim, 1, hid)
            inv[f"{p}.residuals.0.block.1.conv.conv.bias"] = (dim,)
            inv[f"{p}.downsample.conv.conv.weight"] = (2 * dim, 2 * r, dim)
            inv[f"{p}.downsample.conv.conv.bias"] = (2 * dim,)
            mult *= 2
        inv["encoder.final_conv1d.conv.conv.weight"] = (D, cfg["last_ksize"], mult * nf)
        inv["encoder.final_conv1d.conv.conv.bias"] = (D,)
        inv["downsample.conv.conv.conv.weight"] = (D, 2 * cfg["upsample_stride"], D)
        for which in ("rvq_first", "rvq_rest"):
            if which == "rvq_first" or cfg["nq"] > 1:
                inv[f"quantizer.{which}.input_proj.weight"] = (Q, 1, D)
        for i in range(cfg["num_layers"]):
            p = f"encoder_transformer.transformer.layers.{i}"
            for nm in ("norm1", "norm2"):
                inv[f"{p}.{nm}.weight"] = (D,)
                inv[f"{p}.{nm}.bias"] = (D,)
            inv[f"{p}.self_attn.in_proj.weight"] = (3 * D, D)
            inv[f"{p}.self_attn.out_proj.weight"] = (D, D)
            inv[f"{p}.layer_scale_1.scale"] = (D,)
            inv[f"{p}.layer_scale_2.scale"] = (D,)
            inv[f"{p}.gating.linear1.weight"] = (cfg["dim_feedforward"], D)
            inv[f"{p}.gating.linear2.weight"] = (D, cfg["dim_feedforward"])
    for which, n in (("rvq_first", 1), ("rvq_rest", cfg["nq"] - 1)):
        for i in range(n):
            inv[f"quantizer.{which}.vq.layers.{i}.codebook.embedding_sum"] = (cfg["bins"], Q)
            inv[f"quantizer.{which}.vq.layers.{i}.codebook.cluster_usage"] = (cfg["bins"],)
        if n > 0:
            inv[f"quantizer.{which}.output_proj.weight"] = (D, 1, Q)
    inv["upsample.convtr.convtr.convtr.weight"] = (1, 2 * cfg["upsample_stride"], D)
    for i in range(cfg["num_layers"]):
        p = f"decoder_transformer.transformer.layers.{i}"
        for nm in ("norm1", "norm2"):
            inv[f"{p}.{nm}.weight"] = (D,)
            inv[f"{p}.{nm}.bias"] = (D,)
        inv[f"{p}.self_attn.in_proj.weight"] = (3 * D, D)
        inv[f"{p}.self_attn.out_proj.weight"] = (D, D)
        inv[f"{p}.layer_scale_1.scale"] = (D,)
        inv[f"{p}.layer_scale_2.scale"] = (D,)
        inv[f"{p}.gating.linear1.weight"] = (cfg["dim_feedforward"], D)
        inv[f"{p}.gating.linear2.weight"] = (D, cfg["dim_feedforward"])
    mult = 1 << len(cfg["ratios"])
    nf = cfg["nfilters"]
    inv["decoder.init_conv1d.conv.conv.weight"] = (mult * nf, cfg["ksize"], D)
    inv["decoder.init_conv1d.conv.conv.bias"] = (mult * nf,)
    for l, r in enumerate(cfg["ratios"]):
        cin, cout = mult * nf, mult * nf // 2
        p = f"decoder.layers.{l}"
        inv[f"{p}.upsample.convtr.convtr.weight"] = (cout, 2 * r, cin)
        inv[f"{p}.upsample.convtr.convtr.bias"] = (cout,)
        hid = cout // cfg["compress"]
        inv[f"{p}.residuals.0.block.0.conv.conv.weight"] = (hid, cfg["residual_ksize"], cout)
        inv[f"{p}.residuals.0.block.0.conv.conv.bias"] = (hid,)
        inv[f"{p}.residuals.0.block.1.conv.conv.weight"] = (cout, 1, hid)
        inv[f"{p}.residuals.0.block.1.conv.conv.bias"] = (cout,)
        mult //= 2
    inv["decoder.final_conv1d.conv.conv.weight"] = (1, cfg["last_ksize"], nf)
    inv["decoder.final_conv1d.conv.conv.bias"] = (1,)
    return inv


def mimi_synth_checkpoint(cfg: dict, seed: int = 0, encode: bool = False) -> dict:
    """Seeded random-init checkpoint (fan-in scaled so activations stay O(1); no real weights exist offline).  The decode-side
    tensors do not depend on `encode` (they are drawn first)."""
    rng = np.random.default_rng(seed)
    w = {}
    inv = mimi_param_inventory(cfg)
    if encode:
        inv.update({k: v for k, v in mimi_param_inventory(cfg, encode=True).items() if k not in inv})
    for name, shape in inv.items():
        if name.endswith("embedding_sum"):
            w[name] = (0.25 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("cluster_usage"):
            u = rng.uniform(0.5, 2.0, shape).astype(np.float32)
            u[:2] = 0.0  # two never-used entries: the 1e-5 floor of quantization.py:25 is exercised ...
            w[name] = u
        elif name.endswith("layer_scale_1.scale") or name.endswith("layer_scale_2.scale"):
            w[name] = rng.uniform(0.05, 0.4, shape).astype(np.float32)
        elif name.endswith("norm1.weight") or name.endswith("norm2.weight"):
            w[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif name.endswith(".bias"):
            w[name] = (0.1 * rng.standard_normal(shape)).astype(np.float32)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            if name.endswith("convtr.weight") and len(shape) == 3 and shape[0] > 1:
                fan_in = shape[2] * 2  # two taps of a k = 2*stride transposed conv reach an output sample
            if name.startswith("upsample."):
                fan_in = 2
            gain = 0.5 if ".block.1." in name else 1.0
            w[name] = (rng.standard_normal(shape) * (gain / np.sqrt(fan_in))).astype(np.float32)
    for name in list(w):
        if name.endswith("cluster_usage"):
            w[name.replace("cluster_usage", "embedding_sum")][:2] = 0.0  # ... without producing 1e5-scale rows
    return w


# ---------------------------------------------------------------------------------------------------------------------
# CSM-1B frame generator (sesame.py:225-273 llama flavours, :276-415 SesameModel)
# ---------------------------------------------------------------------------------------------------------------------
def csm_config() -> dict:
    rope = dict(rope_theta=500000.0, rope_factor=32.0, rms_eps=1e-5)
    return dict(text_vocab_size=128256, audio_vocab_size=2051, audio_num_codebooks=32, max_seq_len=2048,
                backbone=dict(num_layers=16, num_heads=32, num_kv_heads=8, head_dim=64, hidden=2048, intermediate=8192, **rope),
                decoder=dict(num_layers=4, num_heads=8, num_kv_heads=2, head_dim=128, hidden=1024, intermediate=8192, **rope))


def csm_tiny_config() -> dict:
    rope = dict(rope_theta=500000.0, rope_factor=32.0, rms_eps=1e-5)
    return dict(text_vocab_size=300, audio_vocab_size=67, audio_num_codebooks=4, max_seq_len=64,
                backbone=dict(num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64, hidden=256, intermediate=512, **rope),
                decoder=dict(num_layers=2, num_heads=2, num_kv_heads=1, head_dim=128, hidden=256, intermediate=384, **rope))


def csm_param_inventory(cfg: dict) -> dict:
    inv = {}
    D, Dd = cfg["backbone"]["hidden"], cfg["decoder"]["hidden"]
    for name, a in (("backbone", cfg["backbone"]), ("decoder", cfg["decoder"])):
        H, KV, hd, Dm, I = a["num_heads"], a["num_kv_heads"], a["head_dim"], a["hidden"], a["intermediate"]
        for i in range(a["num_layers"]):
            p = f"{name}.layers.{i}"
            inv[f"{p}.self_attn.q_proj.weight"] = (H * hd, Dm)
            inv[f"{p}.self_attn.k_proj.weight"] = (KV * hd, Dm)
            inv[f"{p}.self_attn.v_proj.weight"] = (KV * hd, Dm)
            inv[f"{p}.self_attn.o_proj.weight"] = (Dm, H * hd)
            inv[f"{p}.mlp.gate_proj.weight"] = (I, Dm)
            inv[f"{p}.mlp.up_proj.weight"] = (I, Dm)
            inv[f"{p}.mlp.down_proj.weight"] = (Dm, I)
            inv[f"{p}.input_layernorm.weight"] = (Dm,)
            inv[f"{p}.post_attention_layernorm.weight"] = (Dm,)
        inv[f"{name}.norm.weight"] = (Dm,)
    inv["text_embeddings.weight"] = (cfg["text_vocab_size"], D)
    inv["audio_embeddings.weight"] = (cfg["audio_vocab_size"] * cfg["audio_num_codebooks"], D)
    inv["projection.weight"] = (Dd, D)
    inv["codebook0_head.weight"] = (cfg["audio_vocab_size"], D)
    inv["audio_head"] = (cfg["audio_num_codebooks"] - 1, Dd, cfg["audio_vocab_size"])
    return inv


def csm_synth_checkpoint(cfg: dict, seed: int = 0) -> dict:
    rng = np.random.default_rng(seed)
    w = {}
    for name, shape in csm_param_inventory(cfg).items():
        if name.endswith("layernorm.weight") or name.endswith("norm.weight"):
            w[name] = rng.uniform(0.5, 1.5, shape).astype(np.float32)
        elif name.endswith("embeddings.weight"):
            w[name] = rng.standard_normal(shape, dtype=np.float32)
        elif name == "audio_head":
            w[name] = (rng.standard_normal(shape, dtype=np.float32) * (3.0 / np.sqrt(shape[1]))).astype(np.float32)
        else:
            gain = 3.0 if name.startswith("codebook0_head") else (0.5 if ("o_proj" in name or "down_proj" in name) else 1.0)
            w[name] = (rng.standard_normal(shape, dtype=np.float32) * (gain / np.sqrt(shape[-1]))).astype(np.float32)
    return w
