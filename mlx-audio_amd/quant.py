"""MLX affine group quantisation (the format `load_model` meets in `mlx-community/*-8bit` checkpoints, mlx_audio/tts/utils.py:241-260):
`nn.quantize(model, group_size, bits, class_predicate)` turns a Linear / Embedding weight [O, I] into
    weight  uint32 [O, I * bits / 32]   (32 / bits values per word, element j of a word in bits [j*bits, (j+1)*bits), least significant first)
    scales  [O, I / group_size],  biases [O, I / group_size]              w ~= scales * q + biases  per group of `group_size` inputs
This module dequantises such triplets at LOAD time; the arithmetic then runs on the ordinary kernels (bf16 MFMA in bf16 mode) with exactly
the weights MLX's `quantized_matmul` multiplies by and bf16 activations -- the reference's arithmetic, and the DEFAULT for such a checkpoint.
The packing layout is upstream-MLX knowledge (MLX is not in the reference tree): PARITY UNPINNED.
Opt-in (`load_model(..., quantization_kernel="mxfp8")`): the 8-bit checkpoint's Linear set runs on the block-scaled fp8 matrix instruction,
the engine re-quantising the dequantised matrices to e4m3 with one power-of-two scale per group (kk_set_quantization, csrc/kk_mxfp8.hip;
SURVEY 8 row Q1) -- faster on those layers, but narrower arithmetic than the reference's (4 significant bits for weights AND activations)."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def quantize_affine(w: np.ndarray, group_size: int = 64, bits: int = 8):
    """Restatement of mx.quantize's documented rule (per group: scale = (max - min) / (2^bits - 1), bias = min, q = round((w - bias) /
    scale)); used by the tests to build fixtures, not by the product path."""
    w = np.asarray(w, np.float32)
    O, I = w.shape
    assert I % group_size == 0 and 32 % bits == 0 and (I * bits) % 32 == 0
    g = w.reshape(O, I // group_size, group_size)
    lo, hi = g.min(-1), g.max(-1)
    scales = ((hi - lo) / np.float32(2 ** bits - 1)).astype(np.float32)
    scales = np.where(scales == 0, np.float32(1.0), scales)
    q = np.clip(np.rint((g - lo[..., None]) / scales[..., None]), 0, 2 ** bits - 1).astype(np.uint32).reshape(O, I)
    per = 32 // bits
    words = np.zeros((O, I // per), np.uint32)
    for j in range(per):
        words |= q[:, j::per] << np.uint32(j * bits)
    return words, scales, lo.astype(np.float32)


def dequantize_affine(words: np.ndarray, scales: np.ndarray, biases: np.ndarray, group_size: int = 64, bits: int = 8) -> np.ndarray:
    words = np.asarray(words).astype(np.uint32)
    O = words.shape[0]
    per = 32 // bits
    mask = np.uint32(2 ** bits - 1)
    q = np.empty((O, words.shape[1] * per), np.float32)
    for j in range(per):
        q[:, j::per] = ((words >> np.uint32(j * bits)) & mask).astype(np.float32)
    I = q.shape[1]
    s = np.repeat(np.asarray(scales, np.float32), group_size, axis=1)[:, :I]
    b = np.repeat(np.asarray(biases, np.float32), group_size, axis=1)[:, :I]
    return q * s + b


def dequantize_checkpoint(weights: Dict[str, np.ndarray], group_size: int, bits: int, per_layer: Optional[dict] = None) -> Dict[str, np.ndarray]:
    """Every `{p}.weight` that comes with `{p}.scales` and `{p}.biases` (the reference's predicate, utils.py:243-252) is replaced by its
    dequantised float32 matrix; the scale / bias tensors are dropped; everything else passes through.  `per_layer` is the rest of
    config["quantization"]: a layer path mapped to its own {"group_size", "bits"} (the "custom per layer quantizations" of utils.py:244-246)
    is dequantised with THOSE parameters; a path mapped to False was never quantised and passes through."""
    per_layer = per_layer or {}
    out = {}
    for k, v in weights.items():
        if k.endswith(".scales") or k.endswith(".biases"):
            continue
        p = k[: -len(".weight")] if k.endswith(".weight") else None
        own = per_layer.get(p) if p is not None else None
        if p is not None and own is not False and f"{p}.scales" in weights and f"{p}.biases" in weights:
            g, nb = (int(own.get("group_size", group_size)), int(own.get("bits", bits))) if isinstance(own, dict) else (group_size, bits)
            out[k] = dequantize_affine(np.asarray(v), np.asarray(weights[f"{p}.scales"], np.float32), np.asarray(weights[f"{p}.biases"], np.float32), g, nb)
        else:
            out[k] = v
    return out


def quantised_layer_names(weights: Dict[str, np.ndarray], group_size: int = 64):
    """The reference's class predicate (tts/utils.py:349-369 / :243-252) on a Kokoro checkpoint: nn.Linear / nn.Embedding weights
    (2-D `*.weight`) whose element count is a multiple of 64 and whose input width is a multiple of the group size.  ConvWeighted,
    the hand-rolled LSTM and nn.Conv1d hold raw arrays and stay as they are."""
    names = []
    for k, v in weights.items():
        if not k.endswith(".weight") or np.ndim(v) != 2:  # LSTM matrices are named Wx_* / Wh_* (weight_ih_* on the PyTorch side)
            continue
        a = np.asarray(v)
        if a.size % 64 == 0 and a.shape[1] % group_size == 0:
            names.append(k)
    return names


def quantize_checkpoint(weights: Dict[str, np.ndarray], group_size: int = 64, bits: int = 8) -> Dict[str, np.ndarray]:
    """What `convert(..., quantize=True)` leaves on disk for the layer set above: uint32-packed `weight`, `scales`, `biases`
    (test fixtures and bench.py --quantized; the product path only ever DEquantises)."""
    out = dict(weights)
    for k in quantised_layer_names(weights, group_size):
        words, scales, biases = quantize_affine(np.asarray(weights[k], np.float32), group_size, bits)
        p = k[: -len(".weight")]
        out[k], out[p + ".scales"], out[p + ".biases"] = words, scales, biases
    return out
