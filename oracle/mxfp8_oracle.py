"""CPU restatement of the MX-fp8 linear of the 8-bit path (BASELINE config 5).  TEST INFRASTRUCTURE ONLY: imported
by tests/, never by the product path.

What the reference does (mlx_audio/tts/utils.py:241-260): `nn.quantize(model, group_size, bits, class_predicate)` swaps every
nn.Linear / nn.Embedding whose `{path}.scales` is in the checkpoint for its quantised twin, and the forward multiplies through
mx.quantized_matmul with w = scale * q + bias per group.  MLX is not in the reference tree, its packing and kernels are upstream
knowledge: PARITY UNPINNED.  BASELINE.json config 5 maps that layer set onto the MI355X block-scaled fp8 matrix
instruction; this file restates the arithmetic of THAT mapping so the HIP kernels can be checked bit-for-bit on their operands:

  OCP microscaling (MX) e4m3: a block of consecutive inputs of one row shares a power-of-two scale 2^e (E8M0 byte e + 127), the
  elements are OCP float8 e4m3fn (max 448), round to nearest even.
  e = floor(log2(amax)) - 8; if amax * 2^-e > 448: e += 1; clamp to [-127, 127]; amax == 0 -> e = 0.
  weights: one e per `group` (64) inputs, computed once; activations: one e per 32 inputs, computed per call from the bf16 row.
  product: sum over k of (qa * 2^ea) * (qw * 2^ew), accumulated in fp32 by the matrix core (any order), + bias, exact-erf GELU.
"""
from __future__ import annotations

import numpy as np
import torch


def block_exponents(x: np.ndarray, block: int) -> np.ndarray:
    """x [..., K] float32 -> e [..., K // block] int32 by the rule above."""
    x = np.asarray(x, np.float32)
    K = x.shape[-1]
    assert K % block == 0
    amax = np.abs(x.reshape(*x.shape[:-1], K // block, block)).max(-1)
    bits = amax.view(np.uint32)
    e = ((bits >> 23) & 0xFF).astype(np.int32) - 127 - 8
    over = np.ldexp(amax.astype(np.float64), -e) > 448.0
    e = np.clip(e + over.astype(np.int32), -127, 127)
    return np.where(amax > 0, e, 0).astype(np.int32)


def e4m3_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest OCP e4m3fn value (ties to even), returned as float32.  torch's conversion is the independent implementation."""
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(torch.float8_e4m3fn).to(torch.float32).numpy()


def e4m3_bits(x: np.ndarray) -> np.ndarray:
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
    return t.to(torch.float8_e4m3fn).view(torch.uint8).numpy()


def mx_quantize(x: np.ndarray, block: int):
    """x [..., K] -> (q [..., K] float32 e4m3 values, e [..., K // block] int32) with x ~ q * 2^e."""
    x = np.asarray(x, np.float32)
    e = block_exponents(x, block)
    scale = np.repeat(np.ldexp(np.float32(1.0), -e), block, axis=-1).astype(np.float32)
    return e4m3_round(x * scale), e


def mx_dequantize(q: np.ndarray, e: np.ndarray, block: int) -> np.ndarray:
    return (q.astype(np.float64) * np.repeat(np.ldexp(1.0, e.astype(np.int64)), block, axis=-1)).astype(np.float64)


def bf16_round(x: np.ndarray) -> np.ndarray:
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(torch.bfloat16).to(torch.float32).numpy()


def linear_mxfp8(x: np.ndarray, w: np.ndarray, bias=None, group: int = 64, act: str = "none") -> np.ndarray:
    """x [M, K] (bf16-representable), w [N, K] float32 -> float64 [M, N]: what the matrix core sums, before the bf16 store."""
    qa, ea = mx_quantize(x, 32)
    qw, ew = mx_quantize(w, group)
    y = mx_dequantize(qa, ea, 32) @ mx_dequantize(qw, ew, group).T
    if bias is not None:
        y = y + np.asarray(bias, np.float64)[None]
    if act == "gelu":
        y = torch.from_numpy(y)
        y = (0.5 * y * (1.0 + torch.erf(y / np.sqrt(2.0)))).numpy()
    return y


def frag_index(row: int, k: int, KS: int):
    """Byte offset of element (row, k) in the fragment-order e4m3 pack and of its block's scale byte (csrc/kk_mxfp8.hip header)."""
    blk, r, ks, half, h, byte = row >> 5, row & 31, k >> 6, (k >> 5) & 1, (k >> 4) & 1, k & 15
    frag = blk * KS + ks
    return ((frag * 2 + half) * 64 + r + 32 * h) * 16 + byte, frag * 64 + r + 32 * half


def unpack_frag(q: np.ndarray, s: np.ndarray, rows: int, K: int):
    """fragment-order bytes -> (e4m3 bit patterns [rows, K] uint8, exponents [rows, K // 32] int32)."""
    KS = K // 64
    rr, kk = np.meshgrid(np.arange(rows), np.arange(K), indexing="ij")
    blk, r, ks, half, h, byte = rr >> 5, rr & 31, kk >> 6, (kk >> 5) & 1, (kk >> 4) & 1, kk & 15
    lane = r + 32 * h
    frag = blk * KS + ks
    bits = q[((frag * 2 + half) * 64 + lane) * 16 + byte]
    kb = np.arange(K // 32)
    rr2, kb2 = np.meshgrid(np.arange(rows), kb, indexing="ij")
    frag2 = (rr2 >> 5) * KS + (kb2 >> 1)
    lane2 = (rr2 & 31) + 32 * (kb2 & 1)
    e = s[frag2 * 64 + lane2].astype(np.int32) - 127
    return bits.astype(np.uint8), e
