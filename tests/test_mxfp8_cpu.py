"""CPU tests of the 8-bit path's host side: the oracle's MX-fp8 rule, and the library's HOST weight packer
(kk_mxfp8_pack_weight: no GPU involved) checked bit for bit against it."""
import ctypes as C

import numpy as np
import pytest
import torch

import mxfp8_oracle as MX


def test_e4m3_rounding_rule_known_values():
    # OCP e4m3fn: max 448, min normal 2^-6, subnormal step 2^-9, 3 mantissa bits, ties to even
    x = np.array([0.0, 1.0, 1.0625, 1.1875, 448.0, 2.0**-6, 2.0**-9, 2.0**-10, 1.5 * 2.0**-9, 17.0, 19.0, -3.3], np.float32)
    want = np.array([0.0, 1.0, 1.0, 1.25, 448.0, 2.0**-6, 2.0**-9, 0.0, 2.0**-8, 16.0, 20.0, -3.25], np.float32)
    np.testing.assert_array_equal(MX.e4m3_round(x), want)


def test_block_exponent_rule():
    # amax in [256, 448] * 2^e keeps e; (448, 512) * 2^e moves to e + 1 so that nothing saturates
    for amax, e in ((1.0, -8), (1.74, -8), (1.75, -8), (1.76, -7), (448.0, 0), (449.0, 1), (0.02, -14), (0.0, 0)):
        x = np.zeros((1, 32), np.float32)
        x[0, 3] = -amax
        assert MX.block_exponents(x, 32)[0, 0] == e, (amax, e)
        q, ee = MX.mx_quantize(x, 32)
        assert np.abs(q).max() <= 448.0
        if amax:
            assert abs(MX.mx_dequantize(q, ee, 32)[0, 3] + amax) <= amax * 2.0**-4


def test_quantisation_error_bound_random():
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((96, 256)) * 0.02).astype(np.float32)
    q, e = MX.mx_quantize(w, 64)
    back = MX.mx_dequantize(q, e, 64)
    # half an ulp of a 3-bit mantissa relative to the element, or half a subnormal step relative to the block scale
    bound = np.maximum(np.abs(w) * 2.0**-4, np.repeat(np.ldexp(1.0, e - 10), 64, axis=-1))
    assert (np.abs(back - w) <= bound * (1 + 1e-6)).all()


def test_library_weight_packer_matches_the_oracle_bit_for_bit():
    from mlx_audio_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(1)
    for (N, K, group) in ((64, 128, 64), (96, 256, 64), (64, 64, 32), (128, 192, 64)):
        w = (rng.standard_normal((N, K)) * rng.choice([0.02, 1.0, 30.0])).astype(np.float32)
        w[0, :64] = 0.0  # an all-zero group
        w[1, 5] = 1e-30  # far below the block's subnormal range
        qb, sb = C.c_size_t(), C.c_size_t()
        _lib.check(lib.kk_mxfp8_bytes(N, K, C.byref(qb), C.byref(sb)), "bytes")
        assert qb.value == N * K and sb.value == N * K // 32
        q = np.zeros(qb.value, np.uint8)
        s = np.zeros(sb.value, np.uint8)
        _lib.check(lib.kk_mxfp8_pack_weight(w.ctypes.data_as(C.c_void_p), N, K, group, q.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)),
                   "pack")
        bits, e = MX.unpack_frag(q, s, N, K)
        qo, eo = MX.mx_quantize(w, group)
        np.testing.assert_array_equal(e, np.repeat(eo, group // 32, axis=-1))
        np.testing.assert_array_equal(bits, MX.e4m3_bits(qo))
    # shapes the kernel cannot take are refused, not silently mis-packed
    assert lib.kk_mxfp8_pack_weight(w.ctypes.data_as(C.c_void_p), 64, 100, 64, q.ctypes.data_as(C.c_void_p), s.ctypes.data_as(C.c_void_p)) != 0
