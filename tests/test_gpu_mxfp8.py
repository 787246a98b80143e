"""GPU parity tests of the 8-bit path (BASELINE config 5): the MX-fp8 linear kernels through the C ABI against
oracle/mxfp8_oracle.py, and the quantised Kokoro model against the fp32 oracle run on the same dequantised weights.

Bars: operand bytes (activation pre-pass) bit-exact; product on exactly representable integer data bit-exact; product on random
data within fp32 summation-order noise + one bf16 rounding of the output; model level statistical (written at the assertions) --
parity with MLX's own quantized_matmul is UNPINNED (MLX is not in the reference tree)."""
import ctypes as C

import numpy as np
import pytest
import torch

import kokoro_oracle as O
import mlx_audio_amd.params as P
import mxfp8_oracle as MX
from _util import err_stats, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mlx_audio_amd import _lib

    return _lib.load()


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run_linear(lib, x, w, bias=None, group=64, act=0, rows_per_item=None, lens=None, ldx=None, ldo=None):
    """x [M, K] float (rounded to bf16 here), w [N, K] float32.  Returns (out [M, N] float32, activation bits, exponents)."""
    from mlx_audio_amd import _lib

    M, K = x.shape
    N = w.shape[0]
    ldx = ldx or K
    ldo = ldo or N
    qb, sb = C.c_size_t(), C.c_size_t()
    _lib.check(lib.kk_mxfp8_bytes(N, K, C.byref(qb), C.byref(sb)), "bytes")
    wq, ws = np.zeros(qb.value, np.uint8), np.zeros(sb.value, np.uint8)
    w = np.ascontiguousarray(w, np.float32)
    _lib.check(lib.kk_mxfp8_pack_weight(w.ctypes.data_as(C.c_void_p), N, K, group, wq.ctypes.data_as(C.c_void_p), ws.ctypes.data_as(C.c_void_p)), "pack")
    _lib.check(lib.kk_mxfp8_bytes(M, K, C.byref(qb), C.byref(sb)), "bytes")
    xd = torch.zeros((M, ldx), dtype=torch.bfloat16, device="cuda")
    xd[:, :K] = torch.as_tensor(x).to(torch.bfloat16)
    aq = torch.zeros(qb.value, dtype=torch.uint8, device="cuda")
    asc = torch.zeros(sb.value, dtype=torch.uint8, device="cuda")
    out = torch.full((M, ldo), 7.0, dtype=torch.bfloat16, device="cuda")
    wqd, wsd = torch.as_tensor(wq).cuda(), torch.as_tensor(ws).cuda()
    bd = torch.as_tensor(np.asarray(bias, np.float32)).cuda() if bias is not None else None
    ld = torch.as_tensor(np.asarray(lens, np.int32)).cuda() if lens is not None else None
    _lib.check(lib.kk_op_linear_mxfp8(_stream(), _p(xd), ldx, M, rows_per_item or M, _p(ld), K, _p(wqd), _p(wsd), N, _p(bd), act, _p(aq), _p(asc),
                                      _p(out), ldo), "kk_op_linear_mxfp8")
    torch.cuda.synchronize()
    bits, e = MX.unpack_frag(aq.cpu().numpy(), asc.cpu().numpy(), M, K)
    return out.float().cpu().numpy(), bits, e


def test_activation_prepass_is_bit_exact(lib):
    rng = np.random.default_rng(0)
    M, K = 200, 256  # M is not a multiple of 32: the tail block's missing rows are zero-filled
    x = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-6, 6, (M, 1)))).astype(np.float32)
    x[3] = 0.0
    x[5, 64:96] = 0.0
    x[7, 10] = 3.0e4  # one outlier owns its block's scale; the other 31 elements fall into the subnormal range
    x = MX.bf16_round(x)
    w = rng.standard_normal((64, K)).astype(np.float32)
    _, bits, e = run_linear(lib, x, w, ldx=K + 8)
    q, eo = MX.mx_quantize(x, 32)
    np.testing.assert_array_equal(e, eo)
    np.testing.assert_array_equal(bits, MX.e4m3_bits(q))


def test_product_on_integer_data_is_exact(lib):
    """Exactly representable operands (small integers: lossless in e4m3 under any block scale they get) and an ASYMMETRIC weight
    matrix: checks the operand lane maps, the k pairing of A and B fragments, the scale plumbing and the C/D layout."""
    rng = np.random.default_rng(1)
    M, K, N = 96, 192, 128
    x = rng.integers(-2, 3, (M, K)).astype(np.float32)
    x *= np.exp2(rng.integers(0, 4, (M, K // 32))).repeat(32, axis=1).astype(np.float32)  # every (row, 32-block) its own exponent
    w = rng.integers(-2, 3, (N, K)).astype(np.float32)
    w *= np.exp2(rng.integers(0, 3, (N, K // 64))).repeat(64, axis=1).astype(np.float32)  # every (row, group) its own exponent
    bias = rng.integers(-3, 4, N).astype(np.float32)
    got, _, _ = run_linear(lib, x, w, bias)
    want = x.astype(np.float64) @ w.astype(np.float64).T + bias[None]
    # integer sums are exact in fp32; both sides then take the same single bf16 rounding
    np.testing.assert_array_equal(got, MX.bf16_round(want.astype(np.float32)))
    # one-hot probes: output (m, n) = x[m, k0] * w[n, k0] for every k0 in one row block -- any lane / k mismatch shows as a wrong cell
    for k0 in (0, 15, 16, 31, 32, 63, 64, 191):
        xo = np.zeros((32, K), np.float32)
        xo[:, k0] = np.arange(1, 33) % 9 + 1  # e4m3 holds the integers up to 16 exactly
        wo = np.zeros((64, K), np.float32)
        wo[:, k0] = np.arange(1, 65) % 7 + 1
        got, _, _ = run_linear(lib, xo, wo)
        np.testing.assert_array_equal(got, np.outer(xo[:, k0], wo[:, k0]).astype(np.float32))


CASES = [("albert_qkv", 3, 130, 768, 2304, 0), ("albert_ffn_gelu", 2, 130, 768, 2048, 2), ("albert_ffn_out", 2, 130, 2048, 768, 0),
         ("map_in", 3, 50, 128, 768, 0), ("tiny_bert_encoder", 2, 40, 128, 64, 0), ("n_192", 1, 70, 64, 192, 2)]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_linear_matches_oracle(lib, case):
    name, B, rows, K, N, act = case
    rng = np.random.default_rng(__import__("zlib").crc32(name.encode()) % 1000)  # (hash() of a str changes from process to process)
    M = B * rows
    x = MX.bf16_round((rng.standard_normal((M, K)) * 1.5).astype(np.float32))
    w = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    lens = rng.integers(1, rows + 1, B)
    lens[0] = rows
    for b in range(B):
        x[b * rows + lens[b] : (b + 1) * rows] = 0.0  # the engine's buffers hold zeros past an utterance's length
    got, _, _ = run_linear(lib, x, w, bias, act=act, rows_per_item=rows, lens=lens, ldo=N + 16)
    want = MX.linear_mxfp8(x, w, bias, 64, "gelu" if act == 2 else "none")
    for b in range(B):
        assert (got[b * rows + lens[b] : (b + 1) * rows, :N] == 0).all()
        want[b * rows + lens[b] : (b + 1) * rows] = 0.0
    assert (got[:, N:] == 7.0).all()  # the pitch padding of the destination is not touched
    d = np.abs(got[:, :N] - want)
    # fp32 accumulation in another order, then ONE bf16 rounding of the stored value: when the two sums straddle a rounding boundary the
    # stored values differ by a whole bf16 ulp, up to 2^-7 of the value
    bound = np.abs(want) * 2.0**-7 + 2e-5 * np.abs(want).max()
    report(f"mxfp8/{name}", **err_stats(got[:, :N], want))
    assert (d <= bound).all(), float((d - bound).max())
    # and the format's own distance from the unquantised product, for the record
    full = x.astype(np.float64) @ w.astype(np.float64).T + bias[None]
    if act == 0:
        for b in range(B):
            full[b * rows + lens[b] : (b + 1) * rows] = 0.0
        report(f"mxfp8/{name}/vs_unquantised", **err_stats(want, full))


class _Fp8Oracle(O.KokoroOracle):
    """The oracle with the six fp8 linears evaluated by the MX-fp8 restatement on bf16-rounded inputs."""

    SET = ("bert.encoder.embedding_hidden_mapping_in", "attention.query", "attention.key", "attention.value", "attention.dense", ".ffn", "bert_encoder")

    def linear(self, x, p):
        if any(p.endswith(s) or p == s for s in self.SET) or p.endswith("ffn_output"):
            w = self.w[p + ".weight"].numpy()
            b = self.w[p + ".bias"].numpy()
            xs = MX.bf16_round(x.reshape(-1, x.shape[-1]).numpy())
            y = MX.linear_mxfp8(xs, w, b, 64)
            return torch.from_numpy(y.astype(np.float32)).reshape(*x.shape[:-1], w.shape[0])
        return super().linear(x, p)


def test_quantised_model_runs_fp8_and_tracks_the_oracle():
    """An 8-bit checkpoint (MLX affine, group 64) -> dequantised at load (quant.py) -> engine in bf16 mode with kk_set_quantization:
    all six linears get an fp8 pack; the text stage stays within the format's distance of the fp32 oracle run on the SAME
    dequantised weights, and as close to the MX-fp8 restatement of the oracle as bf16 activations allow."""
    from mlx_audio_amd import _lib
    from mlx_audio_amd.engine import KokoroEngine
    from mlx_audio_amd.quant import dequantize_checkpoint, quantize_checkpoint

    cfg = P.kokoro_config()
    w = dequantize_checkpoint(quantize_checkpoint(P.synth_checkpoint(cfg, 0), 64, 8), 64, 8)
    rng = np.random.default_rng(70)
    utts = [rng.integers(1, 178, n).tolist() for n in (30, 21)]
    pack = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "af_heart_rows.npz"))["rows"]
    ref_s = pack[rng.integers(0, pack.shape[0], 2)].astype(np.float32)
    eng = KokoroEngine(cfg, w, compute_dtype="bfloat16", quantization={"group_size": 64, "bits": 8})
    assert eng.quantized_layers() == 6
    dev = eng.device
    ids, lens, Tmax = eng.pack_ids(utts)
    durs = np.zeros((2, Tmax), np.int32)
    for b, u in enumerate(utts):
        durs[b, : len(u) + 2] = 3
    Fmax = int(durs.sum(1).max())
    refs = {}
    for kind, cls in (("fp32", O.KokoroOracle), ("mxfp8", _Fp8Oracle)):
        orc = cls(w, cfg)
        refs[kind] = [orc.forward(utts[b], ref_s[b : b + 1], 1.0, forced_dur=durs[b, : len(utts[b]) + 2], sine_noise=None, return_inter=True)[2]
                      for b in range(2)]
    res = {}
    for mode, flag in (("fp8", 0), ("bf16", 8)):
        eng.lib.kk_debug_force_generic(eng._h, flag)
        wav, _, nfr = eng.forward(ids, lens, torch.tensor(ref_s, device=dev), torch.ones(2, device=dev), Fmax, forced_dur=torch.tensor(durs, device=dev),
                                  noise_mode=_lib.NOISE_ZERO)
        torch.cuda.synchronize()
        assert torch.isfinite(wav).all()
        for name in ("bert_dur", "d", "duration"):
            got = eng.debug_fetch(name).cpu().numpy()
            for b in range(2):
                for kind in refs:
                    ref = np.asarray(refs[kind][b][name])
                    ref = ref[0] if ref.ndim == 3 else ref.reshape(-1, 1)
                    e = err_stats(got[b, : ref.shape[0], : ref.shape[1]], ref)
                    report(f"q8/{mode}/{name}/b{b}/vs_{kind}", **e)
                    res[(mode, name, b, kind)] = e["rms_rel"]
    eng.lib.kk_debug_force_generic(eng._h, 0)
    for b in range(2):
        for name in ("bert_dur", "d"):
            # e4m3 keeps 4 significant bits and Albert applies the SAME quantised layer 12 times: measured on MI355X 11-12 % RMS on
            # bert_dur and 22-23 % on d against the unquantised arithmetic -- the MX-fp8 restatement of the oracle sits at the same
            # distance (11 % / 24 %), i.e. this is the format on this random-init checkpoint, not the kernels
            assert res[("fp8", name, b, "fp32")] < (0.2 if name == "bert_dur" else 0.35), (name, b, res[("fp8", name, b, "fp32")])
            # ... and the kernels add nothing to the format's own error: the distance to the MX-fp8 restatement (measured 3.7 % / 7-10 %:
            # bf16 rounding of the activations flips e4m3 rounding decisions, 12 layers deep) stays well under the format's distance
            assert res[("fp8", name, b, "mxfp8")] < 0.5 * res[("fp8", name, b, "fp32")] + 0.01, (name, b, res[("fp8", name, b, "mxfp8")])
            # the DEFAULT for an 8-bit checkpoint (exact semantics: dequantised weights on the bf16 MFMA kernels, bf16 activations) sits at
            # the plain bf16 bar against the dequantised-weights oracle (measured 0.4-1.3 % rms)
            assert res[("bf16", name, b, "fp32")] < 0.015, (name, b, res[("bf16", name, b, "fp32")])


def test_load_model_on_an_8bit_checkpoint_exact_default_and_fp8_opt_in(tmp_path):
    """The reference's load path for `*-8bit` checkpoints (tts/utils.py:241-260): config["quantization"] + uint32 `weight` / `scales` / `biases`
    triplets on disk.  load_model dequantises them (quant.py).  DEFAULT: the dequantised weights run on the mode's ordinary kernels (no layer
    is re-quantised) and the predicted durations equal the fp32-oracle's on the same dequantised weights.  Opt-in quantization_kernel="mxfp8":
    the group size goes to the engine (kk_set_quantization) and all six eligible linears run on the fp8 kernels; durations agree with an
    engine built directly from the dequantised weights with the same setting."""
    import json

    from safetensors.numpy import save_file

    from mlx_audio_amd.engine import KokoroEngine
    from mlx_audio_amd.quant import dequantize_checkpoint, quantize_checkpoint
    from mlx_audio_amd.utils import load_model

    cfg = P.tiny_config()
    cfg["vocab"] = P.load_vocab()
    w = P.synth_checkpoint(cfg, 0)
    wq = quantize_checkpoint(w, 64, 8)
    assert wq["bert_encoder.weight"].dtype == np.uint32 and "bert_encoder.scales" in wq
    d = tmp_path / "kokoro-8bit"
    d.mkdir()
    json.dump(dict(cfg, model_type="kokoro", quantization={"group_size": 64, "bits": 8}), open(d / "config.json", "w"))
    save_file({k: np.ascontiguousarray(v) for k, v in wq.items()}, str(d / "model.safetensors"))
    ps = "hɛlˈoʊ wˈɜɹld"
    ref_s = np.load(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "af_heart_rows.npz"))["rows"][3][None]  # any real style row
    exact = load_model(str(d), compute_dtype="bfloat16")  # default: exact semantics
    assert exact.engine.quantized_layers() == 0
    oe = exact(ps, ref_s, 1.0, return_output=True)
    orc = O.KokoroOracle(dequantize_checkpoint(wq, 64, 8), {k: v for k, v in cfg.items() if k != "vocab"})
    want = orc.text_stage(exact._ids(ps), ref_s, 1.0)
    report("q8/load_model_exact/durations", got=oe.pred_dur.cpu().numpy().tolist(), want=want.tolist())
    assert np.abs(oe.pred_dur.cpu().numpy() - want).max() <= 1 and (oe.pred_dur.cpu().numpy() != want).sum() <= 1  # bf16 next to a .5 boundary
    with pytest.raises(ValueError):
        load_model(str(d), compute_dtype="float32", quantization_kernel="mxfp8")  # the fp8 opt-in is a bf16-mode feature
    with pytest.raises(ValueError):
        load_model(str(d), quantization_kernel="int4")
    model = load_model(str(d), compute_dtype="bfloat16", quantization_kernel="mxfp8")
    assert model.engine.quantized_layers() == 6
    out = model(ps, ref_s, 1.0, return_output=True)
    assert torch.isfinite(out.audio).all() and out.audio.shape[1] == 600 * int(out.pred_dur.sum())
    direct = KokoroEngine({k: v for k, v in cfg.items() if k != "vocab"}, dequantize_checkpoint(wq, 64, 8), compute_dtype="bfloat16",
                          quantization={"group_size": 64, "bits": 8})
    ids, lens, Tmax = direct.pack_ids([model._ids(ps)])
    pred = direct.forward_text(ids, lens, torch.tensor(ref_s, device=direct.device), torch.ones(1, device=direct.device))
    np.testing.assert_array_equal(pred.cpu().numpy()[0, : int(lens[0])], out.pred_dur.cpu().numpy())
    m32 = load_model(str(d), compute_dtype="float32")  # the exact path never quantises
    assert m32.engine.quantized_layers() == 0
