"""GPU parity tests, kernel level: every HIP kernel family against the CPU oracle / a plain fp32
PyTorch-CPU statement of the same op, called THROUGH THE C ABI (kk_op_*).

Tolerances are written next to each assertion.  fp32 kernels differ from the oracle only by
summation order, so relative 1e-5..1e-4 of the tensor's max is the bar; the north-star tolerance
(1e-3 on the waveform) applies to the end-to-end tests in test_gpu_forward.py.
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import kokoro_oracle as O
from _util import err_stats, report

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from mlx_audio_amd import _lib

    return _lib.load()


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(device="cuda", dtype=dtype).contiguous()


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_w(w_oki):
    """[O][K][I] -> packed [K][I][ldw] (ldw = O rounded up to 64), as kk_finalize does."""
    O_, K, I = w_oki.shape
    ldw = (O_ + 63) // 64 * 64
    p = np.zeros((K, I, ldw), np.float32)
    p[:, :, :O_] = np.transpose(w_oki, (1, 2, 0))
    return p, ldw


def run_conv(lib, x_nlc, w_oki, bias, *, transposed=False, stride=1, pad=0, dil=1, in_shift=0, in_slope=1.0, act=0, act_slope=0.0,
             res=None, scale=1.0, accumulate=False, out_init=None, Lout=None, lin=None, lout=None):
    from mlx_audio_amd import _lib

    B, Lin, Cin = x_nlc.shape
    O_ = w_oki.shape[0]
    wp, ldw = pack_w(w_oki)
    xd, wd = dev(x_nlc), dev(wp)
    bd = dev(bias) if bias is not None else None
    out = dev(out_init) if out_init is not None else torch.full((B, Lout, O_), 7.0, device="cuda")
    rd = dev(res) if res is not None else None
    lind = dev(np.asarray(lin, np.int32), torch.int32) if lin is not None else None
    loutd = dev(np.asarray(lout, np.int32), torch.int32) if lout is not None else None
    rc = lib.kk_op_conv1d(stream(), B, P(xd), Cin, Lin, P(lind), P(wd), ldw, P(bd), Cin, O_, w_oki.shape[1], int(transposed), stride, pad,
                          dil, in_shift, in_slope, act, act_slope, P(rd), O_, scale, int(accumulate), P(out), O_, Lout, P(loutd),
                          _lib.KK_F32, _lib.KK_F32)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    return out.cpu().numpy()


CONV_CASES = [
    # name, Cin, Cout, K, stride, pad, dil, L
    ("k1_linear", 96, 80, 1, 1, 0, 1, 70),
    ("k3_p1", 50, 70, 3, 1, 1, 1, 200),
    ("k5_p2", 64, 64, 5, 1, 2, 1, 131),
    ("k7_d3", 32, 32, 7, 1, 9, 3, 333),
    ("k11_d5", 16, 24, 11, 1, 25, 5, 400),
    ("k3_s2_c1", 1, 1, 3, 2, 1, 1, 112),
    ("k12_s6_c22", 22, 40, 12, 6, 3, 1, 6721),
    ("k7_c22out", 16, 22, 7, 1, 3, 1, 500),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv1d_generic(lib, case):
    name, Cin, Cout, K, s, p, d, L = case
    rng = np.random.default_rng(hash(name) % 2**31)
    B = 2
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, K, Cin)) / math.sqrt(K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = F.conv1d(torch.tensor(x).transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(b), s, p, d).transpose(1, 2).numpy()
    got = run_conv(lib, x, w, b, stride=s, pad=p, dil=d, Lout=ref.shape[1])
    e = err_stats(got, ref)
    report(f"conv1d/{name}", **e)
    assert e["rel_max"] < 2e-5  # fp32, summation order only


def test_conv1d_epilogue_residual_scale_accumulate_gelu_lrelu(lib):
    rng = np.random.default_rng(3)
    B, L, Cin, Cout, K = 2, 150, 48, 40, 3
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, K, Cin)) / math.sqrt(K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    res = rng.standard_normal((B, L, Cout)).astype(np.float32)
    init = rng.standard_normal((B, L, Cout)).astype(np.float32)
    xin = np.where(x > 0, x, x * 0.1)
    base = F.conv1d(torch.tensor(xin).transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(b), 1, 1).transpose(1, 2)
    ref = ((F.gelu(base) + torch.tensor(res)) * 0.25 + torch.tensor(init)).numpy()
    got = run_conv(lib, x, w, b, pad=1, in_slope=0.1, act=2, res=res, scale=0.25, accumulate=True, out_init=init, Lout=L)
    e = err_stats(got, ref)
    report("conv1d/epilogue_gelu_res_scale_acc", **e)
    assert e["rel_max"] < 2e-5
    ref2 = F.leaky_relu(F.conv1d(torch.tensor(x).transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(b), 1, 1), 0.2).transpose(1, 2).numpy()
    got2 = run_conv(lib, x, w, b, pad=1, act=1, act_slope=0.2, Lout=L)
    assert err_stats(got2, ref2)["rel_max"] < 2e-5


def test_conv1d_ragged_lengths_equal_independent_calls(lib):
    """Rows past an utterance's length are zero and do not leak into valid rows (== B independent calls)."""
    rng = np.random.default_rng(4)
    B, L, C, K, d = 3, 90, 24, 7, 3
    lens = [90, 37, 5]
    x = rng.standard_normal((B, L, C)).astype(np.float32)
    for b, n in enumerate(lens):
        x[b, n:] = 0  # invariant kept by every producer kernel
    w = (rng.standard_normal((C, K, C)) / math.sqrt(K * C)).astype(np.float32)
    bias = rng.standard_normal(C).astype(np.float32)
    got = run_conv(lib, x, w, bias, pad=9, dil=d, Lout=L, lin=lens, lout=lens)
    for b, n in enumerate(lens):
        ref = F.conv1d(torch.tensor(x[b : b + 1, :n]).transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(bias), 1, 9, d).transpose(1, 2).numpy()[0]
        assert err_stats(got[b, :n], ref)["rel_max"] < 2e-5
        assert np.all(got[b, n:] == 0)


@pytest.mark.parametrize("cfg", [(20, 10, 5, 64, 32, 28), (12, 6, 3, 32, 16, 100), (3, 2, 1, 8, 8, 33)], ids=["ups0", "ups1", "k3s2"])
def test_conv_transpose1d(lib, cfg):
    K, s, p, Cin, Cout, L = cfg
    rng = np.random.default_rng(K)
    B = 2
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    w_iko = (rng.standard_normal((Cin, K, Cout)) / math.sqrt(K * Cin / s)).astype(np.float32)  # weight_v layout of Generator.ups
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = F.conv_transpose1d(torch.tensor(x).transpose(1, 2), torch.tensor(w_iko).permute(0, 2, 1), torch.tensor(b), s, p).transpose(1, 2).numpy()
    w_oki = np.transpose(w_iko, (2, 1, 0))  # packed form is [K][Cin][Cout] either way
    got = run_conv(lib, x, w_oki, b, transposed=True, stride=s, pad=p, Lout=ref.shape[1])
    e = err_stats(got, ref)
    report(f"convT/k{K}s{s}", **e)
    assert e["rel_max"] < 2e-5


def test_conv1x1_with_nearest_upsampled_input(lib):
    rng = np.random.default_rng(9)
    B, L, Cin, Cout = 2, 41, 30, 20
    x = rng.standard_normal((B, L, Cin)).astype(np.float32)
    w = rng.standard_normal((Cout, 1, Cin)).astype(np.float32)
    xu = np.repeat(x, 2, axis=1)
    ref = np.einsum("blc,oc->blo", xu, w[:, 0, :])
    got = run_conv(lib, x, w, None, in_shift=1, Lout=2 * L)
    assert err_stats(got, ref)["rel_max"] < 2e-5


def _oracle_stub(weights=None):
    import mlx_audio_amd.params as Pm

    cfg = Pm.tiny_config()
    return O.KokoroOracle(weights or {}, cfg)


@pytest.mark.parametrize("act", ["snake", "lrelu", "lrelu_pool"])
def test_adain_act(lib, act):
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(11)
    B, L, C = 2, 700, 50
    lens = [700, 333]
    x = (rng.standard_normal((B, L, C)) * 2 + 3).astype(np.float32)
    for b, n in enumerate(lens):
        x[b, n:] = 0
    gb = (rng.standard_normal((B, 2 * C)) * 0.5).astype(np.float32)
    alpha = rng.uniform(0.5, 1.5, C).astype(np.float32)
    pw = rng.standard_normal((C, 3)).astype(np.float32)
    pb = rng.standard_normal(C).astype(np.float32)
    pool = act.endswith("pool")
    Lout = 2 * L if pool else L
    Cpad = 64
    out = torch.full((B, Lout, Cpad), 5.0, device="cuda")
    scratch = torch.empty(B * ((L + 511) // 512) * 2 * C + 2 * B * C + 64, device="cuda")
    xd, gbd, ald, pwd, pbd = dev(x), dev(gb), dev(alpha), dev(pw), dev(pb)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_adain(stream(), B, P(xd), C, L, P(lend), C, P(gbd), 2 * C, _lib.ACT_SNAKE if act == "snake" else _lib.ACT_LRELU, 0.2,
                         P(ald), int(pool), P(pwd), P(pbd), P(out), Cpad, Cpad, Lout, P(scratch), scratch.numel(), _lib.KK_F32, 0)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for b, n in enumerate(lens):
        xb = torch.tensor(x[b : b + 1, :n]).transpose(1, 2)  # [1,C,n]
        mean = xb.mean(-1, keepdim=True)
        var = ((xb - mean) ** 2).mean(-1, keepdim=True)
        xn = (xb - mean) / torch.sqrt(var + 1e-5)
        y = (1 + torch.tensor(gb[b, :C])[None, :, None]) * xn + torch.tensor(gb[b, C:])[None, :, None]
        if act == "snake":
            a = torch.tensor(alpha)[None, :, None]
            y = y + (1 / a) * torch.sin(a * y) ** 2
        else:
            y = torch.where(y > 0, y, 0.2 * y)
        if pool:
            y = F.conv_transpose1d(y, torch.tensor(pw)[:, None, :], torch.tensor(pb), 2, 1, 0, C)
            y = F.pad(y, (1, 0))
        ref = y.transpose(1, 2).numpy()[0]
        no = 2 * n if pool else n
        e = err_stats(got[b, :no, :C], ref)
        report(f"adain/{act}/b{b}", **e)
        assert e["rel_max"] < 1e-5
        assert np.all(got[b, no:] == 0) and np.all(got[b, :, C:] == 0)


def test_layernorm_and_adaln(lib):
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(12)
    B, L = 2, 37
    for C in (128, 512, 768):
        x = rng.standard_normal((B, L, C)).astype(np.float32) * 3 + 1
        r = rng.standard_normal((B, L, C)).astype(np.float32)
        w = rng.standard_normal(C).astype(np.float32)
        b = rng.standard_normal(C).astype(np.float32)
        out = torch.empty((B, L, C), device="cuda")
        xd, rd, wd, bd = dev(x), dev(r), dev(w), dev(b)
        rc = lib.kk_op_layernorm(stream(), B, P(xd), C, P(rd), C, L, None, C, P(wd), P(bd), None, 0, 1e-12, 0, 0.0, P(out), C, _lib.KK_F32)
        assert rc == 0, lib.kk_last_error()
        ref = F.layer_norm(torch.tensor(x + r), (C,), torch.tensor(w), torch.tensor(b), 1e-12).numpy()
        torch.cuda.synchronize()
        e = err_stats(out.cpu().numpy(), ref)
        report(f"layernorm/C{C}", **e)
        assert e["rel_max"] < 1e-5
    C = 64
    x = rng.standard_normal((B, L, C)).astype(np.float32)
    gb = rng.standard_normal((B, 2 * C)).astype(np.float32)
    out = torch.empty((B, L, C), device="cuda")
    xd, gd = dev(x), dev(gb)
    rc = lib.kk_op_layernorm(stream(), B, P(xd), C, None, 0, L, None, C, None, None, P(gd), 2 * C, 1e-5, 0, 0.0, P(out), C, _lib.KK_F32)
    assert rc == 0, lib.kk_last_error()
    xn = F.layer_norm(torch.tensor(x), (C,), None, None, 1e-5)
    ref = ((1 + torch.tensor(gb[:, None, :C])) * xn + torch.tensor(gb[:, None, C:])).numpy()
    torch.cuda.synchronize()
    assert err_stats(out.cpu().numpy(), ref)["rel_max"] < 1e-5


@pytest.mark.parametrize("H,I,L", [(32, 48, 21), (256, 640, 40)])
def test_lstm_bidirectional(lib, H, I, L):
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(H)
    B = 2
    lens = [L, max(1, L // 3)]
    s = 1.0 / math.sqrt(H)
    w = {}
    for d in ("forward", "backward"):
        w[f"l.Wx_{d}"] = rng.uniform(-s, s, (4 * H, I)).astype(np.float32)
        w[f"l.Wh_{d}"] = rng.uniform(-s, s, (4 * H, H)).astype(np.float32)
        w[f"l.bias_ih_{d}"] = rng.uniform(-s, s, 4 * H).astype(np.float32)
        w[f"l.bias_hh_{d}"] = rng.uniform(-s, s, 4 * H).astype(np.float32)
    orc = _oracle_stub(w)
    x = rng.standard_normal((B, L, I)).astype(np.float32)
    xproj = np.zeros((B, L, 2, 4 * H), np.float32)
    whT = np.zeros((2, H, 4 * H), np.float32)
    for di, d in enumerate(("forward", "backward")):
        xproj[:, :, di] = (w[f"l.bias_ih_{d}"] + w[f"l.bias_hh_{d}"]) + x @ w[f"l.Wx_{d}"].T
        whT[di] = w[f"l.Wh_{d}"].T
    out = torch.full((B, L, 2 * H), 3.0, device="cuda")
    xp, wt = dev(xproj), dev(whT)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_lstm(stream(), B, P(xp), P(wt), H, L, P(lend), P(out), 2 * H, _lib.KK_F32)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for b, n in enumerate(lens):
        ref = orc.lstm(torch.tensor(x[b : b + 1, :n]), "l").numpy()[0]
        e = err_stats(got[b, :n], ref)
        report(f"lstm/H{H}/b{b}", **e)
        assert e["max_abs"] < 2e-5  # outputs are in (-1, 1)
        assert np.all(got[b, n:] == 0)


def test_attention(lib):
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(21)
    B, T, heads = 2, 130, 3
    hs = heads * 64
    lens = [130, 77]
    qkv = rng.standard_normal((B, T, 3 * hs)).astype(np.float32)
    out = torch.full((B, T, hs), 9.0, device="cuda")
    qd = dev(qkv)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_attention(stream(), B, P(qd), 3 * hs, T, P(lend), heads, P(out), hs, _lib.KK_F32)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for b, n in enumerate(lens):
        t = torch.tensor(qkv[b, :n])
        q, k, v = [t[:, i * hs : (i + 1) * hs].view(n, heads, 64).permute(1, 0, 2) for i in range(3)]
        pr = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
        ref = (pr @ v).permute(1, 0, 2).reshape(n, hs).numpy()
        e = err_stats(got[b, :n], ref)
        report(f"attention/b{b}", **e)
        assert e["rel_max"] < 1e-5
        assert np.all(got[b, n:] == 0)


def test_attention_bf16_mfma(lib):
    """bf16 tensors take the matrix-core kernel (one wave per 32-query tile); reference = fp32 softmax attention on the
    same bf16-rounded inputs.  P and the output are rounded to bf16: tolerance 1.5e-2 of the tensor's max."""
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(22)
    B, T, heads = 3, 130, 3
    hs = heads * 64
    lens = [130, 77, 33]
    qkv = torch.tensor(rng.standard_normal((B, T, 3 * hs)).astype(np.float32)).to(torch.bfloat16)
    out = torch.full((B, T, hs), 9.0, device="cuda", dtype=torch.bfloat16)
    qd = qkv.cuda()
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_attention(stream(), B, P(qd), 3 * hs, T, P(lend), heads, P(out), hs, _lib.KK_BF16)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    for b, n in enumerate(lens):
        t = qkv[b, :n].float()
        q, k, v = [t[:, i * hs : (i + 1) * hs].view(n, heads, 64).permute(1, 0, 2) for i in range(3)]
        pr = torch.softmax(q @ k.transpose(-1, -2) / 8.0, dim=-1)
        ref = (pr @ v).permute(1, 0, 2).reshape(n, hs).numpy()
        e = err_stats(got[b, :n], ref)
        report(f"attention_bf16_mfma/b{b}", **e)
        assert e["rel_max"] < 1.5e-2
        assert np.all(got[b, n:] == 0)


def test_source_and_stft(lib):
    """SineGen + SourceModuleHnNSF + STFT(20) against the oracle with INJECTED noise.  The phase channels are
    compared where the bin magnitude is not tiny (the angle of a ~1e-7 bin is noise in any implementation)."""
    from mlx_audio_amd import _lib
    import mlx_audio_amd.params as Pm

    rng = np.random.default_rng(5)
    cfg = Pm.tiny_config()
    lw = rng.standard_normal((1, 9)).astype(np.float32) * 0.7
    lb = np.array([0.05], np.float32)
    orc = O.KokoroOracle({"decoder.generator.m_source.l_linear.weight": lw, "decoder.generator.m_source.l_linear.bias": lb}, cfg)
    B, L2 = 2, 48
    f0 = (rng.standard_normal((B, L2)) * 80 + 120).astype(np.float32)
    f0[0, 5:9] = -3.0  # unvoiced stretch
    noise = rng.standard_normal((B, 300 * L2, 9)).astype(np.float32)
    Tf = 60 * L2 + 1
    phase = torch.empty((B, 9, L2), device="cuda")
    hs = torch.empty((B, 300 * L2), device="cuda")
    har = torch.empty((B, Tf, 22), device="cuda")
    fd, ld_, nd = dev(f0), dev(lw[0]), dev(noise)
    rc = lib.kk_op_source_stft(stream(), B, P(fd), L2, None, P(ld_), float(lb[0]), _lib.NOISE_INJECTED, P(nd), C.c_uint64(0), P(phase), P(hs),
                               P(har), 22, _lib.KK_F32)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    for b in range(B):
        har_ref, hs_ref = orc.har_features(f0[b : b + 1], None, noise[b : b + 1])
        e = err_stats(hs.cpu().numpy()[b], hs_ref[0])
        report(f"source/har_source/b{b}", **e)
        # phase accumulators reach 1e4 rad in float32: sin() arguments agree to a few ulp -> ~1e-3 abs is the honest bar
        assert e["max_abs"] < 5e-3 and e["rms_rel"] < 2e-3
        got = har.cpu().numpy()[b]  # [Tf, 22]
        mag_ref, ph_ref = har_ref[0, :11].T, har_ref[0, 11:].T
        em = err_stats(got[:, :11], mag_ref)
        report(f"source/stft_mag/b{b}", **em)
        assert em["max_abs"] < 5e-3
        strong = mag_ref > 1e-2
        dphi = np.angle(np.exp(1j * (got[:, 11:] - ph_ref)))
        report(f"source/stft_phase/b{b}", max_abs=float(np.abs(dphi[strong]).max()), frac_strong=float(strong.mean()))
        assert np.abs(dphi[strong]).max() < 0.5


def test_istft_head(lib):
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(7)
    B, Tf = 3, 481
    lens = [481, 121, 1]
    x = (rng.standard_normal((B, Tf, 22)) * 1.5).astype(np.float32)
    orc = _oracle_stub()
    wav = torch.full((B, 5 * (Tf - 1)), 4.0, device="cuda")
    xd = dev(x)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_istft_head(stream(), B, P(xd), 22, Tf, P(lend), P(wav), _lib.KK_F32, 0)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got = wav.cpu().numpy()
    for b, n in enumerate(lens):
        if n > 1:
            ref = orc.istft_head(np.transpose(x[b : b + 1, :n], (0, 2, 1)))[0, 0]
            e = err_stats(got[b, : 5 * (n - 1)], ref)
            report(f"istft_head/b{b}", **e)
            assert e["rel_max"] < 2e-5
        assert np.all(got[b, 5 * (n - 1) :] == 0)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_istft_head_fast_path(lib, dtype):
    """The bf16 mode's iSTFT head (hardware exp / sin, polynomial sin / cos of the phase's sine, literal window tables, edge waves
    only rebuild the window sum): same oracle, utterance lengths that put edges at every position of a wave's 61 hop blocks."""
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(8)
    lens = [733, 1, 2, 3, 4, 5, 59, 60, 61, 62, 63, 64, 65, 122, 123, 244, 245, 246, 732]
    B, Tf = len(lens), 733
    x = (rng.standard_normal((B, Tf, 22)) * 1.5).astype(np.float32)
    if dtype == "bf16":
        x = torch.tensor(x).to(torch.bfloat16).float().numpy()
    orc = _oracle_stub()
    wav = torch.full((B, 5 * (Tf - 1)), 4.0, device="cuda")
    ldx = 24  # the model's conv_post pitch (22 rounded up to 8)
    xd = torch.zeros((B, Tf, ldx), dtype=torch.bfloat16 if dtype == "bf16" else torch.float32, device="cuda")
    xd[:, :, :22] = torch.tensor(x).to(xd.dtype)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_istft_head(stream(), B, P(xd), ldx, Tf, P(lend), P(wav), _lib.KK_BF16 if dtype == "bf16" else _lib.KK_F32, 1)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got = wav.cpu().numpy()
    worst = 0.0
    for b, n in enumerate(lens):
        if n > 1:
            ref = orc.istft_head(np.transpose(x[b : b + 1, :n], (0, 2, 1)))[0, 0]
            e = err_stats(got[b, : 5 * (n - 1)], ref)
            worst = max(worst, e["rel_max"])
            assert e["rel_max"] < 1e-4, (n, e)  # hardware exp / sin: ~1e-6 relative each; measured worst case reported below
        assert np.all(got[b, 5 * (n - 1) :] == 0), n
    report(f"istft_head_fast/{dtype}", worst_rel_max=worst)


def test_fused_conv_post_istft_head(lib):
    """kk_head.hip: LeakyReLU(0.01) -> conv_post (128 -> 22, k 7, pad 3) -> exp / sin -> inverse STFT -> overlap-add in ONE kernel, against
    torch's conv1d on the same bf16 operands (fp32 accumulation) followed by the oracle's iSTFT head (istftnet.py:798-806,497-523;
    utils.py:104-158).  Utterance lengths put edges at every interesting position of the 253-hop-block tiles and the 64-lane waves; the
    bounds are those of test_istft_head_fast_path (the iSTFT arithmetic is shared, kk_istft_math.h)."""
    import torch.nn.functional as F

    rng = np.random.default_rng(18)
    lens = [1013, 1, 2, 3, 4, 5, 61, 64, 65, 250, 251, 252, 253, 254, 255, 256, 257, 259, 505, 506, 507, 508, 759, 1012]
    B, Tf, C = len(lens), 1013, 128
    x = torch.tensor(rng.standard_normal((B, Tf, C)).astype(np.float32)).to(torch.bfloat16)
    w = torch.tensor((rng.standard_normal((7, 22, C)) * (0.6 / (7 * C) ** 0.5)).astype(np.float32)).to(torch.bfloat16)  # [tap][cout][cin]
    bias = (rng.standard_normal(22) * 0.3).astype(np.float32)
    xd, wd, bd = x.cuda(), w.cuda(), dev(bias)
    for b, n in enumerate(lens):
        xd[b, n:] = 0  # rows past an utterance hold zeros in the model (every producer writes them)
    wf = torch.empty(7 * 8 * 64 * 8, dtype=torch.bfloat16, device="cuda")
    assert lib.kk_op_pack_head_w(stream(), P(wd), P(wf)) == 0, lib.kk_last_error()
    wav = torch.full((B, 5 * (Tf - 1)), 4.0, device="cuda")
    cp = torch.full((B, Tf, 24), 9.0, dtype=torch.bfloat16, device="cuda")
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_conv_post_istft(stream(), B, P(xd), C, Tf, P(lend), P(wf), P(bd), 0.01, P(wav), P(cp), 24)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got, cpg = wav.cpu().numpy(), cp.float().cpu().numpy()
    orc = _oracle_stub()
    worst, worst_cp = 0.0, 0.0
    for b, n in enumerate(lens):
        xi = F.leaky_relu(x[b : b + 1, :n].float(), 0.01).to(torch.bfloat16).float()  # the kernel rounds the activated input to bf16
        y = F.conv1d(xi.transpose(1, 2), w.float().permute(1, 2, 0).contiguous(), torch.tensor(bias), padding=3).numpy()  # [1, 22, n]
        e = err_stats(cpg[b, :n, :22], y[0].T)
        worst_cp = max(worst_cp, e["rel_max"])
        assert e["rel_max"] < 8e-3, (n, e)  # conv_post as stored for the debug hook: one bf16 rounding
        assert np.all(cpg[b, n:, :22] == 9.0)  # (rows past the utterance are not written by the kernel; the model zero-fills the buffer)
        if n > 1:
            ref = orc.istft_head(y.astype(np.float32))[0, 0]
            e = err_stats(got[b, : 5 * (n - 1)], ref)
            worst = max(worst, e["rel_max"])
            assert e["rel_max"] < 1e-4, (n, e)
        assert np.all(got[b, 5 * (n - 1) :] == 0), n
    report("fused_head", worst_rel_max_wav=worst, worst_rel_max_conv_post=worst_cp)
    # no debug tensor: same waveform bits
    wav2 = torch.full((B, 5 * (Tf - 1)), 4.0, device="cuda")
    assert lib.kk_op_conv_post_istft(stream(), B, P(xd), C, Tf, P(lend), P(wf), P(bd), 0.01, P(wav2), None, 0) == 0
    torch.cuda.synchronize()
    assert torch.equal(wav, wav2)


# ------------------------------------------------------------------------------------------------
# bf16 MFMA convolution kernel
# ------------------------------------------------------------------------------------------------
def pack_w_bf16(w_oki):
    O_, K, I = w_oki.shape
    CinP, CoutP = (I + 63) // 64 * 64, (O_ + 127) // 128 * 128
    p = np.zeros((K, CoutP, CinP), np.float32)
    p[:, :O_, :I] = np.transpose(w_oki, (1, 0, 2))
    return torch.tensor(p).to(torch.bfloat16), CinP, CoutP


_VARIANT4 = False  # set by the `mfma4` fixture: route the kk_op_conv1d_bf16* calls to variant 4 (weights in fragment order)


class _wfrag:
    """Context: re-lay the [Kw][CoutP][CinP] pack out in MFMA fragment order on the device and select variant 4."""

    def __init__(self, lib, wd):
        self.lib, self.wd, self.on = lib, wd, _VARIANT4

    def __enter__(self):
        if self.on:
            K, CoutP, CinP = self.wd.shape
            self.wf = torch.empty_like(self.wd)
            rc = self.lib.kk_op_pack_w_frag(stream(), P(self.wd), P(self.wf), K, CoutP, CinP)
            assert rc == 0, self.lib.kk_last_error()
            self.lib.kk_debug_set_op_wfrag(P(self.wf))
            self.lib.kk_debug_set_op_variant(5 if self.on == 5 else 4)
        return self

    def __exit__(self, *a):
        if self.on:
            torch.cuda.synchronize()
            self.lib.kk_debug_set_op_wfrag(None)
            self.lib.kk_debug_set_op_variant(4)


def run_conv_bf16(lib, x_nlc, w_oki, bias, *, transposed=False, stride=1, pad=0, dil=1, in_shift=0, in_slope=1.0, act=0, act_slope=0.0,
                  res=None, scale=1.0, accumulate=False, out_init=None, Lout=None, lin=None, lout=None, out_f32=False):
    from mlx_audio_amd import _lib

    B, Lin, Cin = x_nlc.shape
    O_ = w_oki.shape[0]
    wp, CinP, CoutP = pack_w_bf16(w_oki)
    xpad = np.zeros((B, Lin, CinP), np.float32)
    xpad[:, :, :Cin] = x_nlc
    odt = torch.float32 if out_f32 else torch.bfloat16
    xd, wd = dev(xpad, torch.bfloat16), wp.cuda().contiguous()
    bp = np.zeros(CoutP, np.float32)
    if bias is not None:
        bp[:O_] = bias
    bd = dev(bp)
    out = dev(out_init, odt) if out_init is not None else torch.full((B, Lout, O_), 7.0, device="cuda", dtype=odt)
    rd = dev(res, odt) if res is not None else None
    lind = dev(np.asarray(lin, np.int32), torch.int32) if lin is not None else None
    loutd = dev(np.asarray(lout, np.int32), torch.int32) if lout is not None else None
    with _wfrag(lib, wd):
        rc = lib.kk_op_conv1d_bf16(stream(), B, P(xd), CinP, Lin, P(lind), P(wd), CinP, CoutP, P(bd), O_, w_oki.shape[1], int(transposed), stride,
                                   pad, dil, in_shift, in_slope, act, act_slope, P(rd), O_, scale, int(accumulate), P(out), O_, Lout, P(loutd),
                                   _lib.KK_F32 if out_f32 else _lib.KK_BF16)
        assert rc == 0, lib.kk_last_error()
        torch.cuda.synchronize()
    return out.float().cpu().numpy()


def _bf(a):
    return torch.tensor(a).to(torch.bfloat16).float().numpy()


MFMA_CASES = [
    # name, Cin, Cout, K, pad, dil, L
    ("gen_k3", 128, 128, 3, 1, 1, 1000),
    ("gen_k7_d3", 128, 128, 7, 9, 3, 777),
    ("gen_k11_d5", 256, 256, 11, 25, 5, 300),
    ("dec_k3_cin1090", 1090, 256, 3, 1, 1, 131),
    ("linear_768_2304", 768, 2304, 1, 0, 1, 130),
    ("cout64_cin514", 514, 64, 1, 0, 1, 70),
]


@pytest.mark.parametrize("case", MFMA_CASES, ids=[c[0] for c in MFMA_CASES])
def test_conv_mfma_bf16(lib, mfma4, case):
    name, Cin, Cout, K, p, d, L = case
    rng = np.random.default_rng(hash(name) % 2**31)
    B = 2
    x = _bf(rng.standard_normal((B, L, Cin)).astype(np.float32))
    w = _bf((rng.standard_normal((Cout, K, Cin)) / math.sqrt(K * Cin)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = F.conv1d(torch.tensor(x).transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(b), 1, p, d).transpose(1, 2).numpy()
    got32 = run_conv_bf16(lib, x, w, b, pad=p, dil=d, Lout=L, out_f32=True)
    e = err_stats(got32, ref)
    report(f"conv_mfma/{name}/f32out", **e)
    assert e["rel_max"] < 2e-5  # same bf16 operands, fp32 accumulation: summation order only
    got = run_conv_bf16(lib, x, w, b, pad=p, dil=d, Lout=L)
    e = err_stats(got, ref)
    report(f"conv_mfma/{name}/bf16out", **e)
    assert e["rel_max"] < 5e-3  # one bf16 rounding of the result (2^-9 relative)


def test_conv_mfma_random_shape_sweep(lib, mfma4):
    """Seeded sweep over channel counts (pad channels, Cout not a multiple of 128), taps, dilations (halo up to the kernel's limit of 50),
    lengths around the tile edges (191 / 192 / 193 / 385 rows) and ragged batches: the MFMA kernels against torch on the same bf16 operands."""
    rng = np.random.default_rng(2024)
    cins = [16, 24, 64, 72, 128, 200, 514]
    couts = [16, 40, 64, 128, 136, 256]
    taps = [(1, 1), (3, 1), (3, 5), (5, 2), (7, 3), (11, 5), (2, 1), (4, 7)]
    lengths = [1, 5, 63, 191, 192, 193, 385, 600]
    for case in range(14):
        Cin, Cout = int(rng.choice(cins)), int(rng.choice(couts))
        K, d = taps[int(rng.integers(len(taps)))]
        L = int(rng.choice(lengths))
        B = int(rng.integers(1, 4))
        pad = int(rng.integers(0, (K - 1) * d + 1))  # any left padding, symmetric or not (causal = (K-1)*d)
        lens = [L] + [int(rng.integers(1, L + 1)) for _ in range(B - 1)]
        x = _bf(rng.standard_normal((B, L, Cin)).astype(np.float32))
        for b, n in enumerate(lens):
            x[b, n:] = 0
        w = _bf((rng.standard_normal((Cout, K, Cin)) / math.sqrt(K * Cin)).astype(np.float32))
        bias = rng.standard_normal(Cout).astype(np.float32)
        use_res = bool(rng.integers(2))
        res = _bf(rng.standard_normal((B, L, Cout)).astype(np.float32)) if use_res else None
        got = run_conv_bf16(lib, x, w, bias, pad=pad, dil=d, res=res, Lout=L, lin=lens, lout=lens)
        for b, n in enumerate(lens):
            xp = F.pad(torch.tensor(x[b, :n])[None].transpose(1, 2), (pad, (K - 1) * d - pad))
            ref = F.conv1d(xp, torch.tensor(w).permute(0, 2, 1), torch.tensor(bias), 1, 0, d).transpose(1, 2)[0].numpy()
            if use_res:
                ref = ref + res[b, :n]
            e = err_stats(got[b, :n], ref)
            report(f"conv_mfma{4 if mfma4 else 2}/sweep{case}_cin{Cin}_cout{Cout}_k{K}d{d}_L{L}/b{b}", **e)
            assert e["max_abs"] < 6e-3 * max(1.0, e["ref_max"]), (case, Cin, Cout, K, d, L, pad, lens, e)
            assert np.all(got[b, n:] == 0)


def test_conv_mfma_epilogue_and_ragged(lib, mfma4):
    rng = np.random.default_rng(33)
    B, L, C, K, d = 3, 300, 128, 7, 3
    lens = [300, 129, 5]
    x = _bf(rng.standard_normal((B, L, C)).astype(np.float32))
    for b, n in enumerate(lens):
        x[b, n:] = 0
    w = _bf((rng.standard_normal((C, K, C)) / math.sqrt(K * C)).astype(np.float32))
    bias = rng.standard_normal(C).astype(np.float32)
    res = _bf(rng.standard_normal((B, L, C)).astype(np.float32))
    init = _bf(rng.standard_normal((B, L, C)).astype(np.float32))
    got = run_conv_bf16(lib, x, w, bias, pad=9, dil=d, in_slope=0.1, res=res, scale=1 / 3, accumulate=True, out_init=init, Lout=L, lin=lens,
                        lout=lens, out_f32=False)
    for b, n in enumerate(lens):
        xin = np.where(x[b, :n] > 0, x[b, :n], _bf(x[b, :n] * 0.1))
        base = F.conv1d(torch.tensor(xin)[None].transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(bias), 1, 9, d).transpose(1, 2)[0]
        ref = ((base + torch.tensor(res[b, :n])) * (1 / 3) + torch.tensor(init[b, :n])).numpy()
        e = err_stats(got[b, :n], ref)
        report(f"conv_mfma/epilogue_ragged/b{b}", **e)
        assert e["rel_max"] < 6e-3
        assert np.all(got[b, n:] == 0)


@pytest.mark.parametrize("cfg", [(20, 10, 5, 512, 256, 130), (12, 6, 3, 256, 128, 300)], ids=["ups0", "ups1"])
def test_conv_transpose_mfma(lib, cfg):
    K, s, p, Cin, Cout, L = cfg
    rng = np.random.default_rng(K + 100)
    B = 2
    x = _bf(rng.standard_normal((B, L, Cin)).astype(np.float32))
    w_iko = _bf((rng.standard_normal((Cin, K, Cout)) / math.sqrt(K * Cin / s)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = F.conv_transpose1d(torch.tensor(x).transpose(1, 2), torch.tensor(w_iko).permute(0, 2, 1), torch.tensor(b), s, p).transpose(1, 2).numpy()
    got = run_conv_bf16(lib, x, np.transpose(w_iko, (2, 1, 0)), b, transposed=True, stride=s, pad=p, Lout=ref.shape[1], out_f32=True)
    e = err_stats(got, ref)
    report(f"convT_mfma/k{K}s{s}", **e)
    assert e["rel_max"] < 2e-5


def test_conv1x1_mfma_upsampled_input(lib):
    rng = np.random.default_rng(19)
    B, L, Cin, Cout = 2, 141, 1090, 512
    x = _bf(rng.standard_normal((B, L, Cin)).astype(np.float32))
    w = _bf((rng.standard_normal((Cout, 1, Cin)) / math.sqrt(Cin)).astype(np.float32))
    ref = np.einsum("blc,oc->blo", np.repeat(x, 2, axis=1), w[:, 0, :])
    got = run_conv_bf16(lib, x, w, None, in_shift=1, Lout=2 * L, out_f32=True)
    assert err_stats(got, ref)["rel_max"] < 2e-5


def test_lstm_h256_bf16_on_chip(lib):
    """The on-chip-weights recurrence (bf16 Wh, bf16-rounded h into the dot products, fp32 state) against the oracle run
    on the same bf16-rounded weights: differences come from rounding h to bf16 inside the recurrent product only."""
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(77)
    H, I, L, B = 256, 640, 60, 3
    lens = [60, 17, 1]
    s = 1.0 / math.sqrt(H)
    w = {}
    for d in ("forward", "backward"):
        w[f"l.Wx_{d}"] = rng.uniform(-s, s, (4 * H, I)).astype(np.float32)
        w[f"l.Wh_{d}"] = _bf(rng.uniform(-s, s, (4 * H, H)).astype(np.float32))
        w[f"l.bias_ih_{d}"] = rng.uniform(-s, s, 4 * H).astype(np.float32)
        w[f"l.bias_hh_{d}"] = rng.uniform(-s, s, 4 * H).astype(np.float32)
    orc = _oracle_stub(w)
    x = rng.standard_normal((B, L, I)).astype(np.float32)
    xproj = np.zeros((B, L, 2, 4 * H), np.float32)
    whb = np.zeros((2, 4 * H, H), np.float32)
    for di, d in enumerate(("forward", "backward")):
        xproj[:, :, di] = (w[f"l.bias_ih_{d}"] + w[f"l.bias_hh_{d}"]) + x @ w[f"l.Wx_{d}"].T
        whb[di] = w[f"l.Wh_{d}"]
    out = torch.full((B, L, 2 * H), 3.0, device="cuda")
    xp, wt = dev(xproj), dev(whb, torch.bfloat16)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    rc = lib.kk_op_lstm_bf16(stream(), B, P(xp), P(wt), L, P(lend), P(out), 2 * H, _lib.KK_F32)
    assert rc == 0, lib.kk_last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for b, n in enumerate(lens):
        ref = orc.lstm(torch.tensor(x[b : b + 1, :n]), "l").numpy()[0]
        e = err_stats(got[b, :n], ref)
        report(f"lstm_bf16/b{b}", **e)
        assert e["max_abs"] < 2e-2 and e["rms_rel"] < 1e-2  # h is rounded to bf16 (2^-9) before each recurrent product
        assert np.all(got[b, n:] == 0)


def test_conv_mfma_fused_adain_snake_and_stats(lib, mfma4):
    """Fused form of the bf16 generator: AdaIN + Snake while staging the input, column statistics of the output.
    Reference: the same math in fp32 on the bf16-rounded operands."""
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(91)
    B, L, Cn, K, d = 2, 700, 128, 7, 3
    lens = [700, 301]
    x = _bf(rng.standard_normal((B, L, Cn)).astype(np.float32) * 2 + 0.5)
    for b, n in enumerate(lens):
        x[b, n:] = 0
    w = _bf((rng.standard_normal((Cn, K, Cn)) / math.sqrt(K * Cn)).astype(np.float32))
    bias = rng.standard_normal(Cn).astype(np.float32)
    res = _bf(rng.standard_normal((B, L, Cn)).astype(np.float32))
    A = (rng.standard_normal((B, Cn)) * 0.5 + 1).astype(np.float32)
    Bv = (rng.standard_normal((B, Cn)) * 0.3).astype(np.float32)
    alpha = rng.uniform(0.5, 1.5, Cn).astype(np.float32)
    wp, CinP, CoutP = pack_w_bf16(w)
    bp = np.zeros(CoutP, np.float32)
    bp[:Cn] = bias
    xd, wd, bd, rd = dev(x, torch.bfloat16), wp.cuda().contiguous(), dev(bp), dev(res, torch.bfloat16)
    ad, bvd, ald = dev(A), dev(Bv), dev(alpha)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    out = torch.full((B, L, Cn), 7.0, device="cuda", dtype=torch.bfloat16)
    ntmax = (L + 127) // 128
    part = torch.zeros((B, ntmax, 2, Cn), device="cuda")
    nt = C.c_int(0)
    with _wfrag(lib, wd):
        rc = lib.kk_op_conv1d_bf16_fused(stream(), B, P(xd), Cn, L, P(lend), P(wd), CinP, CoutP, P(bd), Cn, Cn, K, 9, d, P(ad), P(bvd), Cn,
                                         _lib.ACT_SNAKE, 0.0, P(ald), P(rd), Cn, 1.0, P(out), Cn, P(part), C.byref(nt))
        assert rc == 0, lib.kk_last_error()
        torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    pt = part.cpu().numpy().reshape(-1)[: B * nt.value * 2 * Cn].reshape(B, nt.value, 2, Cn)
    for b, n in enumerate(lens):
        y = x[b, :n] * A[b][None, :] + Bv[b][None, :]
        y = y + (1.0 / alpha)[None, :] * np.sin(alpha[None, :] * y) ** 2
        y = _bf(y.astype(np.float32))
        base = F.conv1d(torch.tensor(y)[None].transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(bias), 1, 9, d).transpose(1, 2)[0]
        ref = (base + torch.tensor(res[b, :n])).numpy()
        e = err_stats(got[b, :n], ref)
        report(f"conv_mfma_fused/out/b{b}", **e)
        assert e["rel_max"] < 1.5e-2  # bf16 rounding of the transformed input (hardware sin) and of the output
        assert np.all(got[b, n:] == 0)
        # column statistics: of the fp32 values BEFORE the bf16 rounding (round 3) -- equal to the fp32 reference's sums up to summation order,
        # and to the sums of the stored bf16 tensor up to its rounding noise (2^-9 relative per element, averaged over n rows)
        s1, s2 = pt[b, :, 0].sum(0), pt[b, :, 1].sum(0)
        np.testing.assert_allclose(s1, ref.astype(np.float64).sum(0), rtol=1e-3, atol=2e-2)
        np.testing.assert_allclose(s2, (ref.astype(np.float64) ** 2).sum(0), rtol=1e-3)
        g64 = got[b, :n].astype(np.float64)
        noise = 2.0 ** -9 / math.sqrt(3.0) * np.sqrt((g64 ** 2).sum(0))  # std of the summed rounding errors of a column
        assert np.all(np.abs(s1 - g64.sum(0)) <= 5 * noise + 2e-2)
        np.testing.assert_allclose(s2, (g64 ** 2).sum(0), rtol=3e-3)


# ------------------------------------------------------------------------------------------------
# persistent 256-row MFMA kernel (long sequences, Q >= 2048)
# ------------------------------------------------------------------------------------------------
LONG_CASES = [
    ("long_k3", 128, 128, 3, 1, 1, 2500),
    ("long_k11_d5", 128, 128, 11, 25, 5, 2300),
    ("long_k7_c256", 256, 256, 7, 3, 1, 2100),
]


@pytest.fixture(params=[False, True, 5], ids=["lds_staged", "variant4", "variant5"])
def mfma4(request):
    """The three bf16 MFMA kernels: the LDS-staged one, variant 4 (W fragments straight from global memory into registers) and variant 5
    (wave-specialised persistent: 4 MFMA waves + 4 service waves per CU; stride-1 convolutions, everything else falls through to variant 4)."""
    global _VARIANT4
    _VARIANT4 = request.param
    yield request.param
    _VARIANT4 = False


@pytest.mark.parametrize("case", LONG_CASES, ids=[c[0] for c in LONG_CASES])
def test_conv_mfma_long(lib, mfma4, case):
    name, Cin, Cout, K, p, d, L = case
    rng = np.random.default_rng(hash(name) % 2**31)
    B = 3
    lens = [L, L - 777, 130]
    x = _bf(rng.standard_normal((B, L, Cin)).astype(np.float32))
    for b, n in enumerate(lens):
        x[b, n:] = 0
    w = _bf((rng.standard_normal((Cout, K, Cin)) / math.sqrt(K * Cin)).astype(np.float32))
    bias = rng.standard_normal(Cout).astype(np.float32)
    res = _bf(rng.standard_normal((B, L, Cout)).astype(np.float32))
    init = _bf(rng.standard_normal((B, L, Cout)).astype(np.float32))
    got = run_conv_bf16(lib, x, w, bias, pad=p, dil=d, in_slope=0.1, res=res, scale=1 / 3, accumulate=True, out_init=init, Lout=L, lin=lens,
                        lout=lens)
    for b, n in enumerate(lens):
        xin = np.where(x[b, :n] > 0, x[b, :n], _bf(x[b, :n] * 0.1))
        base = F.conv1d(torch.tensor(xin)[None].transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(bias), 1, p, d).transpose(1, 2)[0]
        ref = ((base + torch.tensor(res[b, :n])) * (1 / 3) + torch.tensor(init[b, :n])).numpy()
        e = err_stats(got[b, :n], ref)
        report(f"conv_mfma{4 if mfma4 else 2}/{name}/b{b}", **e)
        assert e["rel_max"] < 6e-3
        assert np.all(got[b, n:] == 0)


def test_conv_mfma_transposed_and_fused(lib, mfma4):
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(123)
    # transposed conv (ups.1 geometry), long rows
    K, s, p, Cin, Cout, L, B = 12, 6, 3, 256, 128, 2200, 2
    x = _bf(rng.standard_normal((B, L, Cin)).astype(np.float32))
    w_iko = _bf((rng.standard_normal((Cin, K, Cout)) / math.sqrt(K * Cin / s)).astype(np.float32))
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = F.conv_transpose1d(torch.tensor(x).transpose(1, 2), torch.tensor(w_iko).permute(0, 2, 1), torch.tensor(b), s, p).transpose(1, 2).numpy()
    got = run_conv_bf16(lib, x, np.transpose(w_iko, (2, 1, 0)), b, transposed=True, stride=s, pad=p, Lout=ref.shape[1])
    e = err_stats(got, ref)
    report(f"conv_mfma{4 if mfma4 else 2}/convT_k12s6", **e)
    assert e["rel_max"] < 6e-3
    # fused AdaIN + Snake input and output statistics
    L, Cn, K, d = 2600, 128, 7, 3
    lens = [2600, 2049]
    x = _bf(rng.standard_normal((B, L, Cn)).astype(np.float32) * 2 + 0.5)
    for bb, n in enumerate(lens):
        x[bb, n:] = 0
    w = _bf((rng.standard_normal((Cn, K, Cn)) / math.sqrt(K * Cn)).astype(np.float32))
    bias = rng.standard_normal(Cn).astype(np.float32)
    res = _bf(rng.standard_normal((B, L, Cn)).astype(np.float32))
    A = (rng.standard_normal((B, Cn)) * 0.5 + 1).astype(np.float32)
    Bv = (rng.standard_normal((B, Cn)) * 0.3).astype(np.float32)
    alpha = rng.uniform(0.5, 1.5, Cn).astype(np.float32)
    wp, CinP, CoutP = pack_w_bf16(w)
    bp = np.zeros(CoutP, np.float32)
    bp[:Cn] = bias
    xd, wd, bd, rd = dev(x, torch.bfloat16), wp.cuda().contiguous(), dev(bp), dev(res, torch.bfloat16)
    ad, bvd, ald = dev(A), dev(Bv), dev(alpha)
    lend = dev(np.asarray(lens, np.int32), torch.int32)
    out = torch.full((B, L, Cn), 7.0, device="cuda", dtype=torch.bfloat16)
    part = torch.zeros((B, (L + 127) // 128, 2, Cn), device="cuda")
    nt = C.c_int(0)
    with _wfrag(lib, wd):
        rc = lib.kk_op_conv1d_bf16_fused(stream(), B, P(xd), Cn, L, P(lend), P(wd), CinP, CoutP, P(bd), Cn, Cn, K, 9, d, P(ad), P(bvd), Cn,
                                         _lib.ACT_SNAKE, 0.0, P(ald), P(rd), Cn, 1.0, P(out), Cn, P(part), C.byref(nt))
        assert rc == 0, lib.kk_last_error()
        torch.cuda.synchronize()
    got = out.float().cpu().numpy()
    pt = part.cpu().numpy().reshape(-1)[: B * nt.value * 2 * Cn].reshape(B, nt.value, 2, Cn)  # pitch = the kernel's tile count
    for bb, n in enumerate(lens):
        y = x[bb, :n] * A[bb][None, :] + Bv[bb][None, :]
        y = y + (1.0 / alpha)[None, :] * np.sin(alpha[None, :] * y) ** 2
        y = _bf(y.astype(np.float32))
        base = F.conv1d(torch.tensor(y)[None].transpose(1, 2), torch.tensor(w).permute(0, 2, 1), torch.tensor(bias), 1, 9, d).transpose(1, 2)[0]
        ref = (base + torch.tensor(res[bb, :n])).numpy()
        e = err_stats(got[bb, :n], ref)
        report(f"conv_mfma{4 if mfma4 else 2}/fused/b{bb}", **e)
        assert e["rel_max"] < 1.5e-2
        assert np.all(got[bb, n:] == 0)
        # (statistics of the fp32 values before the bf16 rounding: see test_conv_mfma_fused_adain_snake_and_stats)
        np.testing.assert_allclose(pt[bb, :, 0].sum(0), ref.astype(np.float64).sum(0), rtol=1e-3, atol=5e-2)
        np.testing.assert_allclose(pt[bb, :, 1].sum(0), (ref.astype(np.float64) ** 2).sum(0), rtol=1e-3)
        g64 = got[bb, :n].astype(np.float64)
        noise = 2.0 ** -9 / math.sqrt(3.0) * np.sqrt((g64 ** 2).sum(0))
        assert np.all(np.abs(pt[bb, :, 0].sum(0) - g64.sum(0)) <= 5 * noise + 5e-2)
        np.testing.assert_allclose(pt[bb, :, 1].sum(0), (g64 ** 2).sum(0), rtol=3e-3)


@pytest.mark.parametrize("B,rows,K,N,act,padded", [(1, 130, 768, 2304, 0, False), (3, 37, 128, 768, 0, True), (2, 130, 2048, 768, 2, False), (9, 130, 768, 512, 2, False)])
def test_linear_rows_streaming_kernel(lib, B, rows, K, N, act, padded):
    """kk_linear_rows.hip on its own (the text side's plain Linears while few rows are in flight): bf16 x / W, fp32 accumulation, bias, exact-erf GELU,
    zeros past an utterance's length; the flat form (dense items: one row axis across utterances, all three wave shapes by row count) and the per-item form
    (item pitch larger than rows * ld), against torch in fp32 on the same bf16 operands (one bf16 rounding of the result)."""
    from mlx_audio_amd import _lib

    rng = np.random.default_rng(B * 1000 + rows + K + N)
    ldx, ldo = K + (8 if padded else 0), N + (16 if padded else 0)
    xbs, obs = rows * ldx + (64 if padded else 0), rows * ldo + (128 if padded else 0)
    x = torch.zeros((B, xbs), dtype=torch.bfloat16)
    xv = torch.tensor(rng.standard_normal((B, rows, K)).astype(np.float32)).to(torch.bfloat16)
    x[:, : rows * ldx].view(B, rows, ldx)[:, :, :K] = xv
    w = torch.tensor((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)).to(torch.bfloat16)
    bias = torch.tensor(rng.standard_normal(N).astype(np.float32))
    lens = torch.tensor(rng.integers(1, rows + 1, B).astype(np.int32))
    lens[0] = rows
    xd, wd, bd, ld = x.cuda(), w.cuda(), bias.cuda(), lens.cuda()
    scratch = torch.empty(((N + 15) // 16 * 16) * K, dtype=torch.bfloat16, device="cuda")
    out = torch.full((B, obs), 7.0, dtype=torch.bfloat16, device="cuda")
    rc = lib.kk_op_linear_rows(stream(), B, P(xd), xbs, ldx, rows, P(ld), P(wd), N, K, P(bd), act, P(scratch), P(out), obs, ldo)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    got = out.cpu()[:, : rows * ldo].view(B, rows, ldo)[:, :, :N].float()
    ref = xv.float() @ w.float().T + bias
    if act == 2:
        ref = torch.nn.functional.gelu(ref)
    for b in range(B):
        ref[b, int(lens[b]) :] = 0
    err = (got - ref).abs()
    tol = 2.0**-8 * ref.abs() + 1e-3  # one bf16 rounding + fp32 summation order
    assert bool((err <= tol).all()), float((err - tol).max())
    assert bool((got[0, rows - 1] != 0).any())
    if padded:  # nothing outside the [rows][N] block of an item is written
        o = out.cpu()
        assert bool((o[:, rows * ldo :] == 7.0).all()) and bool((o[:, : rows * ldo].view(B, rows, ldo)[:, :, N:] == 7.0).all())
