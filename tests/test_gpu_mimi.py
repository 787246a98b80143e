"""Mimi.decode through the C ABI (kk_mimi_*) against the CPU oracle (oracle/mimi_oracle.py) on identical synthetic weights and
codes.  fp32 path: every stage within 2e-4 of its max, pcm within 1e-3 (BASELINE north_star tolerance)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import mimi_oracle as M  # noqa: E402
import mlx_audio_amd.params as P  # noqa: E402
from _util import err_stats, report  # noqa: E402

pytestmark = pytest.mark.gpu

STAGES = ["quantized", "upsampled", "transformer", "layer0", "layer1", "layer2", "layer3"]


def _pair(cfg, seed, B, Nf, tag):
    from mlx_audio_amd.mimi import Mimi, MimiConfig

    w = P.mimi_synth_checkpoint(cfg, seed)
    rng = np.random.default_rng(seed + 100)
    codes = rng.integers(0, cfg["bins"], (B, cfg["nq"], Nf))
    codes[0, :, 0] = 0  # a never-used code-book entry (cluster_usage = 0 -> the 1e-5 floor)
    ref, inter = M.MimiOracle(w, cfg).decode(codes, return_inter=True)
    model = Mimi(MimiConfig.from_dict(cfg), w)
    pcm = model.decode(torch.tensor(codes))
    torch.cuda.synchronize()
    got = pcm.cpu().numpy()
    assert got.shape == ref.shape == (B, 1, 1920 * Nf)
    worst = {}
    for name in STAGES:
        g = model.debug_fetch(name).cpu().numpy()  # [B][rows][C]
        r = np.transpose(inter[name], (0, 2, 1))
        e = err_stats(g, r)
        report(f"mimi/{tag}/{name}", **e)
        worst[name] = e["rel_max"]
    e = err_stats(got, ref)
    report(f"mimi/{tag}/pcm", **e)
    return e, worst, model, codes, got


def test_mimi_tiny_decode_matches_oracle():
    e, worst, model, codes, got = _pair(P.mimi_tiny_config(), 3, 3, 11, "tiny")
    for k, v in worst.items():
        assert v < 2e-4, (k, v)
    assert e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), e
    # batch items are independent: item 1 alone gives the same bits
    one = model.decode(torch.tensor(codes[1:2])).cpu().numpy()
    np.testing.assert_array_equal(one[0], got[1])


def test_mimi_202407_decode_matches_oracle():
    """The real configuration (57 M decode-side parameters), 20 frames = 1.6 s of audio, batch 2."""
    e, worst, _, _, _ = _pair(P.mimi_config(32), 4, 2, 20, "202407")
    for k, v in worst.items():
        assert v < 2e-4, (k, v)
    assert e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), e


@pytest.mark.parametrize("which", ["tiny", "202407"])
def test_mimi_bf16_mode_tracks_the_fp32_oracle(which):
    """compute_dtype bfloat16: bf16 activations and MFMA convolutions (fp32 accumulation).  Measured against the fp32 oracle the pcm
    sits at the 1 % level of its peak (16 residual updates + a 4-stage vocoder in bf16); asserted at 6 %."""
    from mlx_audio_amd.mimi import Mimi, MimiConfig

    cfg = P.mimi_tiny_config() if which == "tiny" else P.mimi_config(32)
    w = P.mimi_synth_checkpoint(cfg, 5)
    rng = np.random.default_rng(55)
    B, Nf = 2, 16
    codes = rng.integers(0, cfg["bins"], (B, cfg["nq"], Nf))
    ref, inter = M.MimiOracle(w, cfg).decode(codes, return_inter=True)
    model = Mimi(MimiConfig.from_dict(cfg), w, compute_dtype="bfloat16")
    got = model.decode(torch.tensor(codes))
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    assert got.shape == ref.shape and np.isfinite(got).all()
    for name in STAGES:
        g = model.debug_fetch(name).cpu().numpy()
        e = err_stats(g, np.transpose(inter[name], (0, 2, 1)))
        report(f"mimi_bf16/{which}/{name}", **e)
        assert e["rms_rel"] < 3e-2, (name, e)
    e = err_stats(got, ref)
    report(f"mimi_bf16/{which}/pcm", **e)
    assert e["rms_rel"] < 3e-2 and e["rel_max"] < 6e-2, e
    one = model.decode(torch.tensor(codes[1:2])).cpu().numpy()  # batch items are independent in bf16 mode too
    np.testing.assert_array_equal(one[0], got[1])


def test_mimi_reference_shape_known_answer_on_gpu():
    """mlx_audio/codec/tests/test_mimi.py: codes [1, 32, 63] -> pcm [1, 1, 120960]."""
    from mlx_audio_amd.mimi import Mimi, mimi_202407

    cfg = P.mimi_config(32)
    model = Mimi(mimi_202407(32), P.mimi_synth_checkpoint(cfg, 0))
    pcm = model.decode(torch.zeros((1, 32, 63), dtype=torch.int64))
    torch.cuda.synchronize()
    assert tuple(pcm.shape) == (1, 1, 120_960) and bool(torch.isfinite(pcm).all())
    assert model.sample_rate == 24000 and model.frame_rate == 12.5
    with pytest.raises(ValueError):
        model.decode(torch.zeros((1, 31, 5), dtype=torch.int64))


@pytest.mark.parametrize("which", ["tiny", "202407"])
def test_mimi_encode_matches_oracle(which):
    """Mimi.encode (row C5): the continuous stages within 2e-4; the codes equal to the oracle's, except where the two best
    code-book entries are a genuine near-tie at the FIRST differing code book of a frame (distance gap < 1e-4 of the spread) --
    after such a flip the residual differs and the later code books of that frame are not comparable."""
    from mlx_audio_amd.mimi import Mimi, MimiConfig

    cfg = P.mimi_tiny_config() if which == "tiny" else P.mimi_config(32)
    w = P.mimi_synth_checkpoint(cfg, 6, encode=True)
    rng = np.random.default_rng(66)
    B, N = (3, 1920 * 6 + 777) if which == "tiny" else (2, 1920 * 8 + 100)
    pcm = (0.3 * rng.standard_normal((B, 1, N))).astype(np.float32)
    trace = []
    ref, inter = M.MimiOracle(w, cfg).encode(pcm, trace=trace, return_inter=True)
    model = Mimi(MimiConfig.from_dict(cfg), w)
    codes = model.encode(torch.tensor(pcm))
    torch.cuda.synchronize()
    got = codes.cpu().numpy()
    assert got.shape == ref.shape
    for name in ("seanet", "transformer", "downsampled"):
        g = model.debug_fetch(name).cpu().numpy()
        e = err_stats(g, np.transpose(inter[name], (0, 2, 1)))
        report(f"mimi_encode/{which}/{name}", **e)
        assert e["rel_max"] < 2e-4, (name, e)
    nq, Nf = ref.shape[1], ref.shape[2]
    frames_equal = 0
    for b in range(B):
        for t in range(Nf):
            diff = np.nonzero(got[b, :, t] != ref[b, :, t])[0]
            if diff.size == 0:
                frames_equal += 1
                continue
            i = int(diff[0])  # first differing code book: same residual on both sides up to round-off
            dist = trace[i][2][b, t]
            gap = abs(float(dist[got[b, i, t]]) - float(dist[ref[b, i, t]]))
            assert gap < 1e-4 * float(dist.max() - dist.min()), (b, t, i, gap)
    report(f"mimi_encode/{which}/frames_with_identical_codes", value=frames_equal, total=B * Nf, max_abs=0.0, ref_max=1.0, rel_max=0.0, rms_rel=0.0)
    assert frames_equal >= 0.9 * B * Nf
    # decode(encode(x)) has the length of whole frames
    out = model.decode(codes)
    assert tuple(out.shape) == (B, 1, 1920 * Nf)


def test_mimi_golden_fixture_without_oracle():
    """tests/golden/mimi_tiny_case.npz (made by tests/golden/make_golden_codec.py): committed inputs and expected outputs; no oracle code runs."""
    from mlx_audio_amd.mimi import Mimi, MimiConfig

    g = np.load(os.path.join(ROOT, "tests", "golden", "mimi_tiny_case.npz"))
    cfg = P.mimi_tiny_config()
    model = Mimi(MimiConfig.from_dict(cfg), P.mimi_synth_checkpoint(cfg, int(g["weights_seed"]), encode=True))
    pcm = model.decode(torch.tensor(g["codes"])).cpu().numpy()
    e = err_stats(pcm, g["pcm_out"])
    report("mimi/golden/pcm", **e)
    assert e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), e
    codes = model.encode(torch.tensor(g["pcm_in"])).cpu().numpy()
    assert (codes == g["codes_out"]).mean() > 0.97  # identical up to near-tie flips of the argmin
    # the streaming decode of the same codes (Mimi.decode_step frame by frame, MimiStreamingDecoder)
    from mlx_audio_amd.mimi import MimiStreamingDecoder

    st = MimiStreamingDecoder(model).decode_frames(torch.tensor(g["codes"])).cpu().numpy()
    es = err_stats(st, g["pcm_stream"])
    report("mimi/golden/pcm_stream", **es)
    assert es["max_abs"] <= 1e-3 * max(1.0, es["ref_max"]), es


@pytest.mark.parametrize("which", ["tiny", "mimi_202407"])
def test_mimi_streaming_decode_matches_stream_oracle(which):
    """Mimi.decode_step / MimiStreamingDecoder (mimi.py:163-168,264-306) through kk_mimi_decode_step against MimiStreamOracle (the
    reference's explicit conv / cache state): every frame's pcm within 1e-3 (fp32: measured ~1e-6) with per-layer carried rows (more
    frames than any module's look-back), more cached positions than the tiny context (the key range slides), reset, batch independence,
    and several frames per step."""
    from mlx_audio_amd.mimi import Mimi, MimiConfig, MimiStreamingDecoder

    cfg = P.mimi_tiny_config() if which == "tiny" else P.mimi_config(32)
    w = P.mimi_synth_checkpoint(cfg, 5)
    rng = np.random.default_rng(9)
    B, Nf = (2, 21) if which == "tiny" else (2, 12)
    codes = rng.integers(0, cfg["bins"], (B, cfg["nq"], Nf))
    orc = M.MimiStreamOracle(w, cfg)
    model = Mimi(MimiConfig.from_dict(cfg), w)
    spf = 1920 if which != "tiny" else int(np.prod(cfg["ratios"])) * cfg["upsample_stride"]
    worst = 0.0
    outs = []
    for i in range(Nf):
        ref, inter = orc.decode_step(codes[:, :, i : i + 1], return_inter=True)
        got = model.decode_step(torch.tensor(codes[:, :, i : i + 1]))
        torch.cuda.synchronize()
        assert tuple(got.shape) == (B, 1, spf) == ref.shape
        for name in ("upsampled", "transformer"):
            e = err_stats(model.debug_fetch(name).cpu().numpy(), np.transpose(inter[name], (0, 2, 1)))
            assert e["rel_max"] < 2e-4, (i, name, e)
        e = err_stats(got.cpu().numpy(), ref)
        worst = max(worst, e["max_abs"] / max(1.0, e["ref_max"]))
        assert e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), (i, e)
        outs.append(got.cpu().numpy())
    report(f"mimi/stream/{which}", worst_rel=worst, frames=Nf)
    # MimiStreamingDecoder: reset, then the same frames again in one call -> the same bits; item 1 alone -> the same bits
    dec = MimiStreamingDecoder(model)
    again = dec.decode_frames(torch.tensor(codes)).cpu().numpy()
    np.testing.assert_array_equal(again, np.concatenate(outs, -1))
    dec.reset()
    alone = dec.decode_frames(torch.tensor(codes[1:2])).cpu().numpy()
    np.testing.assert_array_equal(alone[0], again[1])
    # streaming is NOT decode(): the offline transformer sees the whole sequence (no mask), the stream only the past
    off = model.decode(torch.tensor(codes)).cpu().numpy()
    assert np.abs(off - again).max() > 1e-4
    # three frames per step: the reference's modules take any length; the positions of one step see each other, so this is its own stream
    dec.reset()
    orc3 = M.MimiStreamOracle(w, cfg)
    for i in range(0, Nf - Nf % 3, 3):
        ref3 = orc3.decode_step(codes[:, :, i : i + 3])
        got3 = model.decode_step(torch.tensor(codes[:, :, i : i + 3])).cpu().numpy()
        e = err_stats(got3, ref3)
        assert got3.shape == ref3.shape and e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), (i, e)
    if which == "tiny":  # a context shorter than the history: the key range slides (transformer.py:94-98)
        model.decode_step(torch.tensor(codes[:, :, :1]))  # (continues the three-frame stream with a one-frame step; reset below)
        from mlx_audio_amd import _lib

        dec.reset()
        _lib.check(model.lib.kk_mimi_stream_set_context(model._streams["dec"]["h"], 6), "set_context")
        short = dec.decode_frames(torch.tensor(codes)).cpu().numpy()
        ref_short = M.MimiStreamOracle(w, cfg, context=6).decode_frames(codes)
        e = err_stats(short, ref_short)
        report("mimi/stream/tiny_context6", **e)
        assert e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]) and np.abs(short - again).max() > 1e-4


@pytest.mark.parametrize("which", ["tiny", "mimi_202407"])
def test_mimi_streaming_encode_matches_stream_oracle(which):
    """Mimi.encode_step (mimi.py:156-161) through kk_mimi_encode_step against MimiStreamOracle.encode_step: the SEANet encoder rows and
    the transformer rows of every chunk within 2e-4 (fp32), codes identical up to near-tie flips of the argmin; one and two frames per
    step; reset; a partial frame is refused."""
    from mlx_audio_amd.mimi import Mimi, MimiConfig

    cfg = P.mimi_tiny_config() if which == "tiny" else P.mimi_config(32)
    w = P.mimi_synth_checkpoint(cfg, 5, encode=True)
    spf = 1920 if which != "tiny" else int(np.prod(cfg["ratios"])) * cfg["upsample_stride"]
    B, Nf = (2, 12) if which == "tiny" else (2, 6)
    rng = np.random.default_rng(11)
    pcm = (rng.standard_normal((B, 1, Nf * spf)) * 0.3).astype(np.float32)
    model = Mimi(MimiConfig.from_dict(cfg), w)
    for F in (1, 2):
        orc = M.MimiStreamOracle(w, cfg)
        model.reset_stream()
        got_all, ref_all = [], []
        for i in range(0, Nf, F):
            ref, inter = orc.encode_step(pcm[..., i * spf : (i + F) * spf], return_inter=True)
            got = model.encode_step(torch.tensor(pcm[..., i * spf : (i + F) * spf]))
            torch.cuda.synchronize()
            assert tuple(got.shape) == (B, cfg["nq"], F) == ref.shape
            for name in ("seanet", "transformer", "downsampled"):
                e = err_stats(model.debug_fetch(name).cpu().numpy(), np.transpose(inter[name], (0, 2, 1)))
                assert e["rel_max"] < 2e-4, (F, i, name, e)
            got_all.append(got.cpu().numpy()); ref_all.append(ref)
        got_all, ref_all = np.concatenate(got_all, -1), np.concatenate(ref_all, -1)
        agree = float((got_all == ref_all).mean())
        report(f"mimi/stream_encode/{which}/F{F}", agree=agree)
        assert agree > 0.97
        if F == 1:
            first = got_all
    model.reset_stream()
    again = np.concatenate([model.encode_step(torch.tensor(pcm[..., i * spf : (i + 2) * spf])).cpu().numpy() for i in range(0, Nf, 2)], -1)
    np.testing.assert_array_equal(again, got_all)
    assert first.shape == again.shape
    with pytest.raises(ValueError):
        model.encode_step(torch.tensor(pcm[..., : spf + 5]))


def test_mimi_stream_continues_across_step_sizes():
    """The reference's decode_step / encode_step accept any number of frames per call and CONTINUE the conv / KV state (mimi.py:156-168,
    conv.py:265-351): a stream fed 2, 2, 1 (then 3, 1) frames is ONE stream -- every step against MimiStreamOracle run with the same step
    sizes, and different from a stream restarted at the size change (the final, shorter chunk used to be coded as a new
    stream, silently).  Growing past the largest step of an open stream is an error, not a restart; max_chunk= sizes a stream up front."""
    from mlx_audio_amd.mimi import Mimi, MimiConfig

    cfg = P.mimi_tiny_config()
    w = P.mimi_synth_checkpoint(cfg, 5, encode=True)
    spf = int(np.prod(cfg["ratios"])) * cfg["upsample_stride"]
    rng = np.random.default_rng(21)
    B, steps = 2, [2, 2, 1, 2, 1]
    Nf = sum(steps)
    codes = rng.integers(0, cfg["bins"], (B, cfg["nq"], Nf))
    model = Mimi(MimiConfig.from_dict(cfg), w)
    orc = M.MimiStreamOracle(w, cfg)
    i, outs = 0, []
    for F in steps:
        ref = orc.decode_step(codes[:, :, i : i + F])
        got = model.decode_step(torch.tensor(codes[:, :, i : i + F])).cpu().numpy()
        e = err_stats(got, ref)
        assert got.shape == ref.shape == (B, 1, spf * F) and e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), (i, F, e)
        outs.append(got)
        i += F
    assert int(model.lib.kk_mimi_stream_frames(model._streams["dec"]["h"])) == Nf
    # restarting at the size change (what round 2 did) is a different signal: the test above is not vacuous
    model.reset_stream()
    model.decode_step(torch.tensor(codes[:, :, 0:2])); model.decode_step(torch.tensor(codes[:, :, 2:4]))
    model.reset_stream()
    restarted = model.decode_step(torch.tensor(codes[:, :, 4:5])).cpu().numpy()
    assert np.abs(restarted - outs[2]).max() > 1e-4
    # a step larger than the stream was opened for, mid-stream: refused loudly
    with pytest.raises(ValueError, match="already consumed frames"):
        model.decode_step(torch.tensor(codes[:, :, 0:3]))
    # max_chunk= on the first call: 1, 3, 2 frames on one stream
    model.reset_stream()
    orc = M.MimiStreamOracle(w, cfg)
    i = 0
    for F in (1, 3, 2):
        ref = orc.decode_step(codes[:, :, i : i + F])
        got = model.decode_step(torch.tensor(codes[:, :, i : i + F]), max_chunk=3).cpu().numpy()
        e = err_stats(got, ref)
        assert e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), (i, F, e)
        i += F
    # encode side: 2, 2, 1 frames of pcm
    pcm = (rng.standard_normal((B, 1, 5 * spf)) * 0.3).astype(np.float32)
    orc = M.MimiStreamOracle(w, cfg)
    i, agree = 0, []
    for F in (2, 2, 1):
        ref, inter = orc.encode_step(pcm[..., i * spf : (i + F) * spf], return_inter=True)
        got = model.encode_step(torch.tensor(pcm[..., i * spf : (i + F) * spf]))
        torch.cuda.synchronize()
        for name in ("seanet", "transformer", "downsampled"):
            e = err_stats(model.debug_fetch(name).cpu().numpy(), np.transpose(inter[name], (0, 2, 1)))
            assert e["rel_max"] < 2e-4, (F, i, name, e)
        agree.append((got.cpu().numpy() == ref).mean())
        i += F
    assert min(agree) > 0.9, agree
