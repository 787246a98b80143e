"""CPU checks of the Mimi decode oracle: the reference's only known answer (shapes, mlx_audio/codec/tests/test_mimi.py:9-19)
and the structural properties the restatement must have."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import mimi_oracle as M  # noqa: E402
import mlx_audio_amd.params as P  # noqa: E402


def test_reference_shape_known_answer():
    """test_mimi.py: codes [1, 32, 63] decode to pcm [1, 1, 120960] (1920 samples per 12.5 Hz frame)."""
    cfg = P.mimi_config(32)
    w = P.mimi_synth_checkpoint(cfg, 0)
    assert abs(sum(int(np.prod(s)) for s in P.mimi_param_inventory(cfg).values()) - 57.0e6) < 0.1e6
    codes = np.zeros((1, 32, 63), np.int64)
    pcm = M.MimiOracle(w, cfg).decode(codes)
    assert pcm.shape == (1, 1, 120960) and np.isfinite(pcm).all()


def test_decode_is_batch_independent_and_not_causal_in_the_transformer():
    cfg = P.mimi_tiny_config()
    w = P.mimi_synth_checkpoint(cfg, 1)
    orc = M.MimiOracle(w, cfg)
    rng = np.random.default_rng(0)
    codes = rng.integers(0, cfg["bins"], (3, cfg["nq"], 9))
    pcm, inter = orc.decode(codes, return_inter=True)
    assert pcm.shape == (3, 1, 1920 * 9)
    for b in range(3):
        np.testing.assert_allclose(orc.decode(codes[b : b + 1])[0], pcm[b], rtol=1e-5, atol=1e-5)
    # everything but the transformer is causal: changing the LAST frame's codes must not move the quantizer / upsample output
    # of earlier frames, but (mask=None, transformer.py:171) it does move the transformer output at t = 0
    c2 = codes.copy()
    c2[:, :, -1] = (c2[:, :, -1] + 1) % cfg["bins"]
    _, i2 = orc.decode(c2, return_inter=True)
    np.testing.assert_array_equal(inter["upsampled"][..., :16], i2["upsampled"][..., :16])
    assert np.abs(inter["transformer"][..., 0] - i2["transformer"][..., 0]).max() > 1e-6
    # never-used code-book entries (cluster_usage = 0) decode through the 1e-5 floor (quantization.py:25-28)
    c3 = np.zeros_like(codes)
    assert np.isfinite(orc.decode(c3)).all()


def test_reference_encode_shape_known_answer_and_round_trip_lengths():
    """test_mimi.py: audio [1, 1, 120000] encodes to codes [1, 32, 63]; decoding them gives [1, 1, 120960]."""
    cfg = P.mimi_config(32)
    w = P.mimi_synth_checkpoint(cfg, 0, encode=True)
    orc = M.MimiOracle(w, cfg)
    codes = orc.encode(np.zeros((1, 1, 120_000), np.float32))
    assert codes.shape == (1, 32, 63) and codes.min() >= 0 and codes.max() < 2048
    assert orc.decode(codes).shape == (1, 1, 120_960)


def test_encode_finds_the_nearest_code_and_is_causal_before_the_transformer():
    cfg = P.mimi_tiny_config()
    w = P.mimi_synth_checkpoint(cfg, 2, encode=True)
    orc = M.MimiOracle(w, cfg)
    rng = np.random.default_rng(3)
    pcm = (0.3 * rng.standard_normal((2, 1, 1920 * 4 + 333))).astype(np.float32)
    trace = []
    codes, inter = orc.encode(pcm, trace=trace, return_inter=True)
    assert codes.shape == (2, cfg["nq"], 5)  # ceil chain: 8013 -> 2004 -> 401 -> 67 -> 9 -> 5
    for which, i, dist, idx in trace:  # the chosen entry is the exact minimiser of the stated distance
        np.testing.assert_array_equal(idx, dist.argmin(-1))
    p2 = pcm.copy()
    p2[..., -200:] += 0.5  # the SEANet encoder is causal: early frames do not move
    _, i2 = orc.encode(p2, return_inter=True)
    np.testing.assert_array_equal(inter["seanet"][..., :6], i2["seanet"][..., :6])


def test_streaming_restatement_properties():
    """MimiStreamOracle (the reference's explicit per-module state, conv.py:265-351): the causal resampler and SEANet give, frame by frame,
    exactly what the offline causal convolutions give at those positions -- the property the GPU stream is built on -- while the
    transformer (cache, no mask) makes decode_step differ from decode(); one frame in, 1920 * prod(ratios) / 1920 samples out."""
    import torch

    cfg = P.mimi_tiny_config()
    w = P.mimi_synth_checkpoint(cfg, 0)
    rng = np.random.default_rng(0)
    codes = rng.integers(0, cfg["bins"], (2, cfg["nq"], 9))
    off = M.MimiOracle(w, cfg)
    pcm_off, inter = off.decode(codes, return_inter=True)
    st = M.MimiStreamOracle(w, cfg)
    outs, ups, trs = [], [], []
    for i in range(codes.shape[-1]):
        p, it = st.decode_step(codes[:, :, i : i + 1], return_inter=True)
        assert p.shape == (2, 1, pcm_off.shape[-1] // codes.shape[-1])
        outs.append(p); ups.append(it["upsampled"]); trs.append(it["transformer"])
    pcm_st = np.concatenate(outs, -1)
    np.testing.assert_allclose(np.concatenate(ups, -1), inter["upsampled"], atol=1e-6)
    with torch.no_grad():
        again = off.seanet_decoder(torch.tensor(np.concatenate(trs, -1))).numpy()
    np.testing.assert_allclose(pcm_st, again, atol=5e-6 * max(1.0, float(np.abs(again).max())))
    assert np.abs(pcm_st - pcm_off).max() > 1e-3  # the streaming transformer sees the past only
    # frame 0 has no past: its two positions see each other in both forms, so the first frame agrees... only if the sequence is one frame long
    one = M.MimiStreamOracle(w, cfg).decode_step(codes[:, :, :1])
    np.testing.assert_allclose(one, off.decode(codes[:, :, :1]), atol=5e-6 * max(1.0, float(np.abs(one).max())))
    # reset gives the same stream again; a context shorter than the history changes later frames only
    st.reset()
    np.testing.assert_array_equal(st.decode_frames(codes), pcm_st)
    short = M.MimiStreamOracle(w, cfg, context=4).decode_frames(codes)
    np.testing.assert_array_equal(short[..., : 3 * one.shape[-1]], pcm_st[..., : 3 * one.shape[-1]])
    assert np.abs(short - pcm_st).max() > 0


def test_streaming_encode_restatement_properties():
    """MimiStreamOracle.encode_step (Mimi.encode_step, mimi.py:156-161, through StreamableConv1d.step, conv.py:265-293): the causal
    SEANet encoder and the 'edge' resampler give, chunk by chunk, what the offline causal convolutions give on whole frames (for the
    convolutions chunk boundaries do not matter, not even ragged ones: every module holds back its partial stride), while the encoder transformer with its
    cache sees the past only, so codes differ from encode() beyond the first chunk."""
    import torch

    cfg = P.mimi_tiny_config()
    w = P.mimi_synth_checkpoint(cfg, 0, encode=True)
    spf = int(np.prod(cfg["ratios"])) * cfg["upsample_stride"]
    rng = np.random.default_rng(1)
    pcm = (rng.standard_normal((2, 1, 6 * spf)) * 0.3).astype(np.float32)
    off = M.MimiOracle(w, cfg)
    _, inter = off.encode(pcm, return_inter=True)
    st = M.MimiStreamOracle(w, cfg)
    sea, trs, cds = [], [], []
    for i in range(6):
        c, it = st.encode_step(pcm[..., i * spf : (i + 1) * spf], return_inter=True)
        assert c.shape == (2, cfg["nq"], 1)
        sea.append(it["seanet"]); trs.append(it["transformer"]); cds.append(c)
    np.testing.assert_allclose(np.concatenate(sea, -1), inter["seanet"], atol=2e-6 * max(1.0, float(np.abs(inter["seanet"]).max())))
    # the resampler + quantiser of the streamed transformer rows, run offline, give the streamed codes
    with torch.no_grad():
        xd = off.strided_causal_conv(torch.tensor(np.concatenate(trs, -1)), "downsample.conv", cfg["upsample_stride"], pad_mode="edge")
        again = off.rvq_encode("rvq_first", xd, 1)
        if cfg["nq"] > 1:
            again = np.concatenate([again, off.rvq_encode("rvq_rest", xd, cfg["nq"] - 1)], axis=1)
    assert (again == np.concatenate(cds, -1)).mean() > 0.98  # near-tie flips of the argmin aside
    # ragged chunks (not whole frames): the same SEANet rows and the same NUMBER of frames come out, later; the codes may differ, because
    # the positions one transformer step holds see each other (no mask) and the chunking decides which those are
    st.reset()
    cuts = [0, 7, spf + 3, 2 * spf, 4 * spf - 1, 6 * spf]
    rag = [st.encode_step(pcm[..., a:b], return_inter=True) for a, b in zip(cuts[:-1], cuts[1:])]
    assert np.concatenate([c for c, _ in rag], -1).shape == (2, cfg["nq"], 6)
    np.testing.assert_allclose(np.concatenate([it["seanet"] for _, it in rag], -1), inter["seanet"], atol=2e-6 * max(1.0, float(np.abs(inter["seanet"]).max())))
    # one chunk holding everything: the transformer sees all positions at once, as encode() does
    st.reset()
    whole = st.encode_step(pcm)
    assert (whole == off.encode(pcm)).mean() > 0.98
