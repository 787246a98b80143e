"""The CPU oracles against INDEPENDENT implementations (transformers Albert / Llama / Csm / Mimi, torch.nn.LSTM, torch.stft / istft,
torch weight_norm convs / InstanceNorm1d / LayerNorm / F.interpolate).

The fixtures under tests/golden/independent/ were written in the build container by tests/golden/make_golden_independent.py (which imports
no oracle); each holds seeded inputs, the independent implementation's outputs and the SHA-256 of the seeded weights, which are rebuilt here
through mlx-audio_amd/params.py.  Nothing here needs `transformers` or a GPU.

Bar: <= 1e-5 relative to the tensor's max (fp32, different summation orders) unless a test states a cited reason for more.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

import csm_oracle as C
import kokoro_oracle as O
import mimi_oracle as M
import mlx_audio_amd.params as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "independent")
TOL = 1e-5


def load(name):
    p = os.path.join(GOLD, name + ".npz")
    if not os.path.exists(p):
        pytest.fail(f"missing fixture {p}: run tests/golden/make_golden_independent.py in the build container")
    return np.load(p)


def wdigest(w: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(w):
        a = np.ascontiguousarray(np.asarray(w[k], np.float32))
        h.update(k.encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def rel(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    return float(np.abs(got - ref).max() / max(1e-30, np.abs(ref).max()))


def albert_cfg():
    cfg = P.tiny_config()
    cfg["plbert"] = dict(hidden_size=96, num_attention_heads=4, intermediate_size=160, max_position_embeddings=64, num_hidden_layers=5, dropout=0.1)
    return cfg


# ---------------------------------------------------------------------------------------------------------------------
# Kokoro
# ---------------------------------------------------------------------------------------------------------------------
def test_albert_matches_transformers_albert_model():
    """K2 (modules.py:438-649): shared-layer Albert, exact-erf GELU, LayerNorm eps 1e-12, additive mask (all ones at batch 1)."""
    g = load("albert")
    cfg = albert_cfg()
    w = P.synth_checkpoint(cfg, int(g["seed"]))
    assert wdigest(w) == str(g["wsha"])
    with torch.no_grad():
        out = O.KokoroOracle(w, cfg).albert(torch.as_tensor(g["ids"])).numpy()
    assert rel(out, g["out"]) <= TOL


def test_lstm_matches_torch_nn_lstm():
    """K4 (modules.py:93-285) + K20 (the rename table `sanitize_lstm_weights`, kokoro.py:24-44, inverted by params.to_torch_layout):
    gate order i, f, g, o; bias_ih + bias_hh; the backward direction runs right to left and is concatenated after the forward one."""
    g = load("lstm")
    cfg = P.tiny_config()
    w = P.synth_checkpoint(cfg, int(g["seed"]))
    assert wdigest(w) == str(g["wsha"])
    orc = O.KokoroOracle(w, cfg)
    for name in ("text_encoder.lstm", "predictor.lstm", "predictor.shared"):
        k = name.replace(".", "__")
        with torch.no_grad():
            y = orc.lstm(torch.as_tensor(g[k + "__x"]), name).numpy()
        assert rel(y, g[k + "__y"]) <= TOL, name


def test_stft_matches_torch_stft():
    """K15 (utils.py:52-101): reflect pad n_fft // 2, symmetric Hann(20), hop 5, rfft -> 11 bins."""
    g = load("stft")
    for b in range(g["x"].shape[0]):
        X = O.stft(g["x"][b], 20, 5, 20).T  # [11, frames]
        assert rel(X.real, g["X_re"][b]) <= TOL and np.abs(X.imag - g["X_im"][b]).max() <= TOL * np.abs(g["X_re"][b]).max()


def test_istft_matches_torch_istft_up_to_the_reference_normalisation():
    """K17 (utils.py:104-158, istftnet.py:497-523).  torch.istft divides the overlap-add by sum(w^2); the reference divides by sum(w)
    (utils.py:143-150, `window_sum` accumulates the window, not its square).  reference = torch.istft * sum(w^2) / sum(w): the envelope is
    in the fixture (built with F.fold), 0.75 everywhere except the 5 first / last samples, where only 3 frames overlap."""
    g = load("stft")
    env = g["env"]
    assert np.allclose(env[5:-5], 0.75, atol=1e-6) and not np.allclose(env[:5], 0.75, atol=1e-3)
    spec = (g["spec_re"] + 1j * g["spec_im"]).astype(np.complex64)
    for b in range(spec.shape[0]):
        y = O.istft(spec[b], 5, 20)
        assert rel(y, g["y_torch"][b] * env) <= TOL
    # the whole head (exp / sin parametrisation included) from the raw 22-channel input
    cfg = P.tiny_config()
    orc = O.KokoroOracle(P.synth_checkpoint(cfg, 0), cfg)
    yh = orc.istft_head(g["head_in"])[:, 0]
    assert rel(yh, g["y_torch"] * env[None]) <= 2e-5  # float32 exp / sin / cos against float64 in the fixture


def _mlx_v(v_torch, transpose, groups):
    """torch weight_v layout -> the MLX-side layout the oracle holds (conv [O,I,K] -> [O,K,I]; the reference's `ups` keep the
    checkpoint's [in, out, K] -> (0,2,1) -> [in, K, out], istftnet.py:161-166)."""
    return np.ascontiguousarray(np.transpose(v_torch, (0, 2, 1)))


@pytest.mark.parametrize("tag,stride,pad,dil,groups,transpose", [
    ("conv_k3", 1, 1, 1, 1, False), ("conv_k7_d3", 1, 9, 3, 1, False), ("conv_k11_d5", 1, 25, 5, 1, False), ("conv_s2", 2, 1, 1, 1, False),
    ("ups_k20_s10", 10, 5, 1, 1, True), ("ups_k12_s6", 6, 3, 1, 1, True), ("pool_dw", 2, 1, 1, 10, True)])
def test_conv_weighted_matches_torch_weight_norm_convs(tag, stride, pad, dil, groups, transpose):
    """K10 / K12 / K16 (istftnet.py:53-170): g * v / ||v|| over (k, in); conv1d, the generator's transposed convs, the depth-wise pool.
    torch's weight_norm has no +1e-7 on the norm (istftnet.py:88 has): <= 1e-6 relative on top of round-off."""
    g = load("primitives")
    cfg = P.tiny_config()
    orc = O.KokoroOracle({"c.weight_g": g[tag + "__g"], "c.weight_v": _mlx_v(g[tag + "__v_torch"], transpose, groups), "c.bias": g[tag + "__b"]}, cfg)
    with torch.no_grad():
        y = orc.conv_weighted(torch.as_tensor(g[tag + "__x"]), "c", stride, pad, dil, groups, transpose).numpy()
    assert rel(y, g[tag + "__y"]) <= TOL


def test_norms_and_interpolation_match_torch():
    """K11 InstanceNorm1d / AdaIN1d (istftnet.py:216-338), K5 AdaLayerNorm (modules.py:71-90), K18 nearest interpolation
    (interpolate.py:72-80).  Linear interpolation with align_corners=False equals torch's except where the reference leaves the low index
    un-clamped (interpolate.py:96: a negative index wraps to the LAST sample) -- the first half-period of outputs."""
    g = load("primitives")
    cfg = P.tiny_config()
    orc = O.KokoroOracle({"a.fc.weight": g["adain__fc_w"], "a.fc.bias": g["adain__fc_b"], "l.fc.weight": g["adaln__fc_w"], "l.fc.bias": g["adaln__fc_b"]}, cfg)
    with torch.no_grad():
        assert rel(orc.instance_norm(torch.as_tensor(g["in__x"])).numpy(), g["in__y"]) <= TOL
        assert rel(orc.adain(torch.as_tensor(g["in__x"]), torch.as_tensor(g["adain__s"]), "a").numpy(), g["adain__y"]) <= TOL
        assert rel(orc.ada_layer_norm(torch.as_tensor(g["adaln__x"]), torch.as_tensor(g["adain__s"][:1]), "l").numpy(), g["adaln__y"]) <= TOL
    np.testing.assert_array_equal(O.interpolate(g["nearest__x"], scale_factor=300, mode="nearest"), g["nearest__y"])
    lin = O.interpolate(g["nearest__x"], scale_factor=np.float32(300), mode="linear")
    ref = g["linear_up__y"]
    assert rel(lin[..., 150:], ref[..., 150:]) <= 2e-5
    # the quirk: outputs 0..149 blend the LAST input sample in where torch clamps to the first
    x = g["nearest__x"]
    frac = (np.arange(150, dtype=np.float32) + 0.5) / 300 - 0.5 + 1.0
    np.testing.assert_allclose(lin[..., :150], x[..., -1:] * (1 - frac) + x[..., :1] * frac, rtol=2e-5, atol=2e-6)


# ---------------------------------------------------------------------------------------------------------------------
# Llama / CSM
# ---------------------------------------------------------------------------------------------------------------------
def csm_small_config():
    rope = dict(rope_theta=500000.0, rope_factor=32.0, rms_eps=1e-5)
    return dict(text_vocab_size=50, audio_vocab_size=19, audio_num_codebooks=4, max_seq_len=2048,
                backbone=dict(num_layers=3, num_heads=4, num_kv_heads=2, head_dim=32, hidden=64, intermediate=96, **rope),
                decoder=dict(num_layers=2, num_heads=2, num_kv_heads=1, head_dim=48, hidden=40, intermediate=56, **rope))


def test_llama_stack_matches_transformers_llama_model():
    """C2 + C3 (attention.py:10-195; mlx_lm's LlamaModel, absent from the reference tree): RMSNorm, GQA attention, llama3-scaled RoPE,
    SwiGLU, final norm, a prompt block followed by two single-token steps on the KV cache, and positions 1500.. where the frequency
    scaling (factor 32, low 1, high 4, old context 8192) is far from the identity."""
    g = load("llama")
    cfg = csm_small_config()
    w = P.csm_synth_checkpoint(cfg, int(g["seed"]))
    assert wdigest(w) == str(g["wsha"])
    a = cfg["backbone"]
    theta = C.llama3_theta(a["head_dim"], a["rope_theta"], a["rope_factor"])
    assert rel(theta, g["inv_freq"]) <= 1e-6
    with torch.no_grad():
        st = C.LlamaStack({k: np.asarray(v, np.float32) for k, v in w.items()}, "backbone", a)
        for xk, yk in (("x0", "y0"), ("x1", "y1"), ("x2", "y2")):
            assert rel(st(torch.as_tensor(g[xk])).numpy(), g[yk]) <= TOL, xk
        st.reset()
        st.offset = int(g["posf"][0, 0])
        assert rel(st(torch.as_tensor(g["xf"])).numpy(), g["yf"]) <= TOL


def test_csm_generate_frame_matches_transformers_csm():
    """C1 (sesame.py:349-415): [5 text positions | 4 audio frames] through the backbone (text rows embed through text_embeddings, audio rows
    as the sum of the 32 -> here 4 code-book embeddings with per-code-book offsets), codebook0_head, then the depth decoder over
    [projected last_h, embed(c0)], embed(c1) ... with audio_head[i-1]; greedy.  Logits <= 1e-5, every code equal."""
    g = load("csm")
    cfg = csm_small_config()
    w = P.csm_synth_checkpoint(cfg, int(g["seed"]))
    assert wdigest(w) == str(g["wsha"])
    orc = C.CsmOracle(w, cfg)
    text, audio = g["text"], g["audio"]
    B, ncb = text.shape[0], cfg["audio_num_codebooks"]
    tok = np.zeros((B, text.shape[1] + audio.shape[1], ncb + 1), np.int64)
    msk = np.zeros_like(tok)
    tok[:, : text.shape[1], -1], msk[:, : text.shape[1], -1] = text, 1
    tok[:, text.shape[1]:, :-1], msk[:, text.shape[1]:, :-1] = audio, 1
    trace = {}
    codes = orc.generate_frame(tok, msk, temp=0.0, trace=trace)
    assert rel(trace["last_h"], g["last_h"]) <= TOL
    assert rel(trace["c0_logits"], g["c0_logits"]) <= TOL
    assert rel(np.stack(trace["ci_logits"], 1), g["ci_logits"]) <= TOL
    np.testing.assert_array_equal(codes, g["codes"])
    # the same prompt fed in two blocks (text, then audio on the cache) is the same frame
    orc.reset_caches()
    orc.backbone(orc.embed_tokens(tok[:, : text.shape[1]]).mul(torch.as_tensor(msk[:, : text.shape[1]], dtype=torch.float32)[..., None]).sum(2))
    codes2 = orc.generate_frame(tok[:, text.shape[1]:], msk[:, text.shape[1]:], temp=0.0)
    np.testing.assert_array_equal(codes2, g["codes"])


# ---------------------------------------------------------------------------------------------------------------------
# Mimi
# ---------------------------------------------------------------------------------------------------------------------
def _mimi():
    g = load("mimi")
    cfg = P.mimi_tiny_config()
    w = P.mimi_synth_checkpoint(cfg, int(g["seed"]), encode=True)
    assert wdigest(w) == str(g["wsha"])
    return g, cfg, w


def test_mimi_decode_blocks_match_transformers_mimi_model():
    """C4 (codec/models/mimi/**): split-RVQ decode with the embedding_sum / max(usage, 1e-5) code books (quantization.py:25-28,97-101,135-182),
    depth-wise causal transposed-conv upsample (conv.py:379-401), the 2-layer transformer (LayerNorm, fused in_proj, traditional RoPE,
    LayerScale, tanh-GELU MLP; transformer.py:62-173) WITHOUT a mask as the reference's non-streaming call runs it (transformer.py:171),
    SEANet decoder (seanet.py:215-307).  Every block is fed what the previous ORACLE block produced."""
    g, cfg, w = _mimi()
    orc = M.MimiOracle(w, cfg)
    pcm, inter = orc.decode(g["codes"], return_inter=True)
    assert rel(inter["quantized"], g["quantized"]) <= TOL
    assert rel(inter["upsampled"], g["upsampled"]) <= TOL
    assert rel(inter["transformer"], g["dec_tr_nomask"]) <= TOL
    assert rel(pcm, g["pcm_from_nomask"]) <= 2e-5
    # transformers' own MimiModel.decode masks causally (sliding window 250): a documented difference of the reference's decode()
    assert rel(pcm, g["hf_decode_causal"]) > 1e-3


def test_mimi_streaming_transformer_matches_transformers_causal_forward():
    """mimi.py:163-168 / transformer.py:79-104: decode_step's transformer with its KV cache and RoPE offset, fed ONE position per step, is the
    causal (sliding-window) transformer -- which is what transformers' MimiTransformerModel.forward computes over the whole block."""
    g, cfg, w = _mimi()
    st = M.MimiStreamOracle(w, cfg)
    x = torch.as_tensor(g["upsampled"])
    with torch.no_grad():
        ys = [st.transformer_step(x[..., i : i + 1]).numpy() for i in range(x.shape[-1])]
    assert rel(np.concatenate(ys, -1), g["dec_tr_causal"]) <= TOL


def test_mimi_encode_blocks_match_transformers_mimi_model():
    """C5 (mimi.py:138-145): SEANet encoder (strided causal convs with the extra right padding, conv.py:189-263), encoder transformer (no mask),
    'edge'-padded stride-2 down-sampler (conv.py:354-367), RVQ encode: the nearest-code search (transformers uses cdist, the reference
    c2 - x.e, quantization.py:35-39: the same argmin) over the residual chain."""
    g, cfg, w = _mimi()
    orc = M.MimiOracle(w, cfg)
    codes, inter = orc.encode(g["wav"], return_inter=True)
    assert rel(inter["seanet"], g["seanet_enc"]) <= TOL
    assert rel(inter["transformer"], g["enc_tr_nomask"]) <= TOL
    assert rel(inter["downsampled"], g["downsampled"]) <= TOL
    np.testing.assert_array_equal(codes, g["enc_codes"])
