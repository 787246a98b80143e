"""Pin the CPU oracle against everything the reference's own tests hold for the hot path.

* mlx_audio/tts/tests/test_interpolate.py:40-84  -- the only numeric known answers on the path
* mlx_audio/tts/tests/test_models.py:92-122      -- Kokoro-82M hyper-parameters (=> 81.76 M parameters)
* mlx_audio/tts/tests/test_models.py:19-77       -- LSTM key renames (sanitize truth table)
* mlx_audio/tts/tests/test_base.py:42-62         -- check_array_shape layout heuristic
* examples/bible-audiobook/audios/**.wav         -- every output is a multiple of 600 samples

No reference test pins a waveform or an activation: waveform-level parity is UNPINNED.
"""
import numpy as np
import pytest

import kokoro_oracle as O
import mlx_audio_amd.params as P


def test_interpolate1d_nearest_known_answers():
    x = np.array([[[1.0, 2.0, 3.0, 4.0]]], np.float32)
    np.testing.assert_allclose(O.interpolate1d(x, 8, "nearest"), [[[1, 1, 2, 2, 3, 3, 4, 4]]], rtol=1e-5)
    np.testing.assert_allclose(O.interpolate1d(x, 2, "nearest"), [[[1, 3]]], rtol=1e-5)


def test_interpolate1d_linear_known_answers():
    x = np.array([[[1.0, 3.0, 5.0, 7.0]]], np.float32)
    r = O.interpolate1d(x, 7, "linear", align_corners=True)
    np.testing.assert_allclose(r, [[[1, 2, 3, 4, 5, 6, 7]]], rtol=1e-5)
    assert O.interpolate1d(x, 7, "linear", align_corners=False).shape == (1, 1, 7)
    xs = np.array([[[5.0]]], np.float32)
    np.testing.assert_allclose(O.interpolate1d(xs, 4, "linear"), [[[5, 5, 5, 5]]], rtol=1e-5)


def test_interpolate_validation_and_sizes():
    with pytest.raises(ValueError):
        O.interpolate(np.zeros((2, 3)), size=4)
    with pytest.raises(ValueError):
        O.interpolate(np.zeros((2, 3, 4)), size=8, scale_factor=2)
    with pytest.raises(ValueError):
        O.interpolate(np.zeros((2, 3, 4)))
    with pytest.raises(ValueError):
        O.interpolate(np.zeros((2, 3, 4, 5)), size=8)
    assert O.interpolate(np.zeros((2, 3, 4), np.float32), size=8).shape == (2, 3, 8)
    assert O.interpolate(np.zeros((2, 3, 4), np.float32), scale_factor=2).shape == (2, 3, 8)


def test_linear_negative_index_wrap_quirk():
    # interpolate.py:88-106: x_low is not clamped, index -1 wraps to the LAST element
    x = np.arange(1, 5, dtype=np.float32)[None, None, :]
    r = O.interpolate1d(x, 1200, "linear")  # scale 300 like the SineGen phase up-sampling
    # first 150 outputs blend in the last input sample (4.0) instead of repeating the first
    assert r[0, 0, 0] > 1.0 and abs(r[0, 0, 0] - (4.0 * (1 - (0.5 / 300 + 0.5)) + 1.0 * (0.5 / 300 + 0.5))) < 1e-4
    assert abs(r[0, 0, 150] - (1.0 + (0.5 / 300))) < 1e-4


def test_scale_factor_float32_size_rule():
    # istftnet.py:568-578 passes 1/mx.array(300): size = ceil(float32(N) * float32(1/300)) must be N/300
    for F in (1, 7, 56, 650, 1111, 3000, 25000):
        a = np.zeros((1, 1, 600 * F), np.float32)
        assert O.interpolate(a, scale_factor=np.float32(1) / np.float32(300), mode="linear").shape[-1] == 2 * F


def test_hyperparameters_give_82M_parameters():
    cfg = P.kokoro_config()
    n = P.param_count(cfg)
    assert abs(n - 81.76e6) < 0.01e6, n
    assert cfg["istftnet"]["upsample_rates"] == [10, 6] and cfg["plbert"]["num_hidden_layers"] == 12


def test_torch_layout_roundtrip_names():
    # kokoro.py:24-44 key map and :197-201 weight_v transposes
    cfg = P.tiny_config()
    w = P.synth_checkpoint(cfg, 0)
    t = P.to_torch_layout(w)
    assert "text_encoder.lstm.weight_ih_l0_reverse" in t and "text_encoder.lstm.Wx_backward" not in t
    assert "text_encoder.cnn.0.1.gamma" in t
    assert t["text_encoder.cnn.0.0.weight_v"].shape == (64, 64, 5)
    assert t["decoder.generator.noise_convs.0.weight"].shape == (32, 22, 12)


def test_istft_window_sum_closed_form():
    # utils.py:121,142-150: periodic Hann, hop = N/4 -> sum(w) = 2 in the interior
    X = np.zeros((11, 41), np.complex64)
    X[0] = 20.0  # constant 1.0 frames
    y = O.istft(X, 5, 20)
    assert y.shape == (200,)
    np.testing.assert_allclose(y[10:-10], 1.0, atol=1e-6)
    np.testing.assert_allclose(y[:5], 1.0, atol=1e-6)  # 3-frame edges are still normalised by their own sum


def test_stft_istft_shapes_and_frame_quantum():
    rng = np.random.default_rng(0)
    x = rng.standard_normal(600 * 3).astype(np.float32)
    S = O.stft(x, 20, 5, 20)
    assert S.shape == (120 * 3 + 1, 11)
    y = O.istft(S.T, 5, 20)
    assert y.shape[0] == 600 * 3  # every reference WAV is a multiple of 600 samples


def test_oracle_forward_tiny_deterministic():
    cfg = P.tiny_config()
    w = P.synth_checkpoint(cfg, 0)
    o = O.KokoroOracle(w, cfg)
    rng = np.random.default_rng(1)
    ids = rng.integers(1, 178, 12).tolist()
    ref_s = (rng.standard_normal((1, 256)) * 0.3).astype(np.float32)
    a1, d1 = o.forward(ids, ref_s, 1.0)
    a2, d2 = o.forward(ids, ref_s, 1.0)
    assert a1.shape[0] == 600 * int(d1.sum()) and d1.shape == (14,)
    np.testing.assert_array_equal(a1, a2)
    # rand_ini cannot change the output (see KokoroOracle.sine_gen docstring)
    a3, _ = o.forward(ids, ref_s, 1.0, rand_ini=rng.standard_normal((1, 9)).astype(np.float32))
    np.testing.assert_array_equal(a1, a3)


def test_oracle_reproduces_committed_golden_case():
    """tests/golden/tiny_case.npz: the well-conditioned stages (durations, F0, N) reproduce on any CPU; the waveform
    reproduces when the vocoder is conditioned on the fixture's F0 / N curves (DESIGN.md "conditioning")."""
    import os

    case = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tiny_case.npz"))
    cfg = P.tiny_config()
    o = O.KokoroOracle(P.synth_checkpoint(cfg, 0), cfg)
    ids = case["ids"].tolist()
    F = int(case["pred_dur"].sum())
    noise = np.random.default_rng(int(case["noise_seed"])).standard_normal((1, 600 * F, 9)).astype(np.float32)
    a, d, it = o.forward(ids, case["ref_s"], 1.0, sine_noise=noise, return_inter=True)
    np.testing.assert_array_equal(d, case["pred_dur"])
    np.testing.assert_allclose(it["F0_pred"], case["F0_pred"], rtol=0, atol=2e-4 * np.abs(case["F0_pred"]).max())
    np.testing.assert_allclose(it["N_pred"], case["N_pred"], rtol=0, atol=2e-4 * np.abs(case["N_pred"]).max())
    a2, _ = o.forward(ids, case["ref_s"], 1.0, sine_noise=noise, f0n_override=(case["F0_pred"], case["N_pred"]))
    assert np.abs(a2 - case["audio"]).max() <= 1e-3 * max(1.0, np.abs(case["audio"]).max())
