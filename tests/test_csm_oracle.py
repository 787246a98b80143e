"""CPU checks of the CSM frame-generator oracle (no reference fixture exists for this path: parity with MLX unpinned; these pin the
restatement's own structure: llama3 RoPE scaling formula, cache == no-cache equivalence, GQA broadcast, sampler rules)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import csm_oracle as C  # noqa: E402
import mlx_audio_amd.params as P  # noqa: E402


def test_llama3_rope_scaling_regions():
    """attention.py:60-82: wavelengths below 8192/4 keep their frequency, above 8192 are divided by the factor, smooth in between."""
    th = C.llama3_theta(64, 500000.0, 32.0)
    base = 1.0 / (500000.0 ** (np.arange(0, 64, 2) / 64.0))
    wl = 2 * np.pi / base
    hi, lo = wl < 2048, wl > 8192
    np.testing.assert_allclose(th[hi], base[hi], rtol=1e-6)
    np.testing.assert_allclose(th[lo], base[lo] / 32.0, rtol=1e-6)
    mid = ~hi & ~lo
    assert mid.any() and np.all(th[mid] < base[mid]) and np.all(th[mid] > base[mid] / 32.0)


def test_real_configuration_parameter_count():
    inv = P.csm_param_inventory(P.csm_config())
    n = sum(int(np.prod(s)) for s in inv.values())
    assert abs(n - 1.553e9) < 0.005e9, n  # llama-1B backbone (0.97 B) + llama-100M decoder + embeddings (0.40 B) + heads


def test_incremental_decoding_equals_full_prefix_and_sampler_rules():
    cfg = P.csm_tiny_config()
    w = P.csm_synth_checkpoint(cfg, 1)
    st = C.LlamaStack({k: np.asarray(v, np.float32) for k, v in w.items()}, "backbone", cfg["backbone"])
    x = torch.tensor(np.random.default_rng(0).standard_normal((2, 6, cfg["backbone"]["hidden"])).astype(np.float32))
    with torch.no_grad():
        full = st(x)
        st.reset()
        a = st(x[:, :4])
        b = st(x[:, 4:5])
        c = st(x[:, 5:6])
    np.testing.assert_allclose(torch.cat([a, b, c], 1).numpy(), full.numpy(), rtol=2e-5, atol=2e-5)
    lg = torch.tensor([[0.1, 2.0, 2.0, -1.0, 1.5]])
    assert C.sample(lg, 0.0, 50, None)[0] == 1  # argmax, first index on ties
    assert C.sample(lg, 1.0, 2, np.array([0.0]))[0] == 1 and C.sample(lg, 1.0, 2, np.array([0.999]))[0] == 2  # top-2 = {1, 2}
    assert C.sample(lg, 1.0, 5, np.array([0.9999]))[0] == 3  # the whole tail is reachable with top_k = V
