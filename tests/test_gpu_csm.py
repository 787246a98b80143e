"""CSM-1B frame generator through the C ABI (kk_csm_*) against the CPU oracle on identical synthetic weights, tokens and injected
uniforms: logits of every code book within 2e-4 of their max, sampled codes identical (greedy and top-k / temperature)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import csm_oracle as C  # noqa: E402
import mlx_audio_amd.params as P  # noqa: E402
from _util import err_stats, report  # noqa: E402

pytestmark = pytest.mark.gpu


def _prompt(cfg, rng, B, n_text, n_audio):
    n = cfg["audio_num_codebooks"]
    S = n_text + n_audio
    tok = np.zeros((B, S, n + 1), np.int64)
    msk = np.zeros((B, S, n + 1), np.float32)
    tok[:, :n_text, -1] = rng.integers(0, cfg["text_vocab_size"], (B, n_text))
    msk[:, :n_text, -1] = 1
    tok[:, n_text:, :n] = rng.integers(0, cfg["audio_vocab_size"], (B, n_audio, n))
    msk[:, n_text:, :n] = 1
    return tok, msk


def test_csm_tiny_frames_match_oracle():
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_tiny_config()
    w = P.csm_synth_checkpoint(cfg, 0)
    rng = np.random.default_rng(7)
    B, n = 3, cfg["audio_num_codebooks"]
    orc = C.CsmOracle(w, cfg)
    model = SesameModel(cfg, w)
    model.setup_caches(B)
    tok, msk = _prompt(cfg, rng, B, 5, 3)
    frames = []
    for step in range(5):
        if step == 0:
            t_in, m_in = tok, msk
            temp, u = 0.0, None  # greedy prompt frame
        else:
            t_in = np.zeros((B, 1, n + 1), np.int64)
            t_in[:, 0, :n] = frames[-1]
            m_in = np.zeros((B, 1, n + 1), np.float32)
            m_in[:, 0, :n] = 1
            temp, u = (0.9, rng.uniform(size=(B, n)).astype(np.float32)) if step % 2 else (0.0, None)
        trace = {}
        ref = orc.generate_frame(t_in, m_in, temp=temp, top_k=10, uniforms=u, trace=trace)
        pos = model.position
        got = model.generate_frame(torch.tensor(t_in), torch.tensor(m_in), input_pos=np.broadcast_to(pos + np.arange(t_in.shape[1]), (B, t_in.shape[1])),
                                   temperature=temp, top_k=10, uniforms=None if u is None else torch.tensor(u))
        torch.cuda.synchronize()
        lg = model.debug_logits().cpu().numpy()
        ref_lg = np.stack([trace["c0_logits"]] + trace["ci_logits"], 0)
        e = err_stats(lg, ref_lg)
        report(f"csm/tiny/frame{step}/logits", **e)
        assert e["rel_max"] < 2e-4, (step, e)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
        assert model.position == orc.backbone.offset
        frames.append(ref)
    # reset: the same prompt gives the same first frame again
    model.reset_caches()
    again = model.generate_frame(torch.tensor(tok), torch.tensor(msk)).cpu().numpy()
    np.testing.assert_array_equal(again, frames[0])
    with pytest.raises(Exception):  # a multi-token block on a non-empty cache is refused (sesame.py:41-48)
        model.generate_frame(torch.tensor(tok), torch.tensor(msk))


def test_csm_head_dims_of_the_real_model_on_a_short_stack():
    """llama-1B / llama-100M head geometry (32 q / 8 kv heads of 64; 8 q / 2 kv heads of 128) with 2 layers each and small vocabularies."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_config()
    cfg = dict(cfg, text_vocab_size=500, audio_vocab_size=131, audio_num_codebooks=6, max_seq_len=64,
               backbone=dict(cfg["backbone"], num_layers=2, intermediate=1024), decoder=dict(cfg["decoder"], num_layers=2, intermediate=768))
    w = P.csm_synth_checkpoint(cfg, 2)
    rng = np.random.default_rng(8)
    B = 2
    orc = C.CsmOracle(w, cfg)
    model = SesameModel(cfg, w)
    model.setup_caches(B)
    tok, msk = _prompt(cfg, rng, B, 9, 4)
    trace = {}
    ref = orc.generate_frame(tok, msk, trace=trace)
    got = model.generate_frame(torch.tensor(tok), torch.tensor(msk))
    torch.cuda.synchronize()
    e = err_stats(model.debug_logits().cpu().numpy(), np.stack([trace["c0_logits"]] + trace["ci_logits"], 0))
    report("csm/realheads/logits", **e)
    assert e["rel_max"] < 2e-4, e
    np.testing.assert_array_equal(got.cpu().numpy(), ref)
