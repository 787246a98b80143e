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


def SesameModel_(cfg, w):
    from mlx_audio_amd.csm import SesameModel

    return SesameModel(cfg, w)


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


def _as_bf16_checkpoint(w):
    """What a bf16 checkpoint holds: every tensor rounded to bf16 (load_model keeps the checkpoint's dtype, tts/utils.py:217-262)."""
    return {k: torch.tensor(np.asarray(v, np.float32)).to(torch.bfloat16).float().numpy() for k, v in w.items()}


@pytest.mark.parametrize("wdt", ["float32", "bfloat16"])
def test_csm_tiny_frames_match_oracle(wdt):
    """wdt = bfloat16: the Linear matrices are stored and streamed as bf16 (kk_csm_set_weight_dtype; SwiGLU fused into the down
    projection's input staging); on a bf16 checkpoint that is lossless, so the bar against the fp32-arithmetic oracle is unchanged."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_tiny_config()
    w = P.csm_synth_checkpoint(cfg, 0)
    if wdt == "bfloat16":
        w = _as_bf16_checkpoint(w)
    rng = np.random.default_rng(7)
    B, n = 3, cfg["audio_num_codebooks"]
    orc = C.CsmOracle(w, cfg)
    model = SesameModel(cfg, w, weight_dtype=wdt)
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
        report(f"csm/tiny/{wdt}/frame{step}/logits", **e)
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


@pytest.mark.parametrize("wdt", ["float32", "bfloat16"])
def test_csm_streams_are_independent_bitexact(wdt):
    """A stream produces the same logits alone (B = 1) and next to others (B = 3), prompt block and single-token frames."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_tiny_config()
    w = P.csm_synth_checkpoint(cfg, 5)
    rng = np.random.default_rng(27)
    n = cfg["audio_num_codebooks"]
    tok, msk = _prompt(cfg, rng, 3, 6, 2)

    def run(sel):
        model = SesameModel(cfg, w, weight_dtype=wdt)
        model.setup_caches(len(sel))
        outs = []
        c = model.generate_frame(torch.tensor(tok[sel]), torch.tensor(msk[sel]))
        outs.append(model.debug_logits().cpu().numpy().copy())
        for _ in range(3):
            t_in = torch.zeros((len(sel), 1, n + 1), dtype=torch.int32)
            t_in[:, 0, :n] = c.cpu()
            m_in = torch.zeros((len(sel), 1, n + 1))
            m_in[:, 0, :n] = 1
            c = model.generate_frame(t_in, m_in)
            outs.append(model.debug_logits().cpu().numpy().copy())
        return np.stack(outs)  # [frames][n_cb][B][V]

    full = run([0, 1, 2])
    for b in range(3):
        np.testing.assert_array_equal(run([b])[:, :, 0], full[:, :, b])


def test_csm_graph_replay_gives_the_same_codes():
    """kk_csm_set_graph_mode: eager, captured and replayed single-token frames produce the codes of the eager run."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_tiny_config()
    w = P.csm_synth_checkpoint(cfg, 4)
    rng = np.random.default_rng(17)
    B, n = 2, cfg["audio_num_codebooks"]
    tok, msk = _prompt(cfg, rng, B, 4, 2)
    us = rng.uniform(size=(7, B, n)).astype(np.float32)

    def run(graph):
        model = SesameModel(cfg, w)
        model.setup_caches(B)
        model.set_graph_mode(graph)
        out = [model.generate_frame(torch.tensor(tok), torch.tensor(msk)).cpu().numpy().copy()]
        for i in range(7):
            t_in = np.zeros((B, 1, n + 1), np.int64)
            t_in[:, 0, :n] = out[-1]
            m_in = np.zeros((B, 1, n + 1), np.float32)
            m_in[:, 0, :n] = 1
            out.append(model.generate_frame(torch.tensor(t_in), torch.tensor(m_in), temperature=0.8, top_k=20, uniforms=torch.tensor(us[i])).cpu().numpy().copy())
        assert model.position == tok.shape[1] + 7
        return np.stack(out)

    np.testing.assert_array_equal(run(True), run(False))


def test_csm_setup_caches_twice_in_graph_mode_drops_the_stale_graphs():
    """setup_caches frees and re-allocates the KV caches; a frame-step graph captured before it has the old pointers baked in and must
    not be replayed (use after free).  Graph mode: setup, a few frames, setup AGAIN (same and larger batch), frames equal to the eager run."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_tiny_config()
    w = P.csm_synth_checkpoint(cfg, 4)
    rng = np.random.default_rng(18)
    B, n = 2, cfg["audio_num_codebooks"]
    tok, msk = _prompt(cfg, rng, B, 4, 2)

    def frames(model, k):
        out = [model.generate_frame(torch.tensor(tok), torch.tensor(msk)).cpu().numpy().copy()]
        for _ in range(k):
            t_in = np.zeros((B, 1, n + 1), np.int64)
            t_in[:, 0, :n] = out[-1]
            m_in = np.zeros((B, 1, n + 1), np.float32)
            m_in[:, 0, :n] = 1
            out.append(model.generate_frame(torch.tensor(t_in), torch.tensor(m_in)).cpu().numpy().copy())
        return np.stack(out)

    eager = SesameModel(cfg, w)
    eager.setup_caches(B)
    ref = frames(eager, 5)
    model = SesameModel(cfg, w)
    model.set_graph_mode(True)
    for mb in (B, B, B + 3):  # the second and third setup free the caches the first graphs were captured on
        model.setup_caches(mb)
        junk = torch.full((1 << 22,), 7.0, device="cuda")  # re-use the freed blocks for something else
        np.testing.assert_array_equal(frames(model, 5), ref)
        del junk


def test_csm_head_dims_of_the_real_model_on_a_short_stack():
    """llama-1B / llama-100M head geometry (32 q / 8 kv heads of 64; 8 q / 2 kv heads of 128) with 2 layers each and small vocabularies."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_config()
    # 1100 audio tokens: large enough for the one-wave sampler the real vocabulary (2051) runs on (64 * 16 < V <= 64 * 36)
    cfg = dict(cfg, text_vocab_size=500, audio_vocab_size=1100, audio_num_codebooks=6, max_seq_len=64,
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
    # a single-token frame with top-50 / temperature sampling on injected uniforms: every sampled code equals the oracle's
    n = cfg["audio_num_codebooks"]
    t_in = np.zeros((B, 1, n + 1), np.int64)
    t_in[:, 0, :n] = ref
    m_in = np.zeros((B, 1, n + 1), np.float32)
    m_in[:, 0, :n] = 1
    u = rng.uniform(size=(B, n)).astype(np.float32)
    ref2 = orc.generate_frame(t_in, m_in, temp=0.9, top_k=50, uniforms=u)
    got2 = model.generate_frame(torch.tensor(t_in), torch.tensor(m_in), temperature=0.9, top_k=50, uniforms=torch.tensor(u))
    np.testing.assert_array_equal(got2.cpu().numpy(), ref2)


def test_csm_full_width_layers_split_k_and_wide_blocks_match_oracle():
    """ONE backbone and ONE decoder layer at the real widths (hidden 2048 / 1024, intermediate 8192): the shapes where the matrix-core GEMV takes its
    other forms -- 64-column blocks (gate|up, N = 16384), 8 split-K slices + combine (down, K = 8192), two straight-line rounds (K = 2048) -- and the
    prompt block its GEMM, against the fp32 oracle on a bf16 checkpoint: logits within 2e-4 of their range, every code equal, prompt and two
    single-token frames (one sampled on injected uniforms), B = 3 (a partly filled 8-row slice)."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_config()
    cfg = dict(cfg, text_vocab_size=400, audio_vocab_size=300, audio_num_codebooks=4, max_seq_len=64,
               backbone=dict(cfg["backbone"], num_layers=1), decoder=dict(cfg["decoder"], num_layers=1))
    w = _as_bf16_checkpoint(P.csm_synth_checkpoint(cfg, 6))
    rng = np.random.default_rng(31)
    B, n = 3, cfg["audio_num_codebooks"]
    orc = C.CsmOracle(w, cfg)
    model = SesameModel(cfg, w, weight_dtype="bfloat16")
    model.setup_caches(B)
    tok, msk = _prompt(cfg, rng, B, 7, 3)
    prev = None
    for step in range(3):
        if step == 0:
            t_in, m_in, temp, u = tok, msk, 0.0, None
        else:
            t_in = np.zeros((B, 1, n + 1), np.int64)
            t_in[:, 0, :n] = prev
            m_in = np.zeros((B, 1, n + 1), np.float32)
            m_in[:, 0, :n] = 1
            temp, u = (0.9, rng.uniform(size=(B, n)).astype(np.float32)) if step == 1 else (0.0, None)
        trace = {}
        ref = orc.generate_frame(t_in, m_in, temp=temp, top_k=20, uniforms=u, trace=trace)
        got = model.generate_frame(torch.tensor(t_in), torch.tensor(m_in), temperature=temp, top_k=20, uniforms=None if u is None else torch.tensor(u))
        torch.cuda.synchronize()
        e = err_stats(model.debug_logits().cpu().numpy(), np.stack([trace["c0_logits"]] + trace["ci_logits"], 0))
        report(f"csm/fullwidth/frame{step}/logits", **e)
        assert e["rel_max"] < 2e-4, (step, e)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
        prev = ref


def test_csm_long_cache_single_token_attention_across_chunks():
    """The backbone's single-token attention over a cache longer than one chunk (attn_decode_kernel: 128 keys per chunk at head_dim 64, online
    softmax across chunks): a 125-position prompt, then six frames whose key counts run 126 .. 131 -- the new key in the last slot of a chunk, alone
    in the next chunk, and behind it -- against the fp32 oracle: logits within 2e-4 of their range, every code equal (one frame sampled)."""
    from mlx_audio_amd.csm import SesameModel

    cfg = dict(P.csm_tiny_config(), max_seq_len=520)  # 5 chunks of 128 keys: three key splits, two of them with keys here
    w = _as_bf16_checkpoint(P.csm_synth_checkpoint(cfg, 9))
    rng = np.random.default_rng(41)
    B, n = 2, cfg["audio_num_codebooks"]
    orc = C.CsmOracle(w, cfg)
    model = SesameModel(cfg, w, weight_dtype="bfloat16")
    model.setup_caches(B)
    tok, msk = _prompt(cfg, rng, B, 100, 25)
    prev = None
    for step in range(7):
        if step == 0:
            t_in, m_in, temp, u = tok, msk, 0.0, None
        else:
            t_in = np.zeros((B, 1, n + 1), np.int64)
            t_in[:, 0, :n] = prev
            m_in = np.zeros((B, 1, n + 1), np.float32)
            m_in[:, 0, :n] = 1
            temp, u = (0.9, rng.uniform(size=(B, n)).astype(np.float32)) if step == 3 else (0.0, None)
        trace = {}
        ref = orc.generate_frame(t_in, m_in, temp=temp, top_k=10, uniforms=u, trace=trace)
        got = model.generate_frame(torch.tensor(t_in), torch.tensor(m_in), temperature=temp, top_k=10, uniforms=None if u is None else torch.tensor(u))
        torch.cuda.synchronize()
        e = err_stats(model.debug_logits().cpu().numpy(), np.stack([trace["c0_logits"]] + trace["ci_logits"], 0))
        report(f"csm/longcache/frame{step}/logits", **e)
        assert e["rel_max"] < 2e-4, (step, e)
        np.testing.assert_array_equal(got.cpu().numpy(), ref)
        prev = ref
    assert model.position == 125 + 6


def test_csm_end_to_end_loop_reference_audio_prompt_to_waveform():
    """Config-4 data flow on tiny models: reference audio -> Mimi.encode -> prompt frames (text ids | audio codes + EOS frame | text
    ids) -> frame loop -> Mimi.decode -> waveform.  The same loop driven by the two CPU oracles gives the same codes."""
    import mimi_oracle as MO
    from mlx_audio_amd.mimi import Mimi, MimiConfig
    from mlx_audio_amd.sesame import Model, Segment

    ccfg = dict(P.csm_tiny_config(), audio_vocab_size=64, audio_num_codebooks=4, max_seq_len=128)
    mcfg = P.mimi_tiny_config()  # nq 4, bins 64: matches the frame generator's code books
    cw = P.csm_synth_checkpoint(ccfg, 3)
    mw = P.mimi_synth_checkpoint(mcfg, 3, encode=True)
    model = SesameModel_(ccfg, cw)
    model.setup_caches(2)
    mimi = Mimi(MimiConfig.from_dict(mcfg), mw)
    loop = Model(model, mimi)
    rng = np.random.default_rng(9)
    ref_audio = [(0.3 * rng.standard_normal(1920 * 3)).astype(np.float32) for _ in range(2)]
    ctx = [[Segment(speaker=0, text=rng.integers(0, 300, 5).tolist(), audio=ref_audio[b])] for b in range(2)]
    prompts = [rng.integers(0, 300, 4).tolist() for _ in range(2)]
    res = loop.generate_batch([loop.prompt_frames(ctx[b], prompts[b], 0, voice_match=False) for b in range(2)], max_audio_length_ms=80 * 6,
                              temperature=0.0, stop_on_eos=False)
    audio = torch.stack(res.audio)
    assert res.frames == [6, 6] and tuple(audio.shape) == (2, 6 * 1920) and bool(torch.isfinite(audio).all())
    # oracle loop
    morc = MO.MimiOracle(mw, mcfg)
    corc = C.CsmOracle(cw, ccfg)
    n = 4
    toks, masks = [], []
    for b in range(2):
        codes = morc.encode(ref_audio[b][None, None])[0]
        codes = np.concatenate([codes, np.zeros((n, 1), codes.dtype)], 1)
        rows = []
        for ids in (ctx[b][0].text,):
            f = np.zeros((len(ids), n + 1), np.int64); f[:, -1] = ids
            m = np.zeros((len(ids), n + 1), np.float32); m[:, -1] = 1
            rows.append((f, m))
        f = np.zeros((codes.shape[1], n + 1), np.int64); f[:, :n] = codes.T
        m = np.zeros((codes.shape[1], n + 1), np.float32); m[:, :n] = 1
        rows.append((f, m))
        f = np.zeros((len(prompts[b]), n + 1), np.int64); f[:, -1] = prompts[b]
        m = np.zeros((len(prompts[b]), n + 1), np.float32); m[:, -1] = 1
        rows.append((f, m))
        toks.append(np.concatenate([r[0] for r in rows], 0))
        masks.append(np.concatenate([r[1] for r in rows], 0))
    t_in, m_in = np.stack(toks), np.stack(masks)
    frames = []
    for _ in range(6):
        c = corc.generate_frame(t_in, m_in)
        frames.append(c)
        t_in = np.zeros((2, 1, n + 1), np.int64); t_in[:, 0, :n] = c
        m_in = np.zeros((2, 1, n + 1), np.float32); m_in[:, 0, :n] = 1
    ref_codes = np.stack(frames, 2)
    ref_pcm = morc.decode(ref_codes)[:, 0]
    e = err_stats(audio.cpu().numpy(), ref_pcm)
    report("csm/e2e_tiny/pcm", **e)
    assert e["max_abs"] <= 1e-3 * max(1.0, e["ref_max"]), e
    # stream=True (sesame.py:689-817 with generate_result(stream=True), :619-629): the same frames in partial results of
    # int(streaming_interval * 12.5) frames, decoded incrementally -- against the STREAMING Mimi oracle on the oracle loop's codes
    parts = list(loop.generate_stream(ctx, prompts, max_audio_length_ms=80 * 6, temperature=0.0, stop_on_eos=False, streaming_interval=0.2))
    assert [p.frames for p in parts] == [[2, 2]] * 3 and all(tuple(torch.stack(p.audio).shape) == (2, 2 * 1920) for p in parts)
    ref_stream = MO.MimiStreamOracle(mw, mcfg).decode_frames(ref_codes)[:, 0]
    es = err_stats(torch.cat([torch.stack(p.audio) for p in parts], dim=1).cpu().numpy(), ref_stream)
    report("csm/e2e_tiny/pcm_stream", **es)
    assert es["max_abs"] <= 1e-3 * max(1.0, es["ref_max"]), es


@pytest.mark.parametrize("wdt", ["float32", "bfloat16"])
def test_csm_ragged_prompts_in_one_batch_equal_single_stream_runs_bitexact(wdt):
    """Streams whose prompts differ in length share a batch (left padding + per-item positions, kk_csm_set_padding): the prompt block, the
    greedy and the sampled frames of every stream carry the bits of the stream's own B = 1 run, and so do the decoded waveforms; per-stream
    EOS: a stream that ends early is trimmed to its own frames while the others go on."""
    from mlx_audio_amd.csm import SesameModel
    from mlx_audio_amd.mimi import Mimi, MimiConfig
    from mlx_audio_amd.sesame import Model, Segment

    ccfg = dict(P.csm_tiny_config(), audio_vocab_size=64, audio_num_codebooks=4, max_seq_len=128)
    mcfg = P.mimi_tiny_config()
    cw = P.csm_synth_checkpoint(ccfg, 3)
    if wdt == "bfloat16":
        cw = _as_bf16_checkpoint(cw)
    mimi = Mimi(MimiConfig.from_dict(mcfg), P.mimi_synth_checkpoint(mcfg, 3, encode=True))
    loop = Model(ccfg, mimi=mimi, weights=cw, weight_dtype=wdt)
    rng = np.random.default_rng(19)
    n = 4
    ctxs, texts = [], []
    for b, (na, nt) in enumerate(((3, 4), (1, 9), (5, 2))):  # prompt lengths 5+na+1+nt: 13, 16, 13 ... with audio of different lengths
        audio = (0.3 * rng.standard_normal(1920 * na)).astype(np.float32)
        ctxs.append([Segment(speaker=b, text=rng.integers(0, 300, 5).tolist(), audio=audio)])
        texts.append(rng.integers(0, 300, nt).tolist())
    prompts = [loop.prompt_frames(ctxs[b], texts[b], b, voice_match=(b == 1)) for b in range(3)]
    assert len({p[0].shape[0] for p in prompts}) > 1
    kw = dict(max_audio_length_ms=80 * 7, temperature=0.8, top_k=20, stop_on_eos=False)
    # per-stream uniforms must match between the batched and the single runs: draw them per stream from the same seed by running B = 1
    singles = []
    for b in range(3):
        singles.append(loop.generate_batch([prompts[b]], seed=100 + b, **kw))
    rngs = [np.random.default_rng(100 + b) for b in range(3)]
    both = loop.generate_batch(prompts, uniforms=lambda i: np.stack([r.uniform(size=(1, n))[0] for r in rngs]), **kw)
    assert both.frames == [7, 7, 7]
    for b in range(3):
        np.testing.assert_array_equal(both.codes[b].cpu().numpy(), singles[b].codes[0].cpu().numpy())
        assert torch.equal(both.audio[b], singles[b].audio[0])


def test_csm_model_generate_surface_voice_match_and_stream():
    """Model.generate (sesame.py:689-817) with pre-tokenised text: ref_audio / ref_text -> one context segment; voice_match (default) joins the
    context text with the prompt and continues the context audio without an EOS frame; one GenerationResult per prompt; stream=True yields
    partial results whose concatenation has the same number of samples."""
    from mlx_audio_amd.mimi import Mimi, MimiConfig
    from mlx_audio_amd.sesame import Model, make_sampler

    ccfg = dict(P.csm_tiny_config(), audio_vocab_size=64, audio_num_codebooks=4, max_seq_len=128)
    mcfg = P.mimi_tiny_config()
    mimi = Mimi(MimiConfig.from_dict(mcfg), P.mimi_synth_checkpoint(mcfg, 3, encode=True))
    loop = Model(ccfg, mimi=mimi, weights=P.csm_synth_checkpoint(ccfg, 3))
    rng = np.random.default_rng(29)
    ref = (0.3 * rng.standard_normal(1920 * 3)).astype(np.float32)
    ref_ids, prompts = rng.integers(0, 300, 5).tolist(), [rng.integers(0, 300, 6).tolist(), rng.integers(0, 300, 3).tolist()]
    with pytest.raises(ValueError):
        list(loop.generate("a string needs a tokenizer", ref_audio=ref, ref_text=ref_ids))
    with pytest.raises(FileNotFoundError):
        list(loop.generate(prompts[0]))  # no context, no ref audio: the default speaker prompt would be downloaded
    res = list(loop.generate(prompts, ref_audio=ref, ref_text=ref_ids, max_audio_length_ms=80 * 5, sampler=make_sampler(temp=0.0), stop_on_eos=False))
    assert len(res) == 2 and all(r.token_count == 5 and r.samples == 5 * 1920 and r.sample_rate == 24000 for r in res)
    f, m = loop.prompt_frames([type("S", (), dict(speaker=0, text=ref_ids, audio=ref))()], prompts[0], 0, voice_match=True)
    assert f.shape[0] == len(ref_ids) + len(prompts[0]) + 3 and m[-1, :4].all()  # text ids of both, then the 3 audio frames, no EOS frame
    parts = list(loop.generate(prompts[0], ref_audio=ref, ref_text=ref_ids, max_audio_length_ms=80 * 5, sampler=make_sampler(temp=0.0), stop_on_eos=False,
                               stream=True, streaming_interval=0.16))
    assert [p.token_count for p in parts] == [2, 2, 1] and sum(p.samples for p in parts) == res[0].samples
    # prompt_frames_batch: every stream's reference clips through as few Mimi.encode calls as their lengths allow -> the same prompts, bit for bit
    from mlx_audio_amd.sesame import Segment

    ref2 = (0.3 * rng.standard_normal(1920 * 3)).astype(np.float32)
    ref3 = (0.3 * rng.standard_normal(1920 * 2)).astype(np.float32)
    ctxs = [[Segment(0, ref_ids, ref)], [Segment(0, ref_ids, ref2), Segment(1, prompts[1], ref3)], [Segment(0, ref_ids, None)]]
    for vm in (False, True):
        use = ctxs[:2] if vm else ctxs
        got = loop.prompt_frames_batch(use, [prompts[0]] * len(use), 0, voice_match=vm)
        for c, (gf, gm) in zip(use, got):
            wf, wm = loop.prompt_frames(c, prompts[0], 0, voice_match=vm)
            np.testing.assert_array_equal(gf, wf)
            np.testing.assert_array_equal(gm, wm)


def test_load_model_routes_sesame_checkpoints(tmp_path):
    """load_model on a directory whose config.json says model_type "sesame" (or whose name carries csm: tts/utils.py:17-22,77-121) builds
    sesame.Model: torchtune-style checkpoint names are sanitised (sesame.py:543-569), the codec comes from config["mimi_path"], and the
    frames equal those of a model built directly from the same weights."""
    import json

    from safetensors.numpy import save_file

    from mlx_audio_amd.mimi import Mimi, MimiConfig
    from mlx_audio_amd.sesame import Model, make_sampler
    from mlx_audio_amd.utils import load_model

    ccfg = dict(P.csm_tiny_config(), audio_vocab_size=64, audio_num_codebooks=4, max_seq_len=128)
    mcfg = P.mimi_tiny_config()
    cw = P.csm_synth_checkpoint(ccfg, 3)
    mw = P.mimi_synth_checkpoint(mcfg, 3, encode=True)

    def torchtune(k):  # the names `sanitize` converts FROM
        k = k.replace("self_attn.o_proj", "attn.output_proj").replace("self_attn", "attn")
        k = k.replace("gate_proj", "w1").replace("down_proj", "w2").replace("up_proj", "w3")
        k = k.replace("input_layernorm.weight", "sa_norm.scale").replace("post_attention_layernorm.weight", "mlp_norm.scale")
        return k.replace("backbone.norm.weight", "backbone.norm.scale").replace("decoder.norm.weight", "decoder.norm.scale")

    d = tmp_path / "csm-tiny"
    d.mkdir()
    md = tmp_path / "mimi"
    md.mkdir()
    save_file({k: np.ascontiguousarray(v) for k, v in mw.items()}, str(md / "model.safetensors"))
    json.dump(mcfg, open(md / "config.json", "w"))  # (a tiny codec: the default would be mimi_202407)
    save_file({torchtune(k): np.ascontiguousarray(v) for k, v in cw.items()}, str(d / "model.safetensors"))
    json.dump(dict(ccfg, model_type="sesame", mimi_path=str(md)), open(d / "config.json", "w"))
    model = load_model(str(d))
    assert isinstance(model, Model) and model.sample_rate == 24000 and model._audio_tokenizer.cfg.nq == 4
    direct = Model(ccfg, mimi=Mimi(MimiConfig.from_dict(dict(mcfg)), mw), weights=cw)
    rng = np.random.default_rng(2)
    ref, ids, prompt = (0.3 * rng.standard_normal(1920 * 2)).astype(np.float32), rng.integers(0, 300, 4).tolist(), rng.integers(0, 300, 5).tolist()
    kw = dict(ref_audio=ref, ref_text=ids, max_audio_length_ms=80 * 4, sampler=make_sampler(temp=0.0), stop_on_eos=False)
    a, b = list(model.generate(prompt, **kw)), list(direct.generate(prompt, **kw))
    assert len(a) == 1 and torch.equal(a[0].audio, b[0].audio)


def test_csm_golden_fixture_without_oracle():
    """tests/golden/csm_tiny_case.npz (made by tests/golden/make_golden_codec.py): prompt block + 3 greedy frames, logits and codes."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "csm_tiny_case.npz"))
    cfg = P.csm_tiny_config()
    model = SesameModel_(cfg, P.csm_synth_checkpoint(cfg, int(g["weights_seed"])))
    B, n = g["tokens"].shape[0], cfg["audio_num_codebooks"]
    model.setup_caches(B)
    t_in, m_in = g["tokens"], g["tokens_mask"]
    for f in range(g["frames"].shape[0]):
        c = model.generate_frame(torch.tensor(t_in), torch.tensor(m_in)).cpu().numpy()
        e = err_stats(model.debug_logits().cpu().numpy(), g["logits"][f])
        report(f"csm/golden/frame{f}/logits", **e)
        assert e["rel_max"] < 2e-4, (f, e)
        np.testing.assert_array_equal(c, g["frames"][f])
        t_in = np.zeros((B, 1, n + 1), np.int32)
        t_in[:, 0, :n] = c
        m_in = np.zeros((B, 1, n + 1), np.float32)
        m_in[:, 0, :n] = 1


@pytest.mark.parametrize("V", [67, 1100, 2051, 4000])
def test_csm_sampler_matches_oracle_incl_tie_walls(V):
    """kk_op_csm_sample against oracle sample(): random logits, quantised logits (many exact ties inside and at the edge of the top-k
    set), an all-equal row (more than 64 candidates: the round-based fallback), -inf entries; greedy and top-k / temperature."""
    import ctypes as CT

    from mlx_audio_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(V)
    rows = [rng.standard_normal(V) * 3, np.round(rng.standard_normal(V) * 2) / 2, np.zeros(V), np.round(rng.standard_normal(V)),
            np.where(rng.uniform(size=V) < 0.5, -np.inf, rng.standard_normal(V)), -np.abs(rng.standard_normal(V)) * 50]
    lg = np.stack(rows).astype(np.float32)
    B = lg.shape[0]
    d = torch.tensor(lg, device="cuda")
    for temp, top_k in ((0.0, 50), (0.9, 50), (0.7, 5), (1.3, 64)):
        for trial in range(3):
            u = rng.uniform(size=B).astype(np.float32)
            ref = C.sample(torch.tensor(lg), temp, min(top_k, V), u if temp else None)
            ud = torch.tensor(u, device="cuda")
            out = torch.full((B,), -1, dtype=torch.int32, device="cuda")
            rc = lib.kk_op_csm_sample(CT.c_void_p(torch.cuda.current_stream().cuda_stream), B, V, CT.c_void_p(d.data_ptr()), temp, top_k,
                                      CT.c_void_p(ud.data_ptr()) if temp else None, CT.c_void_p(out.data_ptr()))
            assert rc == 0, lib.kk_last_error()
            torch.cuda.synchronize()
            np.testing.assert_array_equal(out.cpu().numpy(), ref, err_msg=f"temp {temp} top_k {top_k}")


def test_csm_shared_weights_two_generators_in_flight():
    """kk_csm_share: a second generator on the SAME device weights (own KV caches, positions, logits, graphs).  Two jobs on two HIP streams and
    two host threads, graph replay on, give the codes each gives alone; the original's caches are untouched by the other's frames."""
    import threading

    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_tiny_config()
    w = P.csm_synth_checkpoint(cfg, 4)
    rng = np.random.default_rng(15)
    B, n = 2, cfg["audio_num_codebooks"]
    prompts = [_prompt(cfg, rng, B, 7, 3), _prompt(cfg, rng, B, 5, 2)]

    def run(model, tok, msk, frames=6):
        model.reset_caches()
        out = [model.generate_frame(torch.tensor(tok), torch.tensor(msk)).clone()]
        for _ in range(frames - 1):
            t_in = torch.zeros((B, 1, n + 1), dtype=torch.int64, device="cuda")
            t_in[:, 0, :n] = out[-1]
            m_in = torch.zeros((B, 1, n + 1), dtype=torch.float32, device="cuda")
            m_in[:, 0, :n] = 1
            out.append(model.generate_frame(t_in, m_in).clone())
        return torch.stack(out).cpu().numpy()

    a = SesameModel(cfg, w)
    a.setup_caches(B)
    want = [run(a, *prompts[0]), run(a, *prompts[1])]
    b = a.share()
    assert b._h.value != a._h.value
    b.setup_caches(B)
    for m in (a, b):
        m.set_graph_mode(True)
    got, errs = {}, []
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def work(k, model):
        try:
            with torch.cuda.stream(streams[k]):
                for rep in range(3):  # eager, capture, replay
                    got[k] = run(model, *prompts[k])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(k, m)) for k, m in enumerate((a, b))]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert not errs, errs
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1], want[1])
    del b  # the shared generator goes first; the weights stay with `a`
    np.testing.assert_array_equal(run(a, *prompts[1]), want[1])


def test_csm_full_depth_config4_batch8():
    """BASELINE config 4 at FULL size: P.csm_config() unmodified -- 16 + 4 layers, 2 051-entry audio vocabulary, 32 code
    books, 2 048-slot caches -- B = 8 streams with RAGGED prompts of up to 190 positions (left-padded, kk_csm_set_padding), bf16 weight
    storage as `bench.py --config csm` runs it, then 4 single-token frames:
      * graph replay == eager: every code and every logit bit-identical;
      * stream b alone (B = 1, its own unpadded prompt) vs stream b in the batch: every sampled code equal, logits within 2e-5 of their range --
        NOT bit-identical at this size (measured 5e-6): the prompt block's skinny GEMMs split K by the number of rows in flight (B x S), so the
        summation order of the prompt's K / V entries depends on the batch; the single-token steps themselves are row-independent (the tiny-size
        test_csm_streams_are_independent_bitexact holds bit for bit because its prompt takes one K slice either way);
      * frame-0 logits of ONE stream against the CPU oracle (fp32 arithmetic on the same bf16-rounded weights) within 2e-4, its codes equal."""
    from mlx_audio_amd.csm import SesameModel

    cfg = P.csm_config()
    w = _as_bf16_checkpoint(P.csm_synth_checkpoint(cfg, 0))
    rng = np.random.default_rng(44)
    B, n, V = 8, cfg["audio_num_codebooks"], cfg["audio_vocab_size"]
    lens = [190, 131, 190, 64, 177, 190, 99, 150]
    S = max(lens)
    tok = np.zeros((B, S, n + 1), np.int64)
    msk = np.zeros((B, S, n + 1), np.float32)
    for b, L in enumerate(lens):  # [text | audio frames], right-aligned: the first S - L positions are padding (all-zero mask)
        nt = L // 3
        tok[b, S - L : S - L + nt, -1] = rng.integers(0, cfg["text_vocab_size"], nt)
        msk[b, S - L : S - L + nt, -1] = 1
        tok[b, S - L + nt :, :n] = rng.integers(0, V, (L - nt, n))
        msk[b, S - L + nt :, :n] = 1
    us = rng.uniform(size=(4, B, n)).astype(np.float32)

    def run(model, sel, graph):
        model.reset_caches()
        model.set_graph_mode(graph)
        pads = [S - lens[b] for b in sel]
        lo = min(pads)  # a lone stream needs no common padding: drop what every selected stream shares
        model.set_padding([p - lo for p in pads])
        codes = [model.generate_frame(torch.tensor(tok[sel][:, lo:]), torch.tensor(msk[sel][:, lo:])).clone()]
        logits = [model.debug_logits().clone()]
        for i in range(4):
            t_in = torch.zeros((len(sel), 1, n + 1), dtype=torch.int32, device="cuda")
            t_in[:, 0, :n] = codes[-1]
            m_in = torch.zeros((len(sel), 1, n + 1), dtype=torch.float32, device="cuda")
            m_in[:, 0, :n] = 1
            temp, u = (0.9, torch.tensor(us[i][sel])) if i % 2 else (0.0, None)
            codes.append(model.generate_frame(t_in, m_in, temperature=temp, top_k=50, uniforms=u).clone())
            logits.append(model.debug_logits().clone())
        torch.cuda.synchronize()
        return torch.stack(codes).cpu().numpy(), torch.stack(logits).cpu().numpy()  # [5][B][n], [5][n][B][V]

    model = SesameModel(cfg, w, weight_dtype="bfloat16")
    model.setup_caches(B)
    allb = list(range(B))
    c_e, l_e = run(model, allb, False)
    for _ in range(3):  # eager, capture, replay of the single-token frame graph
        c_g, l_g = run(model, allb, True)
    np.testing.assert_array_equal(c_g, c_e)
    np.testing.assert_array_equal(l_g, l_e)
    for b in (3, 0):  # the shortest and a full-length stream, alone
        c_1, l_1 = run(model, [b], False)
        np.testing.assert_array_equal(c_1[:, 0], c_e[:, b])
        e1 = err_stats(l_1[:, :, 0], l_e[:, :, b])
        report(f"csm/full_depth/alone_vs_batch/stream{b}", **e1)
        assert e1["rel_max"] < 2e-5, e1
    # the oracle on stream 3 (64 positions: the CPU cost is per position), frame 0
    b = 3
    orc = C.CsmOracle(w, cfg)
    trace = {}
    ref = orc.generate_frame(tok[b : b + 1, S - lens[b]:], msk[b : b + 1, S - lens[b]:], trace=trace)
    ref_lg = np.stack([trace["c0_logits"]] + trace["ci_logits"], 0)[:, 0]  # [n][V]
    e = err_stats(l_e[0][:, b], ref_lg)
    report("csm/full_depth/frame0/logits_stream3", **e)
    assert e["rel_max"] < 2e-4, e
    np.testing.assert_array_equal(c_e[0][b], ref[0])
