"""CSM-1B frame generation rate (config 4): B streams, one prompt block then N single-token frames; audio-seconds (80 ms per frame)
per wall-second.  python tools/bench_csm.py [--batch 8] [--prompt 64] [--frames 10] [--layers 16]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_amd.params as P  # noqa: E402
from mlx_audio_amd.csm import SesameModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--prompt", type=int, default=64)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--e2e", action="store_true", help="config 4 end to end: reference-audio prompt (Mimi.encode) -> frame loop -> Mimi.decode")
ap.add_argument("--weights", default="float32", choices=["float32", "bfloat16"], help="weight storage of the Linear layers (kk_csm_set_weight_dtype)")
a = ap.parse_args()
cfg = P.csm_config()
t0 = time.time()
w = P.csm_synth_checkpoint(cfg, 0)
t1 = time.time()
model = SesameModel(cfg, w, weight_dtype=a.weights)
del w
model.setup_caches(a.batch)
t2 = time.time()
rng = np.random.default_rng(0)
n, B = cfg["audio_num_codebooks"], a.batch
if a.e2e:
    from mlx_audio_amd.mimi import Mimi, mimi_202407
    from mlx_audio_amd.sesame import Model, Segment

    mcfg = P.mimi_config(32)
    mimi = Mimi(mimi_202407(32), P.mimi_synth_checkpoint(mcfg, 0, encode=True), compute_dtype="bfloat16")
    loop = Model(model, mimi)
    ref = [(0.1 * rng.standard_normal(24000 * 2)).astype(np.float32) for _ in range(B)]  # 2 s of reference audio per stream
    ctx = [[Segment(speaker=0, text=rng.integers(0, cfg["text_vocab_size"], 24).tolist(), audio=ref[b])] for b in range(B)]
    prompts = [loop.prompt_frames(ctx[b], rng.integers(0, cfg["text_vocab_size"], 24).tolist(), 0, voice_match=False) for b in range(B)]
    loop.generate_batch(prompts, max_audio_length_ms=80 * 3, stop_on_eos=False)  # warm-up
    res = loop.generate_batch(prompts, max_audio_length_ms=80 * a.frames, stop_on_eos=False)
    secs = res.audio[0].shape[0] / 24000.0
    print(json.dumps({"metric": "audio-sec/sec (xRT), CSM-1B end to end: reference-audio prompt (Mimi.encode) + text ids -> frames -> Mimi.decode",
                      "value": B * secs / res.processing_time_seconds, "wall_s": res.processing_time_seconds, "audio_s_per_stream": secs, "batch": B,
                      "frames": res.frames[0], "prompt_frames": 24 + 26 + 24, "dtype": ("bf16-weight" if a.weights == "bfloat16" else "f32") + " frame generator, fp32 Mimi.encode, bf16 Mimi.decode",
                      "data": "synthetic (random-init weights, random token ids, noise reference audio, EOS ignored)",
                      "setup_s": {"synth_checkpoint": round(t1 - t0, 1), "load_finalize": round(t2 - t1, 1)}}))
    sys.exit(0)
tok = np.zeros((B, a.prompt, n + 1), np.int64)
msk = np.zeros((B, a.prompt, n + 1), np.float32)
tok[:, :, -1] = rng.integers(0, cfg["text_vocab_size"], (B, a.prompt))
msk[:, :, -1] = 1
torch.cuda.synchronize()
tp = time.perf_counter()
codes = model.generate_frame(torch.tensor(tok), torch.tensor(msk), temperature=0.9, top_k=50, uniforms=torch.tensor(rng.uniform(size=(B, n)).astype(np.float32)))
torch.cuda.synchronize()
prefill_ms = (time.perf_counter() - tp) * 1e3
# the same prompt again on reset caches: without the first call's one-time costs (kernel loading, workspace allocation)
model.reset_caches()
ptok, pmsk, pu = torch.tensor(tok).cuda(), torch.tensor(msk).cuda(), torch.tensor(rng.uniform(size=(B, n)).astype(np.float32)).cuda()
torch.cuda.synchronize()
tp = time.perf_counter()
codes = model.generate_frame(ptok, pmsk, temperature=0.9, top_k=50, uniforms=pu)
torch.cuda.synchronize()
prefill2_ms = (time.perf_counter() - tp) * 1e3
step_tok = torch.zeros((B, 1, n + 1), dtype=torch.int32, device="cuda")
step_msk = torch.zeros((B, 1, n + 1), dtype=torch.float32, device="cuda")
step_msk[:, 0, :n] = 1
us = torch.tensor(rng.uniform(size=(a.frames + 3, B, n)).astype(np.float32), device="cuda")
model.set_graph_mode(True)
for i in range(3):  # warm-up frames (eager, capture, first replay)
    step_tok[:, 0, :n] = codes
    codes = model.generate_frame(step_tok, step_msk, temperature=0.9, top_k=50, uniforms=us[i])
torch.cuda.synchronize()
ts = time.perf_counter()
for i in range(a.frames):
    step_tok[:, 0, :n] = codes
    codes = model.generate_frame(step_tok, step_msk, temperature=0.9, top_k=50, uniforms=us[3 + i])
torch.cuda.synchronize()
dt = (time.perf_counter() - ts) / a.frames
print(json.dumps({"metric": "audio-sec/sec (xRT), CSM-1B frame generation (80 ms of audio per frame and stream), " + ("bf16 weights / fp32 arithmetic" if a.weights == "bfloat16" else "fp32"), "value": B * 0.08 / dt,
                  "ms_per_frame": dt * 1e3, "batch": B, "prompt_tokens": a.prompt, "prefill_ms": prefill_ms, "prefill_ms_second_call": prefill2_ms, "frames_timed": a.frames, "dtype": "bf16 weights, f32 arithmetic" if a.weights == "bfloat16" else "f32",
                  "data": "synthetic (random-init CSM-1B weights, random prompt, injected uniforms)",
                  "setup_s": {"synth_checkpoint": round(t1 - t0, 1), "load_finalize": round(t2 - t1, 1)}}))
