"""Mimi.decode throughput (CSM row C4): B items x Nf frames of random codes -> pcm; audio-seconds per wall-second, plus the CPU
oracle on one item.  python tools/bench_mimi.py [--batch 8] [--frames 125] [--steps 10]
--stream: the frame-by-frame path instead (Mimi.decode_step, kk_mimi_decode_step): ms per one-frame step of the whole batch."""
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
from mlx_audio_amd.mimi import Mimi, mimi_202407  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--frames", type=int, default=125)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--no-cpu-baseline", action="store_true")
ap.add_argument("--dtype", default="bfloat16", choices=["float32", "bfloat16"])
ap.add_argument("--stream", action="store_true")
a = ap.parse_args()
cfg = P.mimi_config(32)
w = P.mimi_synth_checkpoint(cfg, 0)
model = Mimi(mimi_202407(32), w, compute_dtype=a.dtype)
codes = torch.tensor(np.random.default_rng(0).integers(0, 2048, (a.batch, 32, a.frames)), device="cuda", dtype=torch.int32)
if a.stream:
    model.decode_step(codes[:, :, :1])  # opens the stream, sizes the workspace
    torch.cuda.synchronize()
    frames = [codes[:, :, i : i + 1].contiguous() for i in range(a.frames)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    model.reset_stream()
    t0 = time.perf_counter()
    e0.record()
    for f in frames:
        model.decode_step(f)
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.frames
    print(json.dumps({"metric": "ms per streaming decode step (one code frame of the whole batch), Mimi.decode_step mimi_202407", "batch": a.batch,
                      "frames": a.frames, "ms_per_step_gpu": e0.elapsed_time(e1) / a.frames, "ms_per_step_wall": wall * 1e3,
                      "xRT": a.batch * 0.08 / wall, "dtype": "f32"}))
    sys.exit(0)
for _ in range(2):
    model.decode(codes)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.steps):
    pcm = model.decode(codes)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.steps
audio_s = a.batch * a.frames / 12.5
out = {"metric": "audio-sec/sec (xRT), Mimi.decode mimi_202407", "value": audio_s / dt, "ms_per_decode": dt * 1e3, "batch": a.batch,
       "frames": a.frames, "audio_s_per_item": a.frames / 12.5, "dtype": "bf16" if a.dtype == "bfloat16" else "f32", "data": "synthetic (random-init decode-side weights, random codes)"}
if not a.no_cpu_baseline:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import mimi_oracle as M

    orc = M.MimiOracle(w, cfg)
    c1 = codes[:1].cpu().numpy()
    t1 = time.perf_counter()
    orc.decode(c1)
    dc = time.perf_counter() - t1
    out["cpu_baseline"] = {"value": (a.frames / 12.5) / dc, "unit": "audio-sec/sec", "cores": int(torch.get_num_threads()), "kind": "port",
                           "sample": f"1 item of {a.frames} frames in {dc:.2f} s (oracle/mimi_oracle.py, torch-CPU fp32)"}
print(json.dumps(out))
