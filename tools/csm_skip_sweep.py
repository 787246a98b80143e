"""In-situ cost of each kernel class of the CSM-1B single-token frame (bf16 weights, B = 8, graph replay): the frame time with the class
NOT launched (kk_csm_debug_skip; the codes are wrong, only the clock is read).  python tools/csm_skip_sweep.py [--frames 10]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mlx_audio_amd.params as P  # noqa: E402
from mlx_audio_amd.csm import SesameModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--batch", type=int, default=8)
a = ap.parse_args()
cfg = P.csm_config()
model = SesameModel(cfg, P.csm_synth_checkpoint(cfg, 0), weight_dtype="bfloat16")
model.setup_caches(a.batch)
rng = np.random.default_rng(0)
n, B = cfg["audio_num_codebooks"], a.batch
tok = np.zeros((B, 64, n + 1), np.int64)
msk = np.zeros((B, 64, n + 1), np.float32)
tok[:, :, -1] = rng.integers(0, cfg["text_vocab_size"], (B, 64))
msk[:, :, -1] = 1
codes = model.generate_frame(torch.tensor(tok), torch.tensor(msk))
step_tok = torch.zeros((B, 1, n + 1), dtype=torch.int32, device="cuda")
step_msk = torch.zeros((B, 1, n + 1), dtype=torch.float32, device="cuda")
step_msk[:, 0, :n] = 1
step_tok[:, 0, :n] = codes
us = torch.tensor(rng.uniform(size=(B, n)).astype(np.float32), device="cuda")
model.set_graph_mode(True)
names = ["q|k|v", "attention", "o", "gate|up", "down", "combine", "heads", "sampler", "projection"]
out = {}


def frame_ms(mask):
    model.lib.kk_csm_debug_skip(mask)
    for _ in range(3):
        model.generate_frame(step_tok, step_msk, temperature=0.9, top_k=50, uniforms=us)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(a.frames):
        model.generate_frame(step_tok, step_msk, temperature=0.9, top_k=50, uniforms=us)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / a.frames * 1e3


full = frame_ms(0)
out["full_frame_ms"] = round(full, 3)
for b, nm in enumerate(names):
    out[f"without {nm}"] = round(frame_ms(1 << b), 3)
    out[f"cost of {nm} (ms)"] = round(full - out[f"without {nm}"], 3)
out["nothing launched but embed / rmsnorm / advance"] = round(frame_ms(511), 3)
for b, nm in enumerate(names):
    out[f"only {nm} launched"] = round(frame_ms(511 & ~(1 << b)), 3)
out["only q|k|v + attention"] = round(frame_ms(511 & ~3), 3)
out["only gate|up + down"] = round(frame_ms(511 & ~24), 3)
out["only gate|up + down + combine"] = round(frame_ms(511 & ~56), 3)
model.lib.kk_csm_debug_skip(0)
print(json.dumps(out, indent=1))
