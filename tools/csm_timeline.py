"""GPU-side timeline of one CSM-1B single-token frame in graph replay (bf16 weights, B = 8): in-kernel wall-clock marks (100 MHz) of the
matrix-core GEMVs, the short-cache attention and the split-K combine -- for each launch the gap since the previous instrumented kernel's
last workgroup ended, its span (first workgroup start to last workgroup end) and when workgroup 0 had its input staged.
python tools/csm_timeline.py [--first 200 --count 60]"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402

import mlx_audio_amd.params as P  # noqa: E402
from mlx_audio_amd.csm import SesameModel  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--first", type=int, default=150)
ap.add_argument("--count", type=int, default=48)
a = ap.parse_args()
cfg = P.csm_config()
B = 8
model = SesameModel(cfg, P.csm_synth_checkpoint(cfg, 0), weight_dtype="bfloat16")
model.setup_caches(B)
rng = np.random.default_rng(0)
n = cfg["audio_num_codebooks"]
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
CAP = 2048
buf = torch.zeros((CAP, 8), dtype=torch.int64, device="cuda")


def frame():
    return model.generate_frame(step_tok, step_msk, temperature=0.9, top_k=50, uniforms=us)


for _ in range(2):  # eager, then capture: both number their launches from slot 0
    model.lib.kk_csm_debug_timestamps(C.c_void_p(buf.data_ptr()), CAP)
    frame()
frame()
torch.cuda.synchronize()
buf[:, 1] = -1
buf[:, 2] = 0
torch.cuda.synchronize()
frame()
torch.cuda.synchronize()
t = buf.cpu().numpy().astype(np.uint64)
model.lib.kk_csm_debug_timestamps(None, 0)
used = int((t[:, 2] != 0).sum())
names = {1: "attention (short cache)", 2: "split-K combine"}
rows = []
prev_end = None
for i in range(used):
    cid, st, en, mid, w0s, c0s, w0e, c0e = (int(v) for v in t[i])
    if cid in names:
        nm = names[cid]
    else:
        nm = f"gemv N={cid >> 4} pro={(cid >> 2) & 3} epi={cid & 3}"
    rows.append(dict(i=i, kernel=nm, gap_us=None if prev_end is None else (st - prev_end) / 100.0, span_us=(en - st) / 100.0,
                     staged_us=(mid - st) / 100.0 if mid else None,
                     wg0=[round((v - w0s) / 100.0, 2) if v else None for v in (mid, c0s, c0e, w0e)], wg0_start=(w0s - st) / 100.0))
    prev_end = en
print(json.dumps({"instrumented_launches": used, "frame_span_us": (int(t[used - 1, 2]) - int(t[0, 1])) / 100.0,
                  "sum_span_us": sum(r["span_us"] for r in rows), "sum_gap_us": sum(r["gap_us"] or 0 for r in rows)}))
for r in rows[a.first : a.first + a.count]:
    print(f"{r['i']:5d} {r['kernel']:34s} gap {(r['gap_us'] if r['gap_us'] is not None else 0.0):7.2f}  span {r['span_us']:7.2f}  staged {r['staged_us'] if r['staged_us'] is None else round(r['staged_us'], 2)}  wg0 +{r['wg0_start']:.2f}: staged / weights consumed / reduced / end {r['wg0']}")
