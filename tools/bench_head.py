"""Micro-benchmark of the fused vocoder head (kk_head.hip) at the bench shape: B = 32 x 78 001 frames x 128 channels.
KK_HEAD_ROWS=128|256 picks the tile; KK_HEAD_DBG bits are timing ablations (wrong results): 1 one k-step, 2 no frame arithmetic, 4 cached input."""
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlx_audio_amd import _lib  # noqa: E402

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, Tf, Cn = 32, 78001, 128
x = torch.randn(B, Tf, Cn, device="cuda").to(torch.bfloat16)
w = (torch.randn(7, 22, Cn, device="cuda") * 0.02).to(torch.bfloat16)
bias = torch.randn(22, device="cuda") * 0.1
wf = torch.empty(7 * 8 * 64 * 8, dtype=torch.bfloat16, device="cuda")
assert lib.kk_op_pack_head_w(st(), P(w), P(wf)) == 0
wav = torch.empty(B, 5 * (Tf - 1), device="cuda")


def call():
    assert lib.kk_op_conv_post_istft(st(), B, P(x), Cn, Tf, None, P(wf), P(bias), C.c_float(float(os.environ.get("KK_HEAD_SLOPE", "0.01"))), P(wav), None, 0) == 0, lib.kk_last_error()


for _ in range(3):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    call()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
by = B * Tf * (Cn * 2 + 20)
print(json.dumps({"rows": os.environ.get("KK_HEAD_ROWS", "256"), "dbg": os.environ.get("KK_HEAD_DBG", "0"), "us": round(us, 1), "GBs": round(by / us / 1e3, 1),
                  "frac_of_8TBs": round(by / us / 1e3 / 8000, 3)}))
