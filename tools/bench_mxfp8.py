"""Times the MX-fp8 linear (activation pre-pass + product, kk_op_linear_mxfp8) per layer shape of config 5 (B = 64, T = 130):
python tools/bench_mxfp8.py [--iters 50]."""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlx_audio_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--only", default="")
a = ap.parse_args()
lib = _lib.load()
B, T = 64, 130
M = B * T
SHAPES = [("map_in", 128, 768, 0), ("qkv", 768, 2304, 0), ("dense", 768, 768, 0), ("ffn_gelu", 768, 2048, 2), ("ffn_out", 2048, 768, 0), ("bert_encoder", 768, 512, 0)]
rng = np.random.default_rng(0)
p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for name, K, N, act in SHAPES:
    if a.only and a.only != name:
        continue
    w = (rng.standard_normal((N, K)) * 0.03).astype(np.float32)
    qb, sb = C.c_size_t(), C.c_size_t()
    _lib.check(lib.kk_mxfp8_bytes(N, K, C.byref(qb), C.byref(sb)), "bytes")
    wq, ws = np.zeros(qb.value, np.uint8), np.zeros(sb.value, np.uint8)
    _lib.check(lib.kk_mxfp8_pack_weight(w.ctypes.data_as(C.c_void_p), N, K, 64, wq.ctypes.data_as(C.c_void_p), ws.ctypes.data_as(C.c_void_p)), "pack")
    _lib.check(lib.kk_mxfp8_bytes(M, K, C.byref(qb), C.byref(sb)), "bytes")
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    aq = torch.zeros(qb.value, dtype=torch.uint8, device="cuda")
    asc = torch.zeros(sb.value, dtype=torch.uint8, device="cuda")
    o = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    wqd, wsd = torch.as_tensor(wq).cuda(), torch.as_tensor(ws).cuda()
    bias = torch.zeros(N, device="cuda")
    lens = torch.full((B,), T, dtype=torch.int32, device="cuda")

    def run():
        _lib.check(lib.kk_op_linear_mxfp8(st(), p(x), K, M, T, p(lens), K, p(wqd), p(wsd), N, p(bias), act, p(aq), p(asc), p(o), N), "op")

    for _ in range(5):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    out[name] = {"us": round(us, 2), "TFLOP/s": round(2.0 * M * N * K / (us * 1e-6) / 1e12, 1)}
print(json.dumps({"M": M, "layers": out}))
