"""Micro-benchmark of the bf16 MFMA conv kernel on the shapes the Kokoro forward launches (B=32, F=650)."""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mlx_audio_amd import _lib  # noqa: E402

lib = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

SHAPES = [
    # name, B, L, Cin, Cout, K, dil, residual
    ("st1_k3", 32, 78001, 128, 128, 3, 1, True),
    ("st1_k7_d3", 32, 78001, 128, 128, 7, 3, True),
    ("st1_k11_d5", 32, 78001, 128, 128, 11, 5, True),
    ("st0_k3", 32, 13000, 256, 256, 3, 1, True),
    ("st0_k7", 32, 13000, 256, 256, 7, 1, True),
    ("st0_k11_d5", 32, 13000, 256, 256, 11, 5, True),
    ("dec_1152_1024_k3", 32, 650, 1152, 1024, 3, 1, False),
    ("dec_1024_1024_k3", 32, 650, 1024, 1024, 3, 1, True),
    ("albert_qkv", 32, 130, 768, 2304, 1, 1, False),
    ("albert_ffn", 32, 130, 768, 2048, 1, 1, False),
]


def run(name, B, L, Cin, Cout, K, dil, res, iters=5, fused=False):
    CinP, CoutP = (Cin + 63) // 64 * 64, (Cout + 127) // 128 * 128
    x = (torch.randn(B, L, CinP, device="cuda") * 1.0).to(torch.bfloat16)
    w = (torch.randn(K, CoutP, CinP, device="cuda") / (K * Cin) ** 0.5).to(torch.bfloat16)
    bias = torch.randn(CoutP, device="cuda")
    out = torch.empty(B, L, Cout, device="cuda", dtype=torch.bfloat16)
    r = torch.randn(B, L, Cout, device="cuda").to(torch.bfloat16) if res else None
    pad = (K * dil - dil) // 2

    A = torch.rand(B, CinP, device="cuda") + 0.5
    Bv = torch.randn(B, CinP, device="cuda") * 0.2
    al = torch.rand(Cin, device="cuda") + 0.5
    part = torch.zeros(B * ((L + 127) // 128) * 2 * Cout, device="cuda")
    nt = C.c_int(0)

    def call_fused():
        rc = lib.kk_op_conv1d_bf16_fused(st(), B, P(x), CinP, L, None, P(w), CinP, CoutP, P(bias), Cin, Cout, K, pad, dil, P(A), P(Bv), CinP,
                                         3, 0.0, P(al), P(r), Cout, 1.0, P(out), Cout, P(part), C.byref(nt))
        assert rc == 0, lib.kk_last_error()

    def call_plain():
        rc = lib.kk_op_conv1d_bf16(st(), B, P(x), CinP, L, None, P(w), CinP, CoutP, P(bias), Cout, K, 0, 1, pad, dil, 0, 1.0, 0, 0.0, P(r), Cout,
                                   1.0, 0, P(out), Cout, L, None, _lib.KK_BF16)
        assert rc == 0, lib.kk_last_error()

    if V4:  # variant 4: weights in MFMA fragment order, straight from global memory into the operand registers
        wf = torch.empty_like(w)
        assert lib.kk_op_pack_w_frag(st(), P(w), P(wf), K, CoutP, CinP) == 0
        lib.kk_debug_set_op_wfrag(P(wf))
        lib.kk_debug_set_op_variant(5 if V5 else 4)
    call = call_fused if fused else call_plain
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    trace = None
    if hasattr(lib, "kk_debug_mfma_trace") and os.environ.get("KK_HIP_LIB"):
        # phase timing (build.py --trace): mean cycles of wave 0 per workgroup
        buf = (C.c_ulonglong * 8)()
        lib.kk_debug_mfma_trace(None, 1)
        call()
        torch.cuda.synchronize()
        lib.kk_debug_mfma_trace(buf, 1)
        n = max(1, buf[5])
        trace = {"blocks": int(buf[5]), "prologue": buf[0] // n, "main_loop": buf[1] // n, "main.w_wait_store": buf[2] // n,
                 "main.x_barrier_transform": buf[3] // n, "epi.acc_in_lds": buf[6] // n, "epi.pass0_stored": buf[7] // n, "total_to_store_end": buf[4] // n}
    if V5 and hasattr(lib, "kk_debug_mfma5_trace") and os.environ.get("KK_HIP_LIB"):
        buf = (C.c_ulonglong * 8)()
        lib.kk_debug_mfma5_trace(None, 1)
        call()
        torch.cuda.synchronize()
        lib.kk_debug_mfma5_trace(buf, 1)
        nwg = 256.0
        trace = {"mfma.loop": buf[0] / nwg, "mfma.wait_A": buf[1] / nwg, "mfma.A_to_B": buf[2] / nwg, "svc.loop": buf[3] / nwg, "svc.wait_A": buf[4] / nwg,
                 "svc.A_to_B": buf[5] / nwg, "svc.epilogue": buf[6] / nwg, "svc.load_transform": buf[7] / nwg}
        trace = {k: int(v) for k, v in trace.items()}
    fl = 2.0 * B * L * Cin * Cout * K
    by = B * L * (Cin + Cout * (2 if res else 1)) * 2
    if V4:
        torch.cuda.synchronize()
        lib.kk_debug_set_op_wfrag(None)
    out = {"name": name + ("+fused" if fused else "") + ("+v5" if V5 else "+v4" if V4 else ""), "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1), "GBs": round(by / ms / 1e6, 1)}
    if trace:
        out["trace_cycles"] = trace
    return out


V5 = "--v5" in sys.argv
V4 = "--v4" in sys.argv or V5

if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    fused = "--fused" in sys.argv
    sel = args or None
    for s in SHAPES:
        if sel and s[0] not in sel:
            continue
        if fused and (s[3] != s[4] or not s[7]):
            continue
        print(json.dumps(run(*s, fused=fused)), flush=True)
