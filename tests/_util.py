"""Shared helpers of the GPU parity tests."""
import json
import os
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REPORT = os.path.join(ROOT, "gpurun_out", "parity_report.jsonl")


def report(name, **kw):
    """Append one line to gpurun_out/parity_report.jsonl (kept by gpurun) and echo it."""
    os.makedirs(os.path.dirname(REPORT), exist_ok=True)
    rec = {"name": name, "t": round(time.time(), 1)}
    rec.update({k: (float(v) if isinstance(v, (np.floating, float)) else v) for k, v in kw.items()})
    with open(REPORT, "a") as f:
        f.write(json.dumps(rec) + "\n")
    print("[parity]", json.dumps(rec))


def err_stats(got, ref):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    d = np.abs(got - ref)
    scale = max(1e-12, float(np.abs(ref).max()))
    return {
        "max_abs": float(d.max()) if d.size else 0.0,
        "ref_max": scale,
        "rel_max": float(d.max() / scale) if d.size else 0.0,
        "rms_rel": float(np.sqrt((d**2).mean()) / max(1e-12, np.sqrt((ref**2).mean()))) if d.size else 0.0,
        "p9999_abs": float(np.quantile(d, 0.9999)) if d.size else 0.0,
    }


def ncl_to_nlc(a):
    return np.ascontiguousarray(np.transpose(np.asarray(a), (0, 2, 1)))
