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
        "p99_abs": float(np.quantile(d, 0.99)) if d.size else 0.0,
    }


def ncl_to_nlc(a):
    return np.ascontiguousarray(np.transpose(np.asarray(a), (0, 2, 1)))


def log_spectral_distance(got, ref, n_fft=1024, hop=256, floor_db=-60.0):
    """Phase-robust distance between two waveforms: RMS difference in dB between their short-time MAGNITUDE spectra
    (Hann window `n_fft`, hop `hop`), over the time-frequency cells where the reference is within `floor_db` of its
    loudest cell (quieter cells are noise in any implementation).  A pure phase change of a partial leaves it at ~0;
    a change of level by x dB in every cell gives x.  Also returns the level-weighted relative magnitude error
    sum|G - R| / sum|R| over all cells."""
    from scipy.signal import stft

    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    _, _, G = stft(got, nperseg=n_fft, noverlap=n_fft - hop, window="hann", boundary=None, padded=False)
    _, _, R = stft(ref, nperseg=n_fft, noverlap=n_fft - hop, window="hann", boundary=None, padded=False)
    G, R = np.abs(G), np.abs(R)
    peak = max(R.max(), 1e-30)
    eps = peak * 10.0 ** (floor_db / 20.0)
    keep = R >= eps
    d = 20.0 * np.log10((G[keep] + eps) / (R[keep] + eps))
    return {"lsd_db": float(np.sqrt(np.mean(d * d))) if d.size else 0.0, "mag_rel_l1": float(np.abs(G - R).sum() / max(R.sum(), 1e-30)),
            "cells": int(keep.sum())}


def band_energy_distance(got, ref, nbands=24, n_fft=2048, hop=1024):
    """RMS difference in dB between the energies of two waveforms in `nbands` log-spaced frequency bands per 85 ms frame
    (n_fft 2048 at 24 kHz): the spectral envelope over time, insensitive to the phase and fine position of partials."""
    from scipy.signal import stft

    def bands(x):
        _, _, S = stft(np.asarray(x, np.float64), nperseg=n_fft, noverlap=n_fft - hop, window="hann", boundary=None, padded=False)
        p = np.abs(S) ** 2
        edges = np.unique(np.round(np.geomspace(2, n_fft // 2 + 1, nbands + 1)).astype(int))
        return np.stack([p[edges[i] : edges[i + 1]].sum(0) for i in range(len(edges) - 1)])

    g, r = bands(got), bands(ref)
    eps = max(r.max(), 1e-30) * 1e-6
    d = 10.0 * np.log10((g + eps) / (r + eps))
    return float(np.sqrt(np.mean(d * d)))
