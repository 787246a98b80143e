"""Build libkokoro_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Usage:  python mlx-audio_amd/build.py [--force]
The shared object lands next to this file so it travels with the repository snapshot to the
GPU box (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libkokoro_hip.so")
OBJ = os.path.join(HERE, "csrc", "_obj")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-std=c++17", "-Wall", "-Wno-unused-function"]


# per-file extras.  kk_source.hip: the iSTFT head is written on explicit register pairs (v_pk_*_f32); hipcc's SLP vectoriser otherwise
# re-pairs the scalar tail across outputs and pays two v_mov per packed op it creates
FILE_FLAGS = {"kk_source.hip": ["-fno-slp-vectorize"], "kk_head.hip": ["-fno-slp-vectorize"], "kk_conv_mfma5.hip": ["-fno-slp-vectorize"], "kk_conv_mfma4.hip": ["-fno-slp-vectorize"], "kk_conv_mfma.hip": ["-fno-slp-vectorize"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest() -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        p = os.path.join(CSRC, f)
        if os.path.isfile(p) and f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(p, "rb").read())
    h.update(open(os.path.join(HERE, "..", "include", "kokoro_hip.h"), "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True, extra_flags=(), out: str = OUT, obj_dir: str = OBJ) -> str:
    """extra_flags / out / obj_dir: instrumented side builds (tools/bench_conv.py --trace); the shipped library uses none."""
    OUT, OBJ = out, obj_dir
    stamp = os.path.join(OBJ, "digest.txt")
    dig = _digest() + " ".join(extra_flags)
    if not force and os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig:
        return OUT
    os.makedirs(OBJ, exist_ok=True)

    def cc(src):
        obj = os.path.join(OBJ, src[:-4] + ".o")
        cmd = [HIPCC, *FLAGS, *FILE_FLAGS.get(src, []), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(cc, _sources()))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {OUT}")
    return OUT


if __name__ == "__main__":
    if "--trace" in sys.argv:  # phase-timing build of the MFMA conv kernel, loaded with KK_HIP_LIB=<path>
        build(force="--force" in sys.argv, extra_flags=("-DKK_MFMA_TRACE",), out=os.path.join(HERE, "libkokoro_hip_trace.so"),
              obj_dir=os.path.join(CSRC, "_obj_trace"))
    elif "--mfma32" in sys.argv:  # A/B build: the round-2 32x32x16 MFMA shape in conv variants 4 / 5 (kk_conv_mfma_shared.h)
        build(force="--force" in sys.argv, extra_flags=("-DKK_MFMA32",), out=os.path.join(HERE, "libkokoro_hip_mfma32.so"),
              obj_dir=os.path.join(CSRC, "_obj_mfma32"))
    elif "--exp16" in sys.argv:  # TIMING-ONLY experiment build (wrong conv results): kk_conv_mfma_shared.h, KK_EXP_MFMA16
        build(force="--force" in sys.argv, extra_flags=("-DKK_EXP_MFMA16",), out=os.path.join(HERE, "libkokoro_hip_exp16.so"),
              obj_dir=os.path.join(CSRC, "_obj_exp16"))
    else:
        build(force="--force" in sys.argv)
