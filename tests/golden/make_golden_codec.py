"""Generates tests/golden/mimi_tiny_case.npz and csm_tiny_case.npz from the CPU oracles (oracle/mimi_oracle.py, oracle/csm_oracle.py) in the
build container: inputs + expected outputs of the tiny configurations, so the GPU tests can check the HIP path against committed
vectors without executing any oracle code.  Run from the repo root:  python tests/golden/make_golden_codec.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import csm_oracle as C  # noqa: E402
import mimi_oracle as M  # noqa: E402
import mlx_audio_amd.params as P  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# ---- Mimi (tiny): decode of random codes, encode of noise
mcfg = P.mimi_tiny_config()
mw = P.mimi_synth_checkpoint(mcfg, 11, encode=True)
rng = np.random.default_rng(111)
codes = rng.integers(0, mcfg["bins"], (2, mcfg["nq"], 9)).astype(np.int32)
pcm_in = (0.3 * rng.standard_normal((2, 1, 1920 * 5 + 321))).astype(np.float32)
orc = M.MimiOracle(mw, mcfg)
np.savez_compressed(os.path.join(HERE, "mimi_tiny_case.npz"), weights_seed=11, codes=codes, pcm_out=orc.decode(codes).astype(np.float32),
                    pcm_in=pcm_in, codes_out=orc.encode(pcm_in).astype(np.int32),
                    pcm_stream=M.MimiStreamOracle(mw, mcfg).decode_frames(codes).astype(np.float32))  # Mimi.decode_step, frame by frame

# ---- CSM (tiny): prompt block + 3 greedy frames
ccfg = P.csm_tiny_config()
cw = P.csm_synth_checkpoint(ccfg, 12)
n = ccfg["audio_num_codebooks"]
B = 2
tok = np.zeros((B, 7, n + 1), np.int32)
msk = np.zeros((B, 7, n + 1), np.float32)
tok[:, :4, -1] = rng.integers(0, ccfg["text_vocab_size"], (B, 4))
msk[:, :4, -1] = 1
tok[:, 4:, :n] = rng.integers(0, ccfg["audio_vocab_size"], (B, 3, n))
msk[:, 4:, :n] = 1
co = C.CsmOracle(cw, ccfg)
frames, logits = [], []
t_in, m_in = tok, msk
for _ in range(4):
    tr = {}
    c = co.generate_frame(t_in, m_in, trace=tr)
    frames.append(c)
    logits.append(np.stack([tr["c0_logits"]] + tr["ci_logits"], 0))
    t_in = np.zeros((B, 1, n + 1), np.int32)
    t_in[:, 0, :n] = c
    m_in = np.zeros((B, 1, n + 1), np.float32)
    m_in[:, 0, :n] = 1
np.savez_compressed(os.path.join(HERE, "csm_tiny_case.npz"), weights_seed=12, tokens=tok, tokens_mask=msk, frames=np.stack(frames).astype(np.int32),
                    logits=np.stack(logits).astype(np.float32))
print("wrote mimi_tiny_case.npz, csm_tiny_case.npz")
