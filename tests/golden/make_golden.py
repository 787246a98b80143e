"""Regenerates the committed fixtures in tests/golden/ (run in the build container, where
/root/reference is mounted and the CPU oracle can run).  The GPU box only reads the results.

  af_heart_rows.npz           10 real style rows of the reference's vendored voice pack
                              (mlx_audio_swift/tts/Swift-TTS/Kokoro/Resources/af_heart.json, a [510,1,256] array):
                              golden INPUTS (pipeline.py:236 picks row len(ps)-1).
  tiny_case.npz               ids / style row / speed / noise seed of one tiny-config utterance with the ORACLE's
                              waveform, durations and F0 curve (what the HIP path must reproduce).
  config2_oracle_digest.json  mean/std of the well-conditioned oracle stages (duration, F0, N, decoder output) for the
                              two BASELINE config-2 utterances of tests/test_gpu_forward.py::
                              test_config2_slice_matches_oracle: pins the oracle run on the GPU box's CPU to the one
                              made here.  (The waveform itself is NOT reproducible across CPUs: it is chaotic in the
                              float32 round-off of F0, see DESIGN.md "conditioning".)
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]

import kokoro_oracle as O  # noqa: E402
import mlx_audio_amd.params as P  # noqa: E402

REF_PACK = "/root/reference/mlx_audio_swift/tts/Swift-TTS/Kokoro/Resources/af_heart.json"
ROWS = [2, 6, 10, 11, 40, 80, 127, 200, 300, 509]


def main():
    if os.path.exists(REF_PACK):
        a = np.array(json.load(open(REF_PACK)), dtype=np.float32)
        np.savez_compressed(os.path.join(HERE, "af_heart_rows.npz"), rows=a[ROWS, 0, :], index=np.array(ROWS))
    rows = np.load(os.path.join(HERE, "af_heart_rows.npz"))["rows"]

    # --- tiny case
    cfg = P.tiny_config()
    w = P.synth_checkpoint(cfg, 0)
    orc = O.KokoroOracle(w, cfg)
    rng = np.random.default_rng(123)
    ids = rng.integers(1, 178, 10).astype(np.int32)
    ref_s = rows[3:4]
    dur = orc.text_stage(ids.tolist(), ref_s, 1.0)
    F = int(dur.sum())
    noise = np.random.default_rng(77).standard_normal((1, 600 * F, 9)).astype(np.float32)
    audio, pred, inter = orc.forward(ids.tolist(), ref_s, 1.0, sine_noise=noise, return_inter=True)
    np.savez_compressed(os.path.join(HERE, "tiny_case.npz"), ids=ids, ref_s=ref_s, speed=np.float32(1.0), noise_seed=np.int64(77),
                        audio=audio, pred_dur=pred, F0_pred=inter["F0_pred"], N_pred=inter["N_pred"])
    print("tiny_case: F =", F, "samples =", audio.shape[0])

    # --- config-2 digest (same seeds as tests/test_gpu_forward.py::test_config2_slice_matches_oracle)
    cfg = P.kokoro_config()
    w = P.synth_checkpoint(cfg, 0)
    orc = O.KokoroOracle(w, cfg)
    rng = np.random.default_rng(40)
    utts = [rng.integers(1, 178, 128).tolist() for _ in range(2)]
    rng2 = np.random.default_rng(2)
    idx = rng2.integers(0, rows.shape[0], 2)
    ref = rows[idx]
    Fmax = 650
    noise = rng2.standard_normal((2, 600 * Fmax, 9)).astype(np.float32)
    out = {"utt": []}
    for b in range(2):
        a, pd, it = orc.forward(utts[b], ref[b : b + 1], 1.0, forced_dur=np.full(130, 5, np.int32), sine_noise=noise[b : b + 1], return_inter=True)
        st = lambda v: [float(np.mean(v, dtype=np.float64)), float(np.std(v, dtype=np.float64))]
        out["utt"].append({"samples": int(a.shape[0]), "wav_std": float(a.astype(np.float64).std()), "F0_pred": st(it["F0_pred"]),
                           "N_pred": st(it["N_pred"]), "dec_out": st(it["dec_out"]), "duration": st(it["duration"]),
                           "pred_dur_sum": int(pd.sum())})
        print("config2 utt", b, out["utt"][-1])
    json.dump(out, open(os.path.join(HERE, "config2_oracle_digest.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
