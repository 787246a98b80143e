"""Debug helper: run a tiny bf16 ragged forward and report the first named stage that holds a non-finite value."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mlx_audio_amd.params as P
from mlx_audio_amd import _lib
from mlx_audio_amd.engine import KokoroEngine

cfg = P.kokoro_config() if "full" in sys.argv else P.tiny_config()
w = P.synth_checkpoint(cfg, 0)
rng = np.random.default_rng(31)
utts = [rng.integers(1, 178, n).tolist() for n in (12, 7, 9)]
flags = int(sys.argv[1]) if len(sys.argv) > 1 else 0
eng = KokoroEngine(cfg, w, compute_dtype="bfloat16")
eng.lib.kk_debug_force_generic(eng._h, flags)
dev = eng.device
rows = np.load(os.path.join(ROOT, "tests", "golden", "af_heart_rows.npz"))["rows"]
ref_s = torch.tensor(rows[rng.integers(0, rows.shape[0], 3)].astype(np.float32), device=dev)
ids, lens, Tmax = eng.pack_ids(utts)
sp = torch.ones(3, device=dev)
ws = eng.workspace(3, Tmax, 110)
ws.fill_(255 if "poison" in sys.argv else 0)  # poison: 0xFFFF bf16 / 0xFFFFFFFF fp32 are NaNs
wav, pred, nfr = eng.forward(ids, lens, ref_s, sp, 110, noise_mode=_lib.NOISE_PHILOX, seed=5)
torch.cuda.synchronize()
print("nframes", nfr.tolist(), "pred", pred.tolist())
print("wav finite per utt", [bool(torch.isfinite(wav[b]).all()) for b in range(3)])
for name in ["bert_dur", "d", "duration", "t_en", "en", "asr", "F0_pred", "N_pred", "dec_out", "har_source", "har", "gen_pre_res0", "gen_stage0",
             "gen_pre_res1", "gen_stage1", "conv_post"]:
    try:
        t = eng.debug_fetch(name)
    except Exception as e:
        print(name, "n/a", e)
        continue
    fin = [bool(torch.isfinite(t[b]).all()) for b in range(t.shape[0])]
    print(name, tuple(t.shape), fin, float(t[torch.isfinite(t)].abs().max()) if torch.isfinite(t).any() else None)
