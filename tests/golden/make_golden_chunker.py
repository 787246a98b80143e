"""Known-answer vectors for the 510-phoneme chunk planner and the timestamp joiner (reference behaviour:
mlx_audio/tts/models/kokoro/pipeline.py:163-226 and :292-328).  Random token streams (text, phonemes, whitespace) with punctuation
at every tier of the split waterfall, with None phonemes, empty phonemes and unicode closers; random duration vectors.

Run ONCE in the build container against the implementation whose behaviour had been checked line by line against the reference text,
the output (tests/golden/chunker_cases.json) pins every later rewrite of mlx-audio_amd/pipeline.py:

    python tests/golden/make_golden_chunker.py
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from mlx_audio_amd.pipeline import KokoroPipeline  # noqa: E402

PUNCT = ["!", ".", "?", "…", ":", ";", ",", "—"]
BUMPS = [")", "”"]
LETTERS = list("abdefhijklmnopstuvwzɾˈˌəɪʊ")


def stream(rng, n, p_punct, wlen):
    toks = []
    for _ in range(n):
        r = rng.random()
        if r < p_punct:
            ch = PUNCT[rng.integers(len(PUNCT))]
            toks.append([ch, ch, " " if rng.random() < 0.8 else ""])
            if rng.random() < 0.15:
                b = BUMPS[rng.integers(len(BUMPS))]
                toks.append([b, b, " "])
        elif r < p_punct + 0.03:
            toks.append(["<unk>", None, " "])
        elif r < p_punct + 0.05:
            toks.append(["", "", " " if rng.random() < 0.5 else ""])
        else:
            k = int(rng.integers(1, wlen))
            ph = "".join(rng.choice(LETTERS, k))
            toks.append([ph.upper(), ph, " " if rng.random() < 0.85 else ""])
    return toks


def main():
    rng = np.random.default_rng(20261004)
    p = KokoroPipeline(lang_code="a", model=False, repo_id="m")
    chunk_cases, ts_cases = [], []
    for n, pp, wl in [(40, 0.1, 6), (300, 0.08, 9), (300, 0.0, 9), (500, 0.02, 12), (700, 0.15, 5), (260, 0.3, 14), (900, 0.05, 4), (120, 0.01, 30),
                      (400, 0.12, 8), (1000, 0.04, 7), (350, 0.06, 10), (200, 0.5, 12)]:
        for _ in range(3):
            toks = stream(rng, n, pp, wl)
            objs = [SimpleNamespace(text=t, phonemes=ph, whitespace=ws) for t, ph, ws in toks]
            out = [[gs, ps, len(tk)] for gs, ps, tk in p.en_tokenize(objs)]
            chunk_cases.append({"tokens": toks, "chunks": out})
    for n in (1, 2, 3, 5, 9, 20, 60):
        for _ in range(6):
            toks = stream(rng, n, 0.15, 7)
            objs = [SimpleNamespace(text=t, phonemes=("" if ph is None else ph), whitespace=ws, start_ts=None, end_ts=None) for t, ph, ws in toks]
            need = 2 + sum(len(o.phonemes) + (1 if o.whitespace else 0) for o in objs)
            L = max(0, need + int(rng.integers(-4, 3)))
            pd = rng.integers(1, 12, L).astype(int).tolist()
            KokoroPipeline.join_timestamps(objs, np.asarray(pd, np.int32))
            ts_cases.append({"tokens": [[o.text, o.phonemes, o.whitespace] for o in objs], "pred_dur": pd,
                             "ts": [[o.start_ts, o.end_ts] for o in objs]})
    json.dump({"chunk_cases": chunk_cases, "timestamp_cases": ts_cases},
              open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "chunker_cases.json"), "w"), ensure_ascii=False)
    print(len(chunk_cases), "chunk cases;", len(ts_cases), "timestamp cases;", sum(len(c["chunks"]) for c in chunk_cases), "chunks in total")


if __name__ == "__main__":
    main()
