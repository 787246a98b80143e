"""Known answers made by the REFERENCE'S OWN host-side functions ("next" #6).  Build container only:

    python tests/golden/make_golden_reference_host.py

Several functions on the path's host side execute no MLX op, but live in modules whose top-level `import mlx.core` / `import misaki`
(ordinary ModuleNotFoundError here) stops a plain import.  This script parses those files with `ast`, takes ONLY the named function
definitions (annotations stripped: they name `mx.array` / `en.MToken`), compiles them from the reference's source text AT RUN TIME and calls
them on seeded inputs -- no stand-in modules are installed, nothing of the reference's text is written into this repository; what is
committed is data: inputs and the reference's outputs.

  reference_chunker_cases.json   pipeline.py:163-226 tokens_to_ps / waterfall_last / tokens_to_text / en_tokenize and :292-328 join_timestamps on
                                 seeded random token streams (numpy int32 stands in for pred_dur: the function
                                 only indexes, slices, `.sum()`s and `.item()`s it)
  reference_host_cases.json      base.py:21-34 check_array_shape on every 3-D shape of the Kokoro checkpoint + edge shapes;
                                 kokoro.py:24-44 sanitize_lstm_weights; kokoro.py:172-252 Model.sanitize + istftnet.py:965-979 Decoder.sanitize on
                                 the PyTorch-layout tiny + full checkpoints (numpy arrays: the functions only call `.transpose(0, 2, 1)`):
                                 resulting key -> shape table and a SHA-256 of the values for the tiny one;
                                 voice.py:9-81 load_voice_tensor on a torch.save'd [510, 1, 256] pack (imports no MLX: loaded by path)
"""
import ast
import hashlib
import importlib.util
import json
import logging
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT]
REF = "/root/reference/mlx_audio"

import mlx_audio_amd.params as P  # noqa: E402

# seeded random token streams (text, phonemes, whitespace): punctuation at every tier of the split waterfall, None / empty phonemes, unicode closers
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


def _strip_annotations(fn: ast.FunctionDef):
    for node in ast.walk(fn):
        if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)):
            node.returns = None
            a = node.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                arg.annotation = None
    return fn


def reference_functions(path: str, names, class_name=None, namespace=None) -> dict:
    """Compile the named top-level functions (or methods of `class_name`, rebuilt as a bare class of just those methods) from the reference
    file's source.  Returns the namespace they were executed in."""
    tree = ast.parse(open(path, encoding="utf-8").read())
    body = tree.body
    if class_name:
        cls = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == class_name)
        body = cls.body
    fns = [_strip_annotations(n) for n in body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(f.name for f in fns) == sorted(names), (sorted(f.name for f in fns), names)
    if class_name:
        mod = ast.Module(body=[ast.ClassDef(name=class_name, bases=[], keywords=[], body=fns, decorator_list=[])], type_ignores=[])
    else:
        mod = ast.Module(body=fns, type_ignores=[])
    ast.fix_missing_locations(mod)
    ns = dict(namespace or {})
    exec(compile(mod, path, "exec"), ns)
    return ns


def sha(a) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def chunker():
    ns = reference_functions(f"{REF}/tts/models/kokoro/pipeline.py", ["tokens_to_ps", "waterfall_last", "tokens_to_text", "en_tokenize", "join_timestamps"],
                             class_name="KokoroPipeline", namespace={"logging": logging})
    K = ns["KokoroPipeline"]
    p = K()
    rng = np.random.default_rng(20261004)
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
            K.join_timestamps(objs, np.asarray(pd, np.int32))
            ts_cases.append({"tokens": [[o.text, o.phonemes, o.whitespace] for o in objs], "pred_dur": pd, "ts": [[o.start_ts, o.end_ts] for o in objs]})
    out = {"source": "reference functions compiled from /root/reference/mlx_audio/tts/models/kokoro/pipeline.py:163-226,292-328", "chunk_cases": chunk_cases,
           "timestamp_cases": ts_cases}
    json.dump(out, open(os.path.join(HERE, "reference_chunker_cases.json"), "w"), ensure_ascii=False)
    print(f"chunker: {len(chunk_cases)} chunk cases, {len(ts_cases)} timestamp cases from the reference")


def host():
    out = {"source": "reference functions compiled from mlx_audio/tts/models/base.py:21-34, kokoro/kokoro.py:24-44,172-252, kokoro/istftnet.py:965-979, kokoro/voice.py"}
    cas = reference_functions(f"{REF}/tts/models/base.py", ["check_array_shape"])["check_array_shape"]
    lstm = reference_functions(f"{REF}/tts/models/kokoro/kokoro.py", ["sanitize_lstm_weights"])["sanitize_lstm_weights"]
    dec = reference_functions(f"{REF}/tts/models/kokoro/istftnet.py", ["sanitize"], class_name="Decoder", namespace={"check_array_shape": cas})["Decoder"]
    mdl = reference_functions(f"{REF}/tts/models/kokoro/kokoro.py", ["sanitize"], class_name="Model",
                              namespace={"check_array_shape": cas, "sanitize_lstm_weights": lstm})["Model"]
    model = mdl()
    model.decoder = dec()

    shapes = sorted({tuple(s) for _, s, _ in P.param_inventory(P.kokoro_config(False)) if len(s) == 3}
                    | {(64, 3, 3), (64, 3, 4), (2, 3, 3), (3, 3, 3), (1, 1, 1), (5, 1, 5), (512, 512, 5), (512, 5, 512), (7, 7), (4, 4, 4, 4)})
    out["check_array_shape"] = [[list(s), bool(cas(np.zeros(s, np.int8)))] for s in shapes]

    keys = ["text_encoder.lstm." + k for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
                                                "bias_ih_l0_reverse", "bias_hh_l0_reverse", "weight", "other_l0x")] + ["lstm", "a.b.weight_ih_l0"]
    out["sanitize_lstm_weights"] = [[k, list(lstm(k, 0).keys())[0]] for k in keys]

    for name, cfg in (("tiny", P.tiny_config()), ("full", P.kokoro_config(False))):
        w = P.synth_checkpoint(cfg, 3)
        wt = P.to_torch_layout(w)
        wt["bert.embeddings.position_ids"] = np.arange(512)[None]  # a PyTorch checkpoint carries it; sanitize drops it (kokoro.py:177-179)
        got = model.sanitize(wt)
        table = {k: list(np.shape(v)) for k, v in sorted(got.items())}
        equal = sorted(got) == sorted(w) and all(np.array_equal(got[k], w[k]) for k in w)
        out[f"sanitize_{name}"] = {"seed": 3, "keys_in": len(wt), "table": table, "equals_mlx_layout_checkpoint": bool(equal),
                                   "differing": sorted(k for k in w if k not in got or not np.array_equal(got[k], w[k]))[:50]}
        if name == "tiny":
            out["sanitize_tiny"]["sha"] = {k: sha(np.asarray(v, np.float32)) for k, v in sorted(got.items())}
        print(f"sanitize[{name}]: {len(wt)} PyTorch-layout keys -> {len(got)}; equals the MLX-layout checkpoint: {equal}"
              + ("" if equal else f"  differing: {out[f'sanitize_{name}']['differing'][:8]}"))

    import torch

    spec = importlib.util.spec_from_file_location("_ref_voice", f"{REF}/tts/models/kokoro/voice.py")  # imports io/pickle/zipfile/numpy only
    voice = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(voice)
    pack = torch.randn(510, 1, 256, generator=torch.Generator().manual_seed(9))
    with tempfile.TemporaryDirectory() as td:
        pth = os.path.join(td, "af_test.pt")
        torch.save(pack, pth)
        arr = voice.load_voice_tensor(pth)
    out["voice_pack"] = {"seed": 9, "shape": list(arr.shape), "dtype": str(arr.dtype), "sha": sha(np.asarray(arr, np.float32)),
                         "equals_torch_tensor": bool(np.array_equal(arr, pack.numpy()))}
    print("voice:", out["voice_pack"])
    json.dump(out, open(os.path.join(HERE, "reference_host_cases.json"), "w"), indent=0)


if __name__ == "__main__":
    assert os.path.isdir(REF), "build container only: /root/reference is not mounted"
    chunker()
    host()
