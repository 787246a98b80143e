"""Host-side mirror of mlx_audio/tts/models/kokoro/pipeline.py + voice.py: voices, chunking, timestamps.

G2P (misaki / espeak) is a third-party dependency of the reference and is not part of the hot path; when it is
not importable the pipeline still serves phoneme strings (`generate_from_tokens`, pipeline.py:238-290) and accepts
any callable `g2p(text) -> (phonemes, tokens)`.
"""
from __future__ import annotations

import json
import logging
import os
import re
from dataclasses import dataclass
from numbers import Number
from typing import Any, Callable, Generator, List, Optional, Tuple, Union

import numpy as np

ALIASES = {"en-us": "a", "en-gb": "b", "es": "e", "fr-fr": "f", "hi": "h", "it": "i", "pt-br": "p", "ja": "j", "zh": "z"}
LANG_CODES = dict(a="American English", b="British English", e="es", f="fr-fr", h="hi", i="it", p="pt-br", j="Japanese",
                  z="Mandarin Chinese")


def load_voice_tensor(path: str) -> np.ndarray:
    """A voice pack as a numpy array [510, 1, 256] (voice.py:9-81).  Formats: PyTorch `.pt` (read with
    `torch.load(weights_only=True)`, which executes nothing from the file), `.npy`, `.npz` (first array),
    `.safetensors` (first tensor) and the `.json` nested lists the reference's Swift package ships."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".pt":
        import torch

        return torch.load(path, map_location="cpu", weights_only=True).float().numpy()
    if ext == ".npy":
        return np.load(path, allow_pickle=False)
    if ext == ".npz":
        z = np.load(path, allow_pickle=False)
        return z[z.files[0]]
    if ext == ".safetensors":
        from safetensors.numpy import load_file

        d = load_file(path)
        return d[sorted(d)[0]]
    if ext == ".json":
        with open(path) as f:
            return np.asarray(json.load(f), dtype=np.float32)
    raise ValueError(f"unsupported voice file: {path}")


PHONEME_LIMIT = 510  # phonemes per forward (512 positions minus <bos>/<eos>, kokoro.py:131-134)
HALF_FRAMES_PER_SECOND = 80
# where a too-long token run may be cut, best first: sentence ends, then clause marks, then commas / dashes; a closing bracket or quote
# directly behind the mark stays with it
SPLIT_TIERS = ("!.?…", ":;", ",—")
CLOSERS = (")", "”")


def _phoneme_text(tokens) -> str:
    """The phoneme string of a token run: each token's phonemes, a blank where the token is followed by white space, ends trimmed."""
    return "".join(f"{t.phonemes} " if t.whitespace else t.phonemes for t in tokens).strip()


def _cut_index(tokens, projected: int, tiers=SPLIT_TIERS, closers=CLOSERS, limit: int = PHONEME_LIMIT) -> int:
    """How many leading tokens to emit when the run plus the incoming token would hold `projected` phonemes: cut behind the LAST mark of the
    best tier whose remainder (projected minus what is emitted) fits the limit; no usable mark -> emit everything."""
    n = len(tokens)
    for tier in tiers:
        marks = frozenset(tier)
        last = max((i for i in range(n) if tokens[i].phonemes in marks), default=-1)
        if last < 0:
            continue
        cut = last + 1
        if cut < n and tokens[cut].phonemes in closers:
            cut += 1
        if projected - len(_phoneme_text(tokens[:cut])) <= limit:
            return cut
    return n


def plan_chunks(tokens, limit: int = PHONEME_LIMIT):
    """Cuts a G2P token stream into runs of at most `limit` phonemes (what one forward accepts), preferring sentence boundaries.
    A token's phonemes are normalised on the way (None -> "", the American flap written as T).  Yields lists of tokens."""
    run, held = [], 0  # `held`: phonemes in `run`, counting the blank behind every token but the stripped ends after a cut
    for tok in tokens:
        tok.phonemes = (tok.phonemes or "").replace("ɾ", "T")
        piece = f"{tok.phonemes} " if tok.whitespace else tok.phonemes
        projected = held + len(piece.rstrip())
        if projected > limit:
            cut = _cut_index(run, projected, limit=limit)
            yield run[:cut]
            run = run[cut:]
            held = len(_phoneme_text(run))
            if not run:
                piece = piece.lstrip()
        run.append(tok)
        held += len(piece)
    if run:
        yield run


class KokoroPipeline:
    """pipeline.py:45-460.  model=False gives a "quiet" pipeline that only phonemises / chunks."""

    def __init__(self, lang_code: str, model, repo_id: str, trf: bool = False, g2p: Optional[Callable] = None):
        lang_code = ALIASES.get(lang_code.lower(), lang_code.lower())
        assert lang_code in LANG_CODES, (lang_code, LANG_CODES)
        self.lang_code = lang_code
        self.repo_id = repo_id
        if repo_id is None:
            raise ValueError("repo_id is required to load voices")
        self.model = model
        self.voices = {}
        self.g2p = g2p
        if g2p is None:
            try:  # optional third-party G2P, exactly what the reference uses (pipeline.py:10,91-126)
                from misaki import en, espeak  # type: ignore

                if lang_code in "ab":
                    try:
                        fallback = espeak.EspeakFallback(british=lang_code == "b")
                    except Exception as e:  # noqa: BLE001
                        logging.warning("EspeakFallback not Enabled: OOD words will be skipped (%s)", e)
                        fallback = None
                    self.g2p = en.G2P(trf=trf, british=lang_code == "b", fallback=fallback, unk="")
                else:
                    self.g2p = espeak.EspeakG2P(language=LANG_CODES[lang_code])
            except ImportError:
                self.g2p = None  # phoneme-string entry points still work

    # ---- voices (pipeline.py:129-161) ------------------------------------------------------------------
    def load_single_voice(self, voice: str):
        if voice in self.voices:
            return self.voices[voice]
        if os.path.exists(voice):
            f = voice
        else:
            from huggingface_hub import hf_hub_download  # network; raises offline, like the reference

            f = hf_hub_download(repo_id=self.repo_id, filename=f"voices/{voice}.pt")
            if not voice.startswith(self.lang_code):
                logging.warning("Language mismatch, loading %s voice into %s pipeline.", voice, LANG_CODES.get(self.lang_code))
        pack = np.asarray(load_voice_tensor(f), dtype=np.float32)
        self.voices[voice] = pack
        return pack

    def load_voice(self, voice: str, delimiter: str = ","):
        """One voice or the mean of several ('af_bella,af_jessica')."""
        if voice in self.voices:
            return self.voices[voice]
        packs = [self.load_single_voice(v) for v in voice.split(delimiter)]
        if len(packs) == 1:
            return packs[0]
        self.voices[voice] = np.mean(np.stack(packs), axis=0)
        return self.voices[voice]

    # ---- chunk planning on duck-typed tokens (.text, .phonemes, .whitespace); behaviour of pipeline.py:163-226, pinned by
    # tests/golden/reference_chunker_cases.json -----------------------------------------------------------------------------------------
    @classmethod
    def tokens_to_ps(cls, tokens) -> str:
        return _phoneme_text(tokens)

    @classmethod
    def tokens_to_text(cls, tokens) -> str:
        return "".join(f"{t.text}{t.whitespace}" for t in tokens).strip()

    @classmethod
    def waterfall_last(cls, tokens, next_count: int, waterfall=SPLIT_TIERS, bumps=CLOSERS) -> int:
        return _cut_index(tokens, next_count, waterfall, bumps)

    def en_tokenize(self, tokens) -> Generator[Tuple[str, str, list], None, None]:
        for part in plan_chunks(tokens):
            yield KokoroPipeline.tokens_to_text(part), _phoneme_text(part), part

    # ---- inference ---------------------------------------------------------------------------------------
    @classmethod
    def infer(cls, model, ps: str, pack, speed: Number = 1):
        # style row = pack[len(ps) - 1]: indexed by the phoneme-STRING length (pipeline.py:236)
        return model(ps, pack[len(ps) - 1], speed, return_output=True)

    def generate_from_tokens(self, tokens: Union[str, list], voice: str, speed: Number = 1, model=None):
        model = model or self.model
        if model and voice is None:
            raise ValueError('Specify a voice: pipeline.generate_from_tokens(..., voice="af_heart")')
        pack = self.load_voice(voice) if model else None
        if isinstance(tokens, str):
            if len(tokens) > 510:
                raise ValueError(f"Phoneme string too long: {len(tokens)} > 510")
            output = KokoroPipeline.infer(model, tokens, pack, speed) if model else None
            yield self.Result(graphemes="", phonemes=tokens, output=output)
            return
        for gs, ps, tks in self.en_tokenize(tokens):
            if not ps:
                continue
            if len(ps) > 510:
                logging.warning("Unexpected len(ps) == %d > 510; truncating", len(ps))
                ps = ps[:510]
            output = KokoroPipeline.infer(model, ps, pack, speed) if model else None
            if output is not None and output.pred_dur is not None:
                KokoroPipeline.join_timestamps(tks, output.pred_dur)
            yield self.Result(graphemes=gs, phonemes=ps, tokens=tks, output=output)

    @classmethod
    def join_timestamps(cls, tokens, pred_dur) -> None:
        """Word-level start_ts / end_ts from the per-phoneme durations (behaviour of pipeline.py:292-328, pinned by
        tests/golden/reference_chunker_cases.json).  Time is kept in HALF frames (a frame is 1/40 s, so 80 per second) so that the pause a
        space character stands for can be split evenly between the word before and the word after it."""
        frames = [int(v) for v in (pred_dur.tolist() if hasattr(pred_dur, "tolist") else pred_dur)]
        if not tokens or len(frames) < 3:  # <bos>, at least one phoneme, <eos>
            return
        eos = len(frames) - 1
        # `opened`: where the next word starts; `closed`: where the previous word's share of the following pause ends
        opened = closed = 2 * max(0, frames[0] - 3)
        at = 1  # index of the next unconsumed duration
        for tok in tokens:
            if at >= eos:
                return
            width = len(tok.phonemes) if tok.phonemes else 0
            if width == 0:
                if tok.whitespace:  # a token that is only a pause: the duration AFTER the current one is charged twice, two entries are consumed
                    pause = frames[at + 1]
                    opened = closed + pause
                    closed = opened + pause
                    at += 2
                continue
            nxt = at + width
            if nxt > eos:
                return
            spoken = sum(frames[at:nxt])
            pause = frames[nxt] if tok.whitespace else 0
            tok.start_ts = opened / HALF_FRAMES_PER_SECOND
            opened = closed + 2 * spoken + pause
            tok.end_ts = opened / HALF_FRAMES_PER_SECOND
            closed = opened + pause
            at = nxt + (1 if tok.whitespace else 0)

    @dataclass
    class Result:
        graphemes: str
        phonemes: str
        tokens: Optional[list] = None
        output: Optional[Any] = None
        text_index: Optional[int] = None

        @property
        def audio(self):
            return None if self.output is None else self.output.audio

        @property
        def pred_dur(self):
            return None if self.output is None else self.output.pred_dur

        def __iter__(self):
            yield self.graphemes
            yield self.phonemes
            yield self.audio

        def __getitem__(self, index):
            return [self.graphemes, self.phonemes, self.audio][index]

        def __len__(self):
            return 3

    def _chunks(self, text: Union[str, List[str]], split_pattern: Optional[str]):
        """The reference's chunking of pipeline.py:371-436 as a flat stream of (text_index, graphemes, phonemes, tokens | None)."""
        if isinstance(text, str):
            text = re.split(split_pattern, text.strip()) if split_pattern else [text]
        for gi, graphemes in enumerate(text):
            if not graphemes.strip():
                continue
            if self.lang_code in "ab":
                _, tokens = self.g2p(graphemes)
                for gs, ps, tks in self.en_tokenize(tokens):
                    if not ps:
                        continue
                    yield gi, gs, ps[:510], tks
            else:
                for chunk in _sentence_chunks(graphemes, 400):  # pipeline.py:408-436
                    ps, _ = self.g2p(chunk)
                    if not ps:
                        continue
                    yield gi, chunk, ps[:510], None

    @staticmethod
    def plan_batches(lengths: List[int], batch_size: int, max_pad: float = 0.25) -> List[List[int]]:
        """Batch scheduler of the chunk stream: chunks sorted by phoneme count and cut into batches of at most `batch_size` whose shortest
        member is no more than `max_pad` shorter than the longest (padded tokens and frames are wasted work on the GPU; a batch's cost is
        set by its longest utterance).  Returns lists of chunk indices; every index appears exactly once."""
        order = sorted(range(len(lengths)), key=lambda i: (-lengths[i], i))
        batches, cur = [], []
        for i in order:
            if cur and (len(cur) >= batch_size or lengths[i] < (1.0 - max_pad) * lengths[cur[0]]):
                batches.append(cur)
                cur = []
            cur.append(i)
        if cur:
            batches.append(cur)
        return batches

    def __call__(self, text: Union[str, List[str]], voice: Optional[str] = None, speed: Number = 1,
                 split_pattern: Optional[str] = r"\n+", batch_size: int = 1):
        """pipeline.py:358-436.  batch_size > 1 (an addition: the reference synthesises chunk by chunk, batch 1): all chunks of the request are
        phonemised first, grouped by `plan_batches` and synthesised as padded batches (`Model.batch_call`: every utterance's result is bit-identical
        to its batch-1 result up to the noise seed); results are still yielded in text order."""
        if voice is None:
            raise ValueError('Specify a voice: en_us_pipeline(text="Hello world!", voice="af_heart")')
        if self.g2p is None:
            raise ImportError("text input needs a G2P: `pip install misaki[en]` (what the reference uses) or pass g2p=...; "
                              "phoneme strings work through generate_from_tokens()")
        pack = self.load_voice(voice) if self.model else None
        if batch_size <= 1 or not self.model:
            for gi, gs, ps, tks in self._chunks(text, split_pattern):
                output = KokoroPipeline.infer(self.model, ps, pack, speed) if self.model else None
                if tks is not None and output is not None and output.pred_dur is not None:
                    KokoroPipeline.join_timestamps(tks, output.pred_dur)
                yield self.Result(graphemes=gs, phonemes=ps, tokens=tks, output=output, text_index=gi)
            return
        chunks = list(self._chunks(text, split_pattern))
        outputs = [None] * len(chunks)
        for idx in KokoroPipeline.plan_batches([len(c[2]) for c in chunks], batch_size):
            ps_list = [chunks[i][2] for i in idx]
            rows = np.stack([np.asarray(pack[len(ps) - 1], np.float32).reshape(256) for ps in ps_list])  # pipeline.py:236, one row per chunk
            for i, o in zip(idx, self.model.batch_call(ps_list, rows, speed)):
                outputs[i] = o
        for (gi, gs, ps, tks), output in zip(chunks, outputs):
            if tks is not None and output.pred_dur is not None:
                KokoroPipeline.join_timestamps(tks, output.pred_dur)
            yield self.Result(graphemes=gs, phonemes=ps, tokens=tks, output=output, text_index=gi)


def _sentence_chunks(graphemes: str, chunk_size: int) -> List[str]:
    sentences = re.split(r"([.!?]+)", graphemes)
    chunks, cur = [], ""
    for i in range(0, len(sentences), 2):
        s = sentences[i] + (sentences[i + 1] if i + 1 < len(sentences) else "")
        if len(cur) + len(s) <= chunk_size:
            cur += s
        else:
            if cur:
                chunks.append(cur.strip())
            cur = s
    if cur:
        chunks.append(cur.strip())
    if not chunks:
        chunks = [graphemes[i : i + chunk_size] for i in range(0, len(graphemes), chunk_size)]
    return [c for c in chunks if c.strip()]
