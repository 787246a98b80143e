"""Request-batching serving entry: the `/tts` handler's semantics (mlx_audio/server.py:107-318) without the HTTP layer, with the
one thing the reference's handler cannot do -- chunks of CONCURRENT requests share padded batches.

The reference serves one request at a time: `tts_model.generate(**gen_params)` (server.py:270) walks the request's chunks batch-1 and the
handler concatenates the segments (server.py:277-288).  Here `TTSService.submit()` returns a future at once; a worker thread drains the
queue, phonemises and chunks every waiting request with the request's own pipeline (KokoroPipeline._chunks: the reference's chunking),
pools ALL their chunks, cuts the pool into length-sorted padded batches (KokoroPipeline.plan_batches) and runs one kk_forward_text +
kk_forward_audio per batch (Model.batch_call); each request's segments are put back in text order and concatenated.  A chunk's result
does not depend on its batch neighbours (bit-identical to batch 1 up to the Philox noise stream), so batching is invisible to callers.

Parameter semantics follow the handler: empty text -> 400 "Text is empty"; `speed` is a string parsed as float in [0.5, 2.0] (400
otherwise); `language` names / codes map to a lang_code with the fallback `voice[0]` (server.py:193-220); `voice` None -> the model's
default voice.  Errors are delivered as `TTSError(status, message)` on the future -- what the handler returns as JSON + status code."""
from __future__ import annotations

import queue
import threading
import time
from concurrent.futures import Future
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

LANGUAGE_CODES = {"american_english": "a", "british_english": "b", "spanish": "e", "french": "f", "hindi": "h", "italian": "i", "portuguese": "p",
                  "japanese": "j", "mandarin_chinese": "z"}
LANGUAGE_CODES.update({c: c for c in "abefhipjz"})


class TTSError(Exception):
    """What the handler answers with `JSONResponse({"error": message}, status_code=status)`."""

    def __init__(self, status: int, message: str):
        super().__init__(message)
        self.status, self.message = status, message


@dataclass
class TTSResponse:
    audio: np.ndarray          # float32 [samples]: the request's segments concatenated (server.py:288)
    sample_rate: int
    segments: int
    phonemes: List[str]
    batches: List[int] = field(default_factory=list)  # ids of the device batches this request's chunks rode in (diagnostics)


@dataclass
class _Request:
    text: str
    voice: str
    speed: float
    lang_code: str
    future: Future
    chunks: List[Tuple[str, str]] = field(default_factory=list)  # (graphemes, phonemes) in text order


def parse_request(text: str, voice: Optional[str] = None, speed: str = "1.0", language: str = "a", default_voice: str = "af_heart"):
    """The handler's validation and mapping (server.py:126-163, 193-220) -> (text, voice, speed_value, lang_code); raises TTSError."""
    if not text or not text.strip():
        raise TTSError(400, "Text is empty")
    try:
        value = float(speed)
    except (TypeError, ValueError):
        raise TTSError(400, "Invalid speed value") from None
    if value < 0.5 or value > 2.0:
        raise TTSError(400, "Speed must be between 0.5 and 2.0")
    has_voice = bool(voice and voice.strip())
    lang_code = LANGUAGE_CODES.get(str(language).lower(), voice[0] if has_voice else "a")
    return text, (voice if has_voice else default_voice), value, lang_code


def plan_pool(lengths: List[int], max_batch: int, max_pad: float = 0.25) -> List[List[int]]:
    """Batches over the pooled chunks of every waiting request: KokoroPipeline.plan_batches (length-sorted, at most `max_batch` per batch,
    padding waste bounded by `max_pad`).  Indices refer to the pool."""
    from .pipeline import KokoroPipeline

    return KokoroPipeline.plan_batches(lengths, max_batch, max_pad)


class TTSService:
    def __init__(self, model, max_batch: int = 32, max_wait_ms: float = 4.0, g2p: Optional[Callable] = None, noise_mode: Optional[int] = None,
                 repo_id: Optional[str] = None, start: bool = True, replicas: Sequence = (), contexts: int = 1):
        """model: a loaded kokoro.Model.  max_batch: utterances per device batch.  max_wait_ms: how long the worker waits for more
        requests after the first one of a round (the batching window).  g2p: passed to the pipelines (tests / phoneme input).
        contexts: batches in flight on ONE copy of the weights: the service makes `contexts - 1` further kk_contexts of the model
        (model.new_context(): own graph cache, side stream, workspace; the kk_model is immutable and shared).  replicas: further loaded models
        (e.g. another device's copy).  Every context / model gets a worker thread with its own HIP stream; rounds are dealt to whichever
        worker is free, so the latency-bound text / LSTM phases of one round run under the conv-bound vocoder of another (bench.py's "two batches
        in flight": 43.8 -> 40.0 ms per B = 32 step).  A request's bits do not depend on which worker ran it."""
        from . import _lib

        self.model = model
        self.models = [model, *[model.new_context() for _ in range(max(1, int(contexts)) - 1)], *replicas]
        self.max_batch = int(max_batch)
        self.max_wait = float(max_wait_ms) / 1e3
        self.noise_mode = _lib.NOISE_PHILOX if noise_mode is None else int(noise_mode)
        self._g2p = g2p
        self._repo_id = repo_id or getattr(model, "repo_id", None) or getattr(model, "REPO_ID", "local")
        self._pipes: Dict[Tuple[int, str], object] = {}
        self._q: "queue.Queue[Optional[_Request]]" = queue.Queue()
        self._batch_id = 0
        self.stats = {"requests": 0, "chunks": 0, "batches": 0, "rounds": 0}
        self._threads: List[threading.Thread] = []
        self._collect_lock = threading.Lock()  # one worker at a time gathers a round
        self._state_lock = threading.Lock()    # batch ids, statistics, pipeline cache
        if start:
            self.start()

    # ---- life cycle
    def start(self):
        self._closed = False
        if not self._threads:
            self._threads = [threading.Thread(target=self._worker, args=(k,), name=f"tts-service-{k}", daemon=True) for k in range(len(self.models))]
            for t in self._threads:
                t.start()

    def close(self):
        """Stop the workers.  Requests still queued behind the stop marker are FAILED (503), never dropped: their callers sit in
        Future.result(); submit() after close() fails at once."""
        self._closed = True
        if self._threads:
            self._q.put(None)  # every worker passes the stop marker on
            for t in self._threads:
                t.join()
            self._threads = []
        try:
            while True:
                req = self._q.get_nowait()
                if req is not None and not req.future.done():
                    req.future.set_exception(TTSError(503, "service closed"))
        except queue.Empty:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- the handler's entry
    def submit(self, text: str, voice: Optional[str] = None, speed: str = "1.0", language: str = "a", **_ignored) -> Future:
        fut: Future = Future()
        if getattr(self, "_closed", False):
            fut.set_exception(TTSError(503, "service closed"))
            return fut
        try:
            text, voice, value, lang = parse_request(text, voice, speed, language)
        except TTSError as e:
            fut.set_exception(e)
            return fut
        self._q.put(_Request(text=text, voice=voice, speed=value, lang_code=lang, future=fut))
        return fut

    def tts(self, text: str, **kw) -> TTSResponse:
        """Blocking form of one request (what a handler thread would call)."""
        return self.submit(text, **kw).result()

    # ---- worker
    def _pipeline(self, lang_code: str, k: int = 0):
        from .pipeline import KokoroPipeline

        with self._state_lock:
            if (k, lang_code) not in self._pipes:
                self._pipes[(k, lang_code)] = KokoroPipeline(lang_code=lang_code, model=self.models[k], repo_id=self._repo_id, g2p=self._g2p)
            return self._pipes[(k, lang_code)]

    def _collect(self) -> Optional[List[_Request]]:
        first = self._q.get()
        if first is None:
            self._q.put(None)  # (the next worker stops too)
            return None
        reqs, deadline = [first], time.monotonic() + self.max_wait
        while True:
            left = deadline - time.monotonic()
            try:
                r = self._q.get(timeout=max(left, 0.0)) if left > 0 else self._q.get_nowait()
            except queue.Empty:
                return reqs
            if r is None:
                self._q.put(None)  # finish this round, then stop
                return reqs
            reqs.append(r)

    def run_round(self, reqs: List[_Request], k: int = 0) -> None:
        """One batching round over `reqs` on model k: chunk, pool, batch, synthesise, hand back.  (Public for tests: a deterministic round.)"""
        import torch

        model = self.models[k]

        pool = []  # (request index, position in request, phonemes, style row, speed)
        live = []
        for ri, r in enumerate(reqs):
            try:
                pipe = self._pipeline(r.lang_code, k)
                if pipe.g2p is None:
                    raise TTSError(500, "no G2P available for text input (misaki is not installed); pass g2p= to TTSService")
                pack = pipe.load_voice(r.voice)
                r.chunks = [(gs, ps) for _, gs, ps, _ in pipe._chunks(r.text, r"\n+")]
                if not r.chunks:
                    raise TTSError(500, "No audio generated")
                for ci, (_, ps) in enumerate(r.chunks):
                    pool.append((ri, ci, ps, np.asarray(pack[len(ps) - 1], np.float32).reshape(256), r.speed))
                live.append(ri)
            except TTSError as e:
                r.future.set_exception(e)
            except Exception as e:  # noqa: BLE001  (the handler answers 500 with the message)
                r.future.set_exception(TTSError(500, f"Failed to generate: {e}"))
        outs: Dict[Tuple[int, int], np.ndarray] = {}
        rode: Dict[int, List[int]] = {ri: [] for ri in live}
        failed: Dict[int, Exception] = {}
        for idx in plan_pool([len(p[2]) for p in pool], self.max_batch):
            with self._state_lock:
                self._batch_id += 1
                bid = self._batch_id
                self.stats["batches"] += 1
            try:
                res = model.batch_call([pool[i][2] for i in idx], np.stack([pool[i][3] for i in idx]), [pool[i][4] for i in idx],
                                       noise_mode=self.noise_mode)
                torch.cuda.current_stream().synchronize()  # this worker's stream only: the other workers keep running
                for i, o in zip(idx, res):
                    outs[(pool[i][0], pool[i][1])] = o.audio[0].detach().float().cpu().numpy()
                    if bid not in rode[pool[i][0]]:
                        rode[pool[i][0]].append(bid)
            except Exception as e:  # noqa: BLE001
                for i in idx:
                    failed[pool[i][0]] = e
        for ri in live:
            r = reqs[ri]
            if ri in failed:
                r.future.set_exception(TTSError(500, f"Failed to generate: {failed[ri]}"))
                continue
            segs = [outs[(ri, ci)] for ci in range(len(r.chunks))]
            r.future.set_result(TTSResponse(audio=np.concatenate(segs, axis=0), sample_rate=int(model.sample_rate), segments=len(segs),
                                            phonemes=[ps for _, ps in r.chunks], batches=rode[ri]))
        with self._state_lock:
            self.stats["requests"] += len(reqs)
            self.stats["chunks"] += len(pool)
            self.stats["rounds"] += 1

    def _worker(self, k: int = 0):
        import torch

        # worker 0 keeps the default stream (a single-model service behaves as before); the others get their own
        stream = torch.cuda.Stream() if (k > 0 and torch.cuda.is_available()) else None
        while True:
            with self._collect_lock:
                reqs = self._collect()
            if reqs is None:
                return
            if stream is not None:
                with torch.cuda.stream(stream):
                    self.run_round(reqs, k)
            else:
                self.run_round(reqs, k)
