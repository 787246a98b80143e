"""Host mirror of the reference's CSM generation loop (mlx_audio/tts/models/sesame/sesame.py:484-541 prompt frames, :689-817 loop,
:619-680 result) on top of the HIP frame generator (csm.py) and the HIP Mimi codec (mimi.py).

The reference tokenises text with the Llama-3.2 tokenizer (`AutoTokenizer.from_pretrained("unsloth/Llama-3.2-1B")`, sesame.py:427-431),
which cannot be fetched offline: this mirror takes TOKEN IDS where the reference takes strings; everything after the tokenizer is the
same data flow.  The watermark (sesame.py:631-642, third-party `silentcipher`) is not applied.  Unlike the reference (batch 1) the loop
runs B streams with equally long prompts in one batch."""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from .csm import SesameModel
from .mimi import Mimi


@dataclass
class Segment:
    """sesame.py:418-424, with the text already tokenised (`[speaker]text` -> ids)."""
    speaker: int
    text_ids: Sequence[int]
    audio: Optional[np.ndarray] = None  # mono float32 at 24 kHz


@dataclass
class GenerationResult:
    audio: torch.Tensor  # [B, samples]
    samples: int
    sample_rate: int
    token_count: int
    processing_time_seconds: float
    real_time_factor: float  # wall / audio seconds of ONE stream (the reference's definition, sesame.py:656)


class Model:
    def __init__(self, csm: SesameModel, mimi: Mimi):
        self.model = csm
        self._audio_tokenizer = mimi
        self.n_cb = csm.cfg["audio_num_codebooks"]
        self.sample_rate = 24000

    # ---- sesame.py:484-541
    def _tokenize_text_ids(self, ids: Sequence[int]):
        f = np.zeros((len(ids), self.n_cb + 1), np.int32)
        m = np.zeros((len(ids), self.n_cb + 1), np.float32)
        f[:, -1] = np.asarray(ids, np.int32)
        m[:, -1] = 1
        return f, m

    def _tokenize_audio(self, audio: np.ndarray, add_eos: bool = True):
        codes = self._audio_tokenizer.encode(torch.tensor(np.asarray(audio, np.float32))[None, None])[0].cpu().numpy()  # (K, T)
        if add_eos:
            codes = np.concatenate([codes, np.zeros((codes.shape[0], 1), codes.dtype)], axis=1)
        f = np.zeros((codes.shape[1], self.n_cb + 1), np.int32)
        m = np.zeros((codes.shape[1], self.n_cb + 1), np.float32)
        f[:, :-1] = codes.T
        m[:, :-1] = 1
        return f, m

    def _tokenize_segment(self, seg: Segment, add_eos: bool = True):
        tf, tm = self._tokenize_text_ids(seg.text_ids)
        if seg.audio is None:
            return tf, tm
        af, am = self._tokenize_audio(seg.audio, add_eos=add_eos)
        return np.concatenate([tf, af], 0), np.concatenate([tm, am], 0)

    # ---- sesame.py:689-817 (non-streaming branch)
    def _prompt(self, contexts, prompts_ids, max_audio_length_ms):
        toks, masks = [], []
        for ctx, pid in zip(contexts, prompts_ids):
            ft, fm = [], []
            for seg in ctx:
                a, b = self._tokenize_segment(seg, add_eos=True)
                ft.append(a)
                fm.append(b)
            a, b = self._tokenize_text_ids(pid)
            ft.append(a)
            fm.append(b)
            toks.append(np.concatenate(ft, 0))
            masks.append(np.concatenate(fm, 0))
        S = toks[0].shape[0]
        if any(t.shape[0] != S for t in toks):
            raise ValueError("all streams of a batch must have prompts of the same length")
        max_audio_frames = int(max_audio_length_ms / 80)
        max_seq_len = self.model.cfg["max_seq_len"] - max_audio_frames
        if S >= max_seq_len:
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {max_seq_len}")  # sesame.py:755-758
        return np.stack(toks), np.stack(masks), max_audio_frames

    def generate_stream(self, contexts: List[List[Segment]], prompts_ids: List[Sequence[int]], max_audio_length_ms: float = 90_000,
                        temperature: float = 0.9, top_k: int = 50, seed: Optional[int] = 0, stop_on_eos: bool = True, streaming_interval: float = 2.0):
        """`generate(..., stream=True)` of the reference (sesame.py:689-817): the same frame loop, but every `streaming_interval` seconds of
        generated frames (sesame.py:719-721: int(streaming_interval * 12.5) frames) are decoded INCREMENTALLY by MimiStreamingDecoder
        (`generate_result(..., stream=True)`, sesame.py:619-629) and yielded as a partial GenerationResult."""
        from .mimi import MimiStreamingDecoder

        B = len(prompts_ids)
        tok, msk, max_audio_frames = self._prompt(contexts, prompts_ids, max_audio_length_ms)
        interval = max(1, int(streaming_interval * 12.5))
        dev = self.model.device
        self.model.reset_caches()
        self.model.set_graph_mode(True)
        decoder = MimiStreamingDecoder(self._audio_tokenizer)
        curr, cmask = torch.tensor(tok, device=dev), torch.tensor(msk, device=dev)
        rng = np.random.default_rng(seed) if seed is not None else None
        step_mask = torch.zeros((B, 1, self.n_cb + 1), dtype=torch.float32, device=dev)
        step_mask[:, 0, : self.n_cb] = 1
        samples, start = [], time.perf_counter()
        done = torch.zeros(B, dtype=torch.bool, device=dev)

        def result(frames):
            audio = decoder.decode_frames(torch.stack(frames, dim=2))[:, 0]
            torch.cuda.synchronize()
            dt = time.perf_counter() - start
            secs = audio.shape[1] / self.sample_rate
            return GenerationResult(audio=audio, samples=int(audio.shape[1]), sample_rate=self.sample_rate, token_count=len(frames),
                                    processing_time_seconds=dt, real_time_factor=dt / secs if secs > 0 else 0.0)

        for _ in range(max_audio_frames):
            u = torch.tensor(rng.uniform(size=(B, self.n_cb)).astype(np.float32), device=dev) if rng is not None and temperature > 0 else None
            sample = self.model.generate_frame(curr, cmask, temperature=temperature, top_k=top_k, uniforms=u)
            if stop_on_eos:
                done |= (sample == 0).all(dim=1)
                if bool(done.all()):
                    break
            samples.append(sample.clone())
            curr = torch.zeros((B, 1, self.n_cb + 1), dtype=torch.int32, device=dev)
            curr[:, 0, : self.n_cb] = sample
            cmask = step_mask
            if len(samples) >= interval:
                yield result(samples)
                samples, start = [], time.perf_counter()
        if samples:
            yield result(samples)

    def generate(self, contexts: List[List[Segment]], prompts_ids: List[Sequence[int]], speaker: int = 0, max_audio_length_ms: float = 90_000,
                 temperature: float = 0.9, top_k: int = 50, seed: Optional[int] = 0, stop_on_eos: bool = True) -> GenerationResult:
        """One entry of `contexts` / `prompts_ids` per stream.  All prompts must assemble to the same number of frames."""
        B = len(prompts_ids)
        start = time.perf_counter()
        toks, masks = [], []
        for ctx, pid in zip(contexts, prompts_ids):
            ft, fm = [], []
            for seg in ctx:
                a, b = self._tokenize_segment(seg, add_eos=True)
                ft.append(a)
                fm.append(b)
            a, b = self._tokenize_text_ids(pid)
            ft.append(a)
            fm.append(b)
            toks.append(np.concatenate(ft, 0))
            masks.append(np.concatenate(fm, 0))
        S = toks[0].shape[0]
        if any(t.shape[0] != S for t in toks):
            raise ValueError("all streams of a batch must have prompts of the same length")
        max_audio_frames = int(max_audio_length_ms / 80)
        max_seq_len = self.model.cfg["max_seq_len"] - max_audio_frames
        if S >= max_seq_len:
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {max_seq_len}")  # sesame.py:755-758
        dev = self.model.device
        self.model.reset_caches()
        self.model.set_graph_mode(True)  # the frame steps after the prompt block are replayed as one hipGraph
        curr = torch.tensor(np.stack(toks), device=dev)
        cmask = torch.tensor(np.stack(masks), device=dev)
        rng = np.random.default_rng(seed) if seed is not None else None
        step_mask = torch.zeros((B, 1, self.n_cb + 1), dtype=torch.float32, device=dev)
        step_mask[:, 0, : self.n_cb] = 1
        samples = []
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        for _ in range(max_audio_frames):
            u = torch.tensor(rng.uniform(size=(B, self.n_cb)).astype(np.float32), device=dev) if rng is not None and temperature > 0 else None
            sample = self.model.generate_frame(curr, cmask, temperature=temperature, top_k=top_k, uniforms=u)
            if stop_on_eos:
                done |= (sample == 0).all(dim=1)  # an all-zero frame is EOS (sesame.py:765-766)
                if bool(done.all()):
                    break
            samples.append(sample.clone())  # (graph replay hands back a view of a persistent buffer)
            curr = torch.zeros((B, 1, self.n_cb + 1), dtype=torch.int32, device=dev)
            curr[:, 0, : self.n_cb] = sample
            cmask = step_mask
        if not samples:
            raise AssertionError("No audio generated")
        codes = torch.stack(samples, dim=2)  # [B, K, T]  (mx.transpose(mx.stack(samples), [1, 2, 0]), sesame.py:623)
        audio = self._audio_tokenizer.decode(codes)[:, 0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - start
        secs = audio.shape[1] / self.sample_rate
        return GenerationResult(audio=audio, samples=int(audio.shape[1]), sample_rate=self.sample_rate, token_count=len(samples),
                                processing_time_seconds=dt, real_time_factor=dt / secs if secs > 0 else 0.0)
