"""Host mirror of the reference's CSM model class (mlx_audio/tts/models/sesame/sesame.py:440-817): `Model(config)` with `sanitize`,
`generate(text, voice, speaker, context, ..., ref_audio, ref_text, stream, streaming_interval, voice_match)` yielding `GenerationResult`s,
on top of the HIP frame generator (csm.py) and the HIP Mimi codec (mimi.py).

What differs from the reference, and why:
  * TEXT MAY BE PRE-TOKENISED.  The reference tokenises `f"[{speaker}]{text}"` with the Llama-3.2 tokenizer
    (`AutoTokenizer.from_pretrained("unsloth/Llama-3.2-1B")`, sesame.py:427-438,484-489), whose files cannot be fetched offline.  Every
    place that takes a string also takes the token ids that tokenizer would produce (a sequence of ints); strings need a tokenizer
    directory on disk (`config["text_tokenizer"]` pointing at a local path, loaded with `local_files_only=True`).
  * The default speaker prompts (`default_speaker_prompt`, sesame.py:583-617) are downloaded by the reference; offline, `voice` must name a
    local `.wav` next to a `.txt` (or pass `ref_audio` / `ref_text` / `context`).
  * The watermark (sesame.py:631-642, third-party `silentcipher`) is not applied.
  * `generate_batch` is an addition: B independent streams in one batch with prompts of DIFFERENT lengths (left-padded, per-item positions:
    kk_csm_set_padding), each stream's result bit-identical to running it alone; per-stream EOS is tracked on the device and every stream is
    trimmed to its own length.  `generate` itself is the reference's batch-1 loop on top of it.
"""
from __future__ import annotations

import re
import time
from dataclasses import dataclass
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Union

import numpy as np
import torch

from .base import GenerationResult
from .csm import SesameModel
from .mimi import Mimi, MimiStreamingDecoder

TextLike = Union[str, Sequence[int]]



@dataclass(frozen=True)
class Sampler:
    """What the frame loop needs of mlx_lm's `make_sampler(temp, top_k)` (sesame.py:14-17,719): the two numbers.  temp 0 = argmax."""
    temp: float = 0.9
    top_k: int = 50


def make_sampler(temp: float = 0.9, top_k: int = 50, **_unused) -> Sampler:
    return Sampler(float(temp), int(top_k))


@dataclass
class Segment:
    """sesame.py:418-424; `text` may be a string or the token ids of `[speaker]text` (see the module docstring)."""
    speaker: int
    text: TextLike
    audio: Optional[np.ndarray] = None  # mono float32 at 24 kHz

    @property
    def text_ids(self):  # (round-1 name)
        return self.text


@dataclass
class BatchResult:
    """Result of `generate_batch`: `audio[b]` is stream b's waveform trimmed to its own length."""
    audio: List[torch.Tensor]
    frames: List[int]          # frames generated per stream (up to, not including, its EOS frame)
    sample_rate: int
    processing_time_seconds: float
    real_time_factor: float    # wall / audio seconds of the longest stream (the reference's per-stream definition, sesame.py:656)
    codes: Optional[torch.Tensor] = None  # [B, n_cb, T] as generated (frames past a stream's EOS are whatever the loop produced)


def _is_ids(t) -> bool:
    return not isinstance(t, str)


class Model:
    """sesame.py:440-817.  `Model(config)` builds the frame generator from `config` (the checkpoint's config.json: backbone / decoder flavours
    or explicit sizes); weights arrive through `load_weights` (load_model does that).  The codec is passed in (`mimi=`) or loaded from
    `config["mimi_path"]` (a directory with the codec's safetensors in the MLX layout) -- the reference downloads it (sesame.py:456)."""

    def __init__(self, config, mimi: Optional[Mimi] = None, weights: Optional[Dict[str, np.ndarray]] = None, weight_dtype: str = "float32",
                 csm: Optional[SesameModel] = None):
        if isinstance(config, SesameModel):  # round-1 signature Model(csm, mimi)
            csm, config = config, config.cfg
        self.config = config
        self._weight_dtype = weight_dtype
        self.model = csm
        self._audio_tokenizer = mimi
        self._streaming_decoder = None
        if self.model is None and weights is not None:
            self.load_weights(weights)
        if self._audio_tokenizer is None and isinstance(config, dict) and config.get("mimi_path"):
            self._audio_tokenizer = _load_mimi(config["mimi_path"], int(self._cfg()["audio_num_codebooks"]))
        self.tokenizer_repo = config.get("text_tokenizer") if isinstance(config, dict) else None
        self._text_tokenizer = None
        if self.tokenizer_repo:
            self._text_tokenizer = _load_llama3_tokenizer(self.tokenizer_repo)
        self._watermarker = None
        self.sample_rate = 24000
        self._seed_counter = 0

    def share(self) -> "Model":
        """A second Model on the SAME device weights (kk_csm_share + Mimi.share): own KV caches, graph cache, codec workspace / streams --
        one per job in flight (own HIP stream / host thread).  The reference's Model is single threaded."""
        import copy

        other = copy.copy(self)
        other.model = self.model.share()
        other._audio_tokenizer = self._audio_tokenizer.share() if self._audio_tokenizer is not None else None
        other._streaming_decoder = None
        other._seed_counter = 0
        return other

    # ---- config / weights ------------------------------------------------------------------------------------------------------------
    def _cfg(self) -> dict:
        return csm_config_from(self.config)

    @property
    def n_cb(self) -> int:
        return int(self._cfg()["audio_num_codebooks"])

    def sanitize(self, weights):
        """sesame.py:543-569: torchtune-style names -> mlx_lm names, every key under `model.`."""
        out = {}
        for k, v in weights.items():
            if not k.startswith("model."):
                k = "model." + k
            if "attn" in k and "self_attn" not in k:
                k = k.replace("attn", "self_attn").replace("output_proj", "o_proj")
            if "mlp" in k:
                k = k.replace("w1", "gate_proj").replace("w2", "down_proj").replace("w3", "up_proj")
            if "sa_norm" in k or "mlp_norm" in k:
                k = k.replace("sa_norm", "input_layernorm").replace("mlp_norm", "post_attention_layernorm").replace("scale", "weight")
            if "decoder.norm" in k or "backbone.norm" in k:
                k = k.replace("scale", "weight")
            out[k] = v
        return out

    def model_quant_predicate(self, p, m, config):
        return not p.startswith("_audio_tokenizer")  # sesame.py:470-474

    def load_weights(self, weights, strict: bool = True):
        """Accepts checkpoint names with or without the `model.` prefix, sanitized or not; codec tensors (`_audio_tokenizer.*`) are split off."""
        items = dict(weights.items() if hasattr(weights, "items") else weights)
        items = self.sanitize(items)
        csm_w, mimi_w = {}, {}
        for k, v in items.items():
            k = k[len("model."):] if k.startswith("model.") else k
            a = v.float().numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)
            if k.startswith("_audio_tokenizer."):
                mimi_w[k[len("_audio_tokenizer."):]] = a
            else:
                csm_w[k] = a
        self.model = SesameModel(self._cfg(), csm_w, weight_dtype=self._weight_dtype)
        if mimi_w and self._audio_tokenizer is None:
            from .mimi import mimi_202407

            self._audio_tokenizer = Mimi(mimi_202407(self.n_cb), mimi_w)
        return self

    # ---- prompt frames (sesame.py:484-541) -------------------------------------------------------------------------------------------------
    def _text_ids(self, text: TextLike, speaker: int) -> List[int]:
        if _is_ids(text):
            return [int(t) for t in text]
        if self._text_tokenizer is None:
            raise ValueError("text was given as a string but no tokenizer is available offline: pass the Llama-3.2 token ids of "
                             f"'[{speaker}]<text>' instead, or set config['text_tokenizer'] to a local tokenizer directory")
        return [int(t) for t in self._text_tokenizer.encode(f"[{speaker}]{text}")]

    def _tokenize_text_segment(self, text: TextLike, speaker: int):
        ids = self._text_ids(text, speaker)
        f = np.zeros((len(ids), self.n_cb + 1), np.int32)
        m = np.zeros((len(ids), self.n_cb + 1), np.float32)
        f[:, -1] = np.asarray(ids, np.int32)
        m[:, -1] = 1
        return f, m

    def _tokenize_text_ids(self, ids):  # (round-1 name)
        return self._tokenize_text_segment(list(ids), 0)

    def encode_audios(self, audios: Sequence[np.ndarray]) -> List[np.ndarray]:
        """Mimi codes (K, T) of several clips; clips of equal length go through ONE `Mimi.encode` call (an addition: the reference encodes
        one clip per call, sesame.py:500-510; a batch item's codes do not depend on its neighbours).  Clips of different lengths are not
        padded into one call: the encoder transformer sees the whole clip (no mask), so padding would change the codes."""
        if self._audio_tokenizer is None:
            raise ValueError("reference audio needs the Mimi codec: pass mimi= or config['mimi_path']")
        out: List[Optional[np.ndarray]] = [None] * len(audios)
        groups: Dict[int, List[int]] = {}
        for i, a in enumerate(audios):
            groups.setdefault(int(np.asarray(a).shape[-1]), []).append(i)
        for n, idx in groups.items():
            batch = np.stack([np.asarray(audios[i], np.float32).reshape(n) for i in idx])[:, None, :]
            codes = self._audio_tokenizer.encode(torch.tensor(batch)).cpu().numpy()  # (len(idx), K, T)
            for j, i in enumerate(idx):
                out[i] = codes[j]
        return out  # type: ignore[return-value]

    def _tokenize_audio(self, audio: np.ndarray, add_eos: bool = True, codes: Optional[np.ndarray] = None):
        if codes is None:
            codes = self.encode_audios([audio])[0]  # (K, T)
        if add_eos:
            codes = np.concatenate([codes, np.zeros((codes.shape[0], 1), codes.dtype)], axis=1)
        f = np.zeros((codes.shape[1], self.n_cb + 1), np.int32)
        m = np.zeros((codes.shape[1], self.n_cb + 1), np.float32)
        f[:, :-1] = codes.T
        m[:, :-1] = 1
        return f, m

    def _tokenize_segment(self, seg: Segment, add_eos: bool = True, codes: Optional[np.ndarray] = None):
        tf, tm = self._tokenize_text_segment(seg.text, seg.speaker)
        if seg.audio is None:
            return tf, tm
        af, am = self._tokenize_audio(seg.audio, add_eos=add_eos, codes=codes)
        return np.concatenate([tf, af], 0), np.concatenate([tm, am], 0)

    def prompt_frames_batch(self, contexts: Sequence[Sequence[Segment]], texts: Sequence[Optional[TextLike]], speaker: int = 0, voice_match: bool = False):
        """`prompt_frames` for several streams with every reference clip of every stream encoded in as few `Mimi.encode` calls as their lengths
        allow (`encode_audios`); the same prompts, bit for bit, as one `prompt_frames` call per stream."""
        clips = [seg.audio for ctx in contexts for seg in (ctx[:1] if voice_match else ctx) if seg.audio is not None]
        codes = {id(a): c for a, c in zip(clips, self.encode_audios(clips))} if clips else {}
        return [self.prompt_frames(ctx, text, speaker, voice_match, _codes=codes) for ctx, text in zip(contexts, texts)]

    def prompt_frames(self, context: Sequence[Segment], text: Optional[TextLike], speaker: int = 0, voice_match: bool = False,
                      _codes: Optional[Dict[int, np.ndarray]] = None):
        """The prompt of one stream as (tokens [S, n_cb+1] int32, mask [S, n_cb+1] float32) -- sesame.py:727-757.
        voice_match (the reference's default): ONE segment whose text is `context[0].text + " " + text` and whose audio is the context's,
        without an EOS frame (the model continues the speaker's audio).  Otherwise: every context segment (with EOS frames), then the text."""
        if voice_match:
            if not context:
                raise ValueError("voice_match needs a context segment")
            c0 = context[0]
            if text is None:
                joined = c0.text
            elif _is_ids(c0.text) or _is_ids(text):
                joined = list(self._text_ids(c0.text, speaker)) + list(self._text_ids(text, speaker))
            else:
                joined = (c0.text + " " + text).strip()
            return self._tokenize_segment(Segment(speaker=speaker, text=joined, audio=c0.audio), add_eos=False,
                                          codes=(_codes or {}).get(id(c0.audio)))
        ft, fm = [], []
        for seg in context:
            a, b = self._tokenize_segment(seg, add_eos=True, codes=(_codes or {}).get(id(seg.audio)))
            ft.append(a)
            fm.append(b)
        if text is not None:
            a, b = self._tokenize_text_segment(text, speaker)
            ft.append(a)
            fm.append(b)
        return np.concatenate(ft, 0), np.concatenate(fm, 0)

    # ---- the frame loop over B streams ------------------------------------------------------------------------------------------------------
    def _frame_loop(self, prompts, max_audio_frames: int, temperature: float, top_k: int, seed: Optional[int], stop_on_eos: bool,
                    uniforms: Optional[Callable[[int], np.ndarray]] = None):
        """Generator over frames: yields (codes [B, n_cb] int32 on the device, done [B] bool: streams whose EOS frame has been seen BEFORE this
        frame).  Ragged prompts are left-padded; the loop ends when every stream has produced its EOS frame (an all-zero frame, sesame.py:765)."""
        B = len(prompts)
        lens = [p[0].shape[0] for p in prompts]
        S = max(lens)
        max_seq_len = self.model.cfg["max_seq_len"] - max_audio_frames
        if S >= max_seq_len:
            raise ValueError(f"Inputs too long, must be below max_seq_len - max_audio_frames: {max_seq_len}")  # sesame.py:755-758
        n = self.n_cb
        tok = np.zeros((B, S, n + 1), np.int32)
        msk = np.zeros((B, S, n + 1), np.float32)
        for b, (t, m) in enumerate(prompts):
            tok[b, S - lens[b]:] = t
            msk[b, S - lens[b]:] = m
        dev = self.model.device
        if not self.model.caches_are_enabled() or self.model.max_batch < B:
            self.model.setup_caches(B)
        self.model.reset_caches()
        if any(l != S for l in lens):
            self.model.set_padding([S - l for l in lens])
        self.model.set_graph_mode(True)  # the frame steps after the prompt block are replayed as one hipGraph
        # make_sampler(temp, top_k) draws from MLX's global RNG; here the uniforms are explicit: seeded, or fresh entropy when seed is None
        rng = np.random.default_rng(seed) if (temperature > 0 and uniforms is None) else None
        curr, cmask = torch.tensor(tok, device=dev), torch.tensor(msk, device=dev)
        step_mask = torch.zeros((B, 1, n + 1), dtype=torch.float32, device=dev)
        step_mask[:, 0, :n] = 1
        done = torch.zeros(B, dtype=torch.bool, device=dev)
        for i in range(max_audio_frames):
            u = None
            if temperature > 0:
                u = torch.tensor(np.asarray(uniforms(i) if uniforms is not None else rng.uniform(size=(B, n)), np.float32), device=dev)
            sample = self.model.generate_frame(curr, cmask, temperature=temperature, top_k=top_k, uniforms=u)
            was_done = done
            if stop_on_eos:
                done = done | (sample == 0).all(dim=1)  # an all-zero frame is EOS (sesame.py:765-766)
            yield sample.clone(), was_done, done  # (graph replay hands back a view of a persistent buffer)
            curr = torch.zeros((B, 1, n + 1), dtype=torch.int32, device=dev)
            curr[:, 0, :n] = sample
            cmask = step_mask

    def generate_batch(self, prompts, max_audio_length_ms: float = 90_000, temperature: float = 0.9, top_k: int = 50, seed: Optional[int] = 0,
                       stop_on_eos: bool = True, eos_check_interval: int = 8, decode: bool = True,
                       uniforms: Optional[Callable[[int], np.ndarray]] = None) -> BatchResult:
        """prompts: one (tokens, mask) pair per stream (`prompt_frames`), lengths may differ.  `uniforms(i)` (optional) supplies frame i's
        [B, n_cb] sampling uniforms instead of the seeded generator.  Frames are generated for all streams until every
        stream has emitted its EOS frame; the host looks at the EOS flags only every `eos_check_interval` frames (one sync per interval instead
        of one per frame), so a few frames past the last EOS may be generated and are dropped.  Stream b's audio holds exactly its own frames."""
        start = time.perf_counter()
        B = len(prompts)
        max_audio_frames = int(max_audio_length_ms / 80)
        frames, first_eos = [], torch.full((B,), -1, dtype=torch.int64, device=self.model.device)
        for i, (sample, was_done, done) in enumerate(self._frame_loop(prompts, max_audio_frames, temperature, top_k, seed, stop_on_eos, uniforms)):
            frames.append(sample)
            newly = done & ~was_done
            first_eos = torch.where(newly, torch.full_like(first_eos, i), first_eos)
            if stop_on_eos and (i + 1) % max(1, eos_check_interval) == 0 and bool(done.all()):
                break
        fe = first_eos.cpu().tolist()
        counts = [fe[b] if fe[b] >= 0 else len(frames) for b in range(B)]
        if max(counts) == 0:
            raise AssertionError("No audio generated")
        T = max(counts)
        codes = torch.stack(frames[:T], dim=2)  # [B, K, T]  (mx.transpose(mx.stack(samples), [1, 2, 0]), sesame.py:623)
        audio = []
        if decode:
            if self._audio_tokenizer is None:
                raise ValueError("decoding needs the Mimi codec: pass mimi= or config['mimi_path']")
            pcm = self._audio_tokenizer.decode(codes)[:, 0]
            spf = pcm.shape[1] // T
            audio = [pcm[b, : counts[b] * spf] for b in range(B)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - start
        secs = T * 0.08
        return BatchResult(audio=audio, frames=counts, sample_rate=self.sample_rate, processing_time_seconds=dt,
                           real_time_factor=dt / secs if secs > 0 else 0.0, codes=codes)

    # ---- results (sesame.py:619-680) ------------------------------------------------------------------------------------------------------------
    def _result(self, audio: torch.Tensor, token_count: int, start_time: float) -> GenerationResult:
        torch.cuda.synchronize()
        seg_t = time.perf_counter() - start_time
        samples = int(audio.shape[0])
        assert samples > 0, "No audio generated"
        dur_s = samples / self.sample_rate
        h, m_, s_, ms = int(dur_s // 3600), int(dur_s // 60), int(dur_s % 60), int((dur_s % 1) * 1000)
        return GenerationResult(
            audio=audio, samples=samples, sample_rate=self.sample_rate, segment_idx=0, token_count=token_count,
            audio_duration=f"{h:02d}:{m_:02d}:{s_:02d}.{ms:03d}", real_time_factor=round(seg_t / dur_s, 2) if dur_s > 0 else 0,
            prompt={"tokens": token_count, "tokens-per-sec": round(token_count / seg_t, 2) if seg_t > 0 else 0},
            audio_samples={"samples": samples, "samples-per-sec": round(samples / seg_t, 2) if seg_t > 0 else 0},
            processing_time_seconds=seg_t, peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9)

    def prepare_prompt(self, text: TextLike, speaker: int, audio_path: str, sample_rate: int = 24000) -> Segment:
        audio, sr = _read_wav(audio_path)
        if sr != sample_rate:
            raise ValueError(f"{audio_path}: {sr} Hz; resample to {sample_rate} Hz first (the reference uses scipy's resample here)")
        return Segment(text=text, speaker=speaker, audio=audio)

    def default_speaker_prompt(self, voice: str, repo_id=None) -> List[Segment]:
        """sesame.py:583-617 downloads prompts/{voice}.wav|.txt; offline `voice` must be a path to a local .wav with a .txt beside it."""
        import os

        wav = voice if voice.endswith(".wav") else voice + ".wav"
        txt = os.path.splitext(wav)[0] + ".txt"
        if not (os.path.exists(wav) and os.path.exists(txt)):
            raise FileNotFoundError(f"speaker prompt {voice!r}: the reference downloads it from the hub; offline, pass a local .wav with a .txt "
                                    "beside it, or ref_audio / ref_text, or context=[Segment(...)]")
        return [self.prepare_prompt(open(txt).read(), 0, wav)]

    # ---- Model.generate (sesame.py:689-817) -----------------------------------------------------------------------------------------------------------
    def generate(self, text: Union[TextLike, List[TextLike]], voice: Optional[str] = None, speaker: int = 0, context: Optional[List[Segment]] = None,
                 split_pattern: Optional[str] = r"\n+", sampler: Callable = None, max_audio_length_ms: float = 90_000, ref_audio=None,
                 ref_text: Optional[TextLike] = None, stream: bool = False, streaming_interval: float = 0.5, voice_match: bool = True,
                 seed: Optional[int] = None, stop_on_eos: bool = True, **kwargs):
        """Yields one GenerationResult per text prompt (per `streaming_interval` seconds of frames with stream=True).  `sampler` is what
        `make_sampler(temp, top_k)` of this module returns (the reference takes mlx_lm's callable of the same name, sesame.py:719) and
        defaults, as there, to temp 0.9 / top_k 50.  A bare `temperature=` -- generate_audio forwards its own default 0.7 to every model
        (generate.py:288-300) -- lands in **kwargs and is IGNORED, exactly as the reference's signature ignores it."""
        sampler = sampler or make_sampler(temp=0.9, top_k=50)
        temperature, top_k = float(sampler.temp), int(sampler.top_k)
        context = list(context or [])
        if not context and ref_audio is not None and ref_text is not None:
            a = ref_audio.detach().cpu().numpy() if isinstance(ref_audio, torch.Tensor) else np.asarray(ref_audio, np.float32)
            context = [Segment(speaker=speaker, text=ref_text, audio=a)]
        elif ref_audio is None and not context:
            context = self.default_speaker_prompt(voice if voice is not None else "conversational_a")
        max_audio_frames = int(max_audio_length_ms / 80)
        interval = max(1, int(streaming_interval * 12.5))
        if isinstance(text, str):
            text = re.split(split_pattern, text.strip()) if split_pattern else [text]
        elif text and isinstance(text[0], (int, np.integer)):
            text = [text]  # one pre-tokenised prompt
        for prompt in text:
            start = time.perf_counter()
            frames = self.prompt_frames(context, prompt, speaker, voice_match=voice_match)
            if stream:
                if self._streaming_decoder is None:
                    self._streaming_decoder = MimiStreamingDecoder(self._audio_tokenizer)
                self._streaming_decoder.reset()
            samples = []
            self._seed_counter += 1
            for sample, was_done, done in self._frame_loop([frames], max_audio_frames, temperature, top_k,
                                                           seed if seed is None else seed + self._seed_counter - 1, stop_on_eos):
                if bool(done[0]):
                    break  # eos (batch 1: one sync per frame, as in the reference)
                samples.append(sample)
                if stream and len(samples) >= interval:
                    audio = self._streaming_decoder.decode_frames(torch.stack(samples, dim=2))[0, 0]
                    yield self._result(audio, len(samples), start)
                    samples, start = [], time.perf_counter()
            if samples:
                codes = torch.stack(samples, dim=2)
                audio = (self._streaming_decoder.decode_frames(codes) if stream else self._audio_tokenizer.decode(codes))[0, 0]
                yield self._result(audio, len(samples), start)

    def generate_stream(self, contexts, prompts_ids, **kw):  # (round-1 name: batch streaming; kept for the streaming test)
        from .mimi import MimiStreamingDecoder as _D

        interval = max(1, int(kw.pop("streaming_interval", 2.0) * 12.5))
        max_audio_frames = int(kw.pop("max_audio_length_ms", 90_000) / 80)
        prompts = [self.prompt_frames(c, p, 0, voice_match=False) for c, p in zip(contexts, prompts_ids)]
        dec = _D(self._audio_tokenizer)
        samples, start = [], time.perf_counter()
        for sample, was_done, done in self._frame_loop(prompts, max_audio_frames, kw.get("temperature", 0.9), kw.get("top_k", 50), kw.get("seed", 0),
                                                       kw.get("stop_on_eos", True)):
            if kw.get("stop_on_eos", True) and bool(done.all()):
                break
            samples.append(sample)
            if len(samples) >= interval:
                yield self._batch_part(dec, samples, start)
                samples, start = [], time.perf_counter()
        if samples:
            yield self._batch_part(dec, samples, start)

    def _batch_part(self, dec, samples, start):
        audio = dec.decode_frames(torch.stack(samples, dim=2))[:, 0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - start
        secs = audio.shape[1] / self.sample_rate
        return BatchResult(audio=list(audio), frames=[len(samples)] * audio.shape[0], sample_rate=self.sample_rate, processing_time_seconds=dt,
                           real_time_factor=dt / secs if secs > 0 else 0.0)


# ---- helpers ----------------------------------------------------------------------------------------------------------------------------------------------
def llama3_2_1B() -> dict:  # sesame.py:225-248
    return dict(num_layers=16, num_heads=32, num_kv_heads=8, head_dim=64, hidden=2048, intermediate=8192, rope_theta=500000.0, rope_factor=32.0, rms_eps=1e-5)


def llama3_2_100M() -> dict:  # sesame.py:250-273
    return dict(num_layers=4, num_heads=8, num_kv_heads=2, head_dim=128, hidden=1024, intermediate=8192, rope_theta=500000.0, rope_factor=32.0, rms_eps=1e-5)


FLAVORS = {"llama-1B": llama3_2_1B, "llama-100M": llama3_2_100M}


def csm_config_from(config: dict) -> dict:
    """The checkpoint's config.json (sesame.py:276-318: backbone_flavor / decoder_flavor / vocab sizes) -> the engine's explicit form; a
    dict that already holds `backbone` / `decoder` size dicts passes through."""
    if "backbone" in config and "decoder" in config:
        return config
    return dict(text_vocab_size=int(config["text_vocab_size"]), audio_vocab_size=int(config["audio_vocab_size"]),
                audio_num_codebooks=int(config["audio_num_codebooks"]), max_seq_len=int(config.get("max_seq_len", 2048)),
                backbone=FLAVORS[config["backbone_flavor"]](), decoder=FLAVORS[config["decoder_flavor"]]())


def _load_llama3_tokenizer(path: str):
    """sesame.py:427-438 with `local_files_only=True`: bos / eos added around the text by a TemplateProcessing post-processor."""
    from tokenizers.processors import TemplateProcessing
    from transformers import AutoTokenizer

    tok = AutoTokenizer.from_pretrained(path, local_files_only=True)
    bos, eos = tok.bos_token, tok.eos_token
    tok._tokenizer.post_processor = TemplateProcessing(single=f"{bos}:0 $A:0 {eos}:0", pair=f"{bos}:0 $A:0 {eos}:0 {bos}:1 $B:1 {eos}:1",
                                                       special_tokens=[(f"{bos}", tok.bos_token_id), (f"{eos}", tok.eos_token_id)])
    return tok


def _load_mimi(path: str, nq: int) -> Mimi:
    import glob
    import os

    from safetensors.numpy import load_file

    from .mimi import mimi_202407

    files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
    if not files:
        raise FileNotFoundError(f"No safetensors found in {path} (Mimi codec)")
    w = {}
    for f in files:
        w.update(load_file(f))
    cfg = mimi_202407(nq)  # what the reference builds (sesame.py:456, mimi.py:252-262); a config.json beside the weights overrides it
    cj = os.path.join(path, "config.json")
    if os.path.exists(cj):
        import json

        from .mimi import MimiConfig

        cfg = MimiConfig.from_dict(json.load(open(cj)))
    return Mimi(cfg, w)


def _read_wav(path: str):
    import wave

    with wave.open(path, "rb") as w:
        sr, n, ch, sw = w.getframerate(), w.getnframes(), w.getnchannels(), w.getsampwidth()
        raw = w.readframes(n)
    if sw != 2:
        raise ValueError(f"{path}: only 16-bit PCM wav is read without soundfile")
    a = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    return (a.reshape(-1, ch).mean(axis=1) if ch > 1 else a), sr
