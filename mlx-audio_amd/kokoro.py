"""Host-side mirror of mlx_audio/tts/models/kokoro/kokoro.py: `ModelConfig`, `Model.__call__`, `Model.generate`.

Same names, argument meaning and error behaviour as the reference; the arithmetic runs in libkokoro_hip.so
through `KokoroEngine`.  Differences a caller can observe:
  * `audio` is a float32 torch tensor on the GPU (`[1, 600*F]`), `pred_dur` an int32 torch tensor `[T]`;
  * `Model.batch_call` runs many utterances in one padded batch (the reference is batch-1 only,
    kokoro.py:135-136) with results bit-identical to per-utterance calls;
  * the noise of the harmonic source (istftnet.py:620) comes from a Philox stream (`seed`), or can be injected.
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from numbers import Number
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .base import BaseModelArgs, GenerationResult
from .engine import KokoroEngine


@dataclass
class ModelConfig(BaseModelArgs):
    """kokoro.py:47-63."""
    istftnet: dict
    dim_in: int
    dropout: float
    hidden_dim: int
    max_conv_dim: int
    max_dur: int
    multispeaker: bool
    n_layer: int
    n_mels: int
    n_token: int
    style_dim: int
    text_encoder_kernel_size: int
    plbert: dict
    vocab: Dict[str, int]
    sample_rate: int = 24000


class Model:
    """kokoro.py:66-170.  Language-blind: maps phonemes -> ids with `config.vocab` and runs the acoustic forward."""

    REPO_ID = "prince-canuma/Kokoro-82M"

    @dataclass
    class Output:
        audio: torch.Tensor
        pred_dur: Optional[torch.Tensor] = None

    def __init__(self, config: ModelConfig, repo_id: str = None, weights: Optional[dict] = None, compute_dtype: str = "float32",
                 quantization: Optional[dict] = None):
        self.repo_id = repo_id
        self.config = config
        self.vocab = config.vocab
        self.context_length = int(config.plbert["max_position_embeddings"])  # kokoro.py:93
        self._pipelines: Dict[str, "KokoroPipeline"] = {}
        self._engine: Optional[KokoroEngine] = None
        self._compute_dtype = compute_dtype
        self._quantization = quantization  # config["quantization"] of an MLX 8-bit checkpoint (tts/utils.py:241-260)
        self._seed = 0
        if weights is not None:
            self.load_weights(weights)

    # -- weights -----------------------------------------------------------------------------------
    def _cfg_dict(self) -> dict:
        c = self.config
        return dict(istftnet=c.istftnet, hidden_dim=c.hidden_dim, max_dur=c.max_dur, n_layer=c.n_layer, n_token=c.n_token,
                    style_dim=c.style_dim, text_encoder_kernel_size=c.text_encoder_kernel_size, plbert=c.plbert)

    def load_weights(self, weights, strict: bool = True):
        """Accepts MLX-side or PyTorch-side names/layouts (what `sanitize`, kokoro.py:172-252, converts between)."""
        items = dict(weights.items() if hasattr(weights, "items") else weights)
        cfg = self._cfg_dict()
        g = items.get("decoder.encode.conv1.weight_g")  # width of the decoder blocks (1024 in Kokoro-82M, istftnet.py:917)
        if g is not None:
            cfg["decoder_hidden"] = int(g.shape[0])
        self._engine = KokoroEngine(cfg, items, compute_dtype=self._compute_dtype, quantization=self._quantization)
        return self

    def sanitize(self, weights):
        """Layout normalisation happens inside kk_load_tensor (by expected shape); nothing to do on the host."""
        return weights

    def new_context(self) -> "Model":
        """A second Model on the SAME device weights (one kk_model, a new kk_context: own graph cache, side stream, workspace) for another
        stream / thread in flight (TTSService(contexts=N), bench.py --streams).  The reference's Model is single threaded (kokoro.py:83-113)."""
        import copy

        other = copy.copy(self)
        other._pipelines = {}
        other._engine = self.engine.new_context()
        return other

    @property
    def engine(self) -> KokoroEngine:
        if self._engine is None:
            raise _lib.KokoroHipError("Model has no weights: call load_weights() or use load_model()")
        return self._engine

    @property
    def sample_rate(self):
        return self.config.sample_rate

    # -- forward -----------------------------------------------------------------------------------
    def _ids(self, phonemes: str) -> List[int]:
        ids = [self.vocab[p] for p in phonemes if p in self.vocab]  # unknown symbols are dropped (kokoro.py:128-130)
        assert len(ids) + 2 <= self.context_length, (len(ids) + 2, self.context_length)  # kokoro.py:131-134
        return ids

    def batch_call(self, phonemes: Sequence[str], ref_s, speed: Union[Number, Sequence[Number]] = 1, seed: Optional[int] = None,
                   noise_mode: int = _lib.NOISE_PHILOX):
        """B utterances in one padded batch.  ref_s [B, 256]; speed one number or one per utterance.  Returns a list of Output
        (audio [1, 600*F_b]).  The text stage runs ONCE: kk_forward_text, the reference's host sync on the durations (kokoro.py:151-153),
        then kk_forward_audio on the text stage's results in the workspace."""
        eng = self.engine
        dev = eng.device
        B = len(phonemes)
        ids, lens, Tmax = eng.pack_ids([self._ids(p) for p in phonemes])
        ref = torch.as_tensor(np.asarray(ref_s.detach().cpu() if isinstance(ref_s, torch.Tensor) else ref_s, dtype=np.float32)).reshape(B, 256).to(dev)
        sp = torch.full((B,), float(speed), device=dev) if isinstance(speed, Number) else torch.tensor(list(speed), dtype=torch.float32, device=dev)
        if seed is None:
            self._seed += 1
            seed = self._seed
        eng.workspace(B, Tmax, 0)
        pred = eng.forward_text(ids, lens, ref, sp)
        F = pred.sum(dim=1).cpu()  # the host sync: durations decide the output length
        Fmax = int(F.max())
        assert 0 < Fmax <= max(1, int(self.config.max_dur / float(sp.min().item()) + 1)) * Tmax
        wav, nfr = eng.forward_audio(B, Tmax, lens, ref, pred, Fmax, noise_mode=noise_mode, seed=seed)
        Ts = lens.cpu().tolist()
        return [self.Output(audio=wav[b : b + 1, : 600 * int(F[b])], pred_dur=pred[b, : Ts[b]]) for b in range(B)]

    def __call__(self, phonemes: str, ref_s, speed: Number = 1, return_output: bool = False, decoder=None):
        o = self.batch_call([phonemes], ref_s, speed)[0]
        return o if return_output else o.audio

    # -- generate ----------------------------------------------------------------------------------
    def _get_pipeline(self, lang_code: str):
        from .pipeline import KokoroPipeline

        if lang_code not in self._pipelines:
            self._pipelines[lang_code] = KokoroPipeline(model=self, repo_id=self.REPO_ID if self.repo_id is None else self.repo_id,
                                                        lang_code=lang_code)
        return self._pipelines[lang_code]

    def generate(self, text: str, voice: str = None, speed: float = 1.0, lang_code: str = "a", split_pattern: str = r"\n+", **kwargs):
        """kokoro.py:269-346: yields one GenerationResult per text segment."""
        pipeline = self._get_pipeline(lang_code)
        if voice is None:
            voice = "af_heart"
        start = time.time()
        # batch_size (an addition): chunks of one request synthesised as padded batches (KokoroPipeline.plan_batches); 1 = the reference's chunk by chunk
        for segment_idx, (graphemes, phonemes, audio) in enumerate(pipeline(text, voice=voice, speed=speed, split_pattern=split_pattern,
                                                                            batch_size=int(kwargs.get("batch_size", 1)))):
            torch.cuda.synchronize()
            now = time.time()
            seg_t, start = now - start, now
            samples = audio.shape[-1] if audio is not None else 0
            assert samples > 0, "No audio generated"
            token_count = len(phonemes) if phonemes is not None else 0
            dur_s = samples / self.config.sample_rate
            rtf = seg_t / dur_s if dur_s > 0 else 0
            h, m, s, ms = int(dur_s // 3600), int(dur_s // 60), int(dur_s % 60), int((dur_s % 1) * 1000)
            yield GenerationResult(
                audio=audio[0], samples=samples, sample_rate=self.config.sample_rate, segment_idx=segment_idx, token_count=token_count,
                audio_duration=f"{h:02d}:{m:02d}:{s:02d}.{ms:03d}", real_time_factor=round(rtf, 2),
                prompt={"tokens": token_count, "tokens-per-sec": round(token_count / seg_t, 2) if seg_t > 0 else 0},
                audio_samples={"samples": samples, "samples-per-sec": round(samples / seg_t, 2) if seg_t > 0 else 0},
                processing_time_seconds=seg_t, peak_memory_usage=torch.cuda.max_memory_allocated() / 1e9,
            )
