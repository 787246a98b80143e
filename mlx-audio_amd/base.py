"""Small shared types of the host surface (mirror of mlx_audio/tts/models/base.py)."""
from __future__ import annotations

import inspect
from dataclasses import dataclass
from typing import Any


@dataclass
class BaseModelArgs:
    @classmethod
    def from_dict(cls, params: dict):
        """Keep only the keys the constructor accepts (base.py:8-18)."""
        accepted = inspect.signature(cls).parameters
        return cls(**{k: v for k, v in params.items() if k in accepted})


def check_array_shape(arr) -> bool:
    """The reference's conv-weight layout heuristic (base.py:21-34): True for a 3-D [O, K, K'] array with
    O >= K == K'.  Kept for interface parity; the loader itself decides layouts by EXPECTED shape
    (kk_load_tensor), because this heuristic is ambiguous whenever K == C_in."""
    shape = arr.shape
    if len(shape) != 3:
        return False
    out_channels, kh, kw = shape
    return bool(out_channels >= kh and out_channels >= kw and kh == kw)


@dataclass
class GenerationResult:
    """Same fields as base.py:71-84."""
    audio: Any
    samples: int
    sample_rate: int
    segment_idx: int
    token_count: int
    audio_duration: str
    real_time_factor: float
    prompt: dict
    audio_samples: dict
    processing_time_seconds: float
    peak_memory_usage: float
