"""Host-side mirror of mlx_audio/tts/generate.py:203-358 (`generate_audio`) for the Kokoro path."""
from __future__ import annotations

import os
import wave
from typing import Optional

import numpy as np


def _write_wav(path: str, audio: np.ndarray, sample_rate: int) -> None:
    pcm = np.clip(np.asarray(audio, np.float32), -1.0, 1.0)
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes((pcm * 32767.0).astype("<i2").tobytes())


def generate_audio(text: str, model_path: str = "prince-canuma/Kokoro-82M", voice: str = "af_heart", speed: float = 1.0,
                   lang_code: str = "a", file_prefix: str = "audio", audio_format: str = "wav", join_audio: bool = False,
                   verbose: bool = True, model=None, **kwargs) -> None:
    """Writes `{file_prefix}_{i:03d}.wav` per segment (or one joined file).  Like the reference it catches every
    exception and prints it (generate.py:349-358) instead of raising."""
    try:
        from .utils import load_model

        if audio_format != "wav":
            raise ValueError("only wav output is available (soundfile is not a dependency of this engine)")
        model = model or load_model(model_path)
        chunks = []
        for i, r in enumerate(model.generate(text=text, voice=voice, speed=speed, lang_code=lang_code, **kwargs)):
            a = r.audio.detach().float().cpu().numpy()
            if join_audio:
                chunks.append(a)
            else:
                _write_wav(f"{file_prefix}_{i:03d}.wav", a, r.sample_rate)
            if verbose:
                print(f"segment {i}: {r.audio_duration} audio, RTF {r.real_time_factor}, {r.audio_samples['samples-per-sec']} samples/s, "
                      f"peak {r.peak_memory_usage:.2f} GB")
        if join_audio and chunks:
            _write_wav(f"{file_prefix}.wav", np.concatenate(chunks), model.sample_rate)
    except Exception as e:  # noqa: BLE001  (reference behaviour)
        print(f"Error generating audio: {e}")
        import traceback

        traceback.print_exc()
