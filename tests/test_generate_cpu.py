"""generate_audio / load_audio (mlx_audio/tts/generate.py:17-127,203-358) on the host side: arguments, files written, error behaviour.
A stand-in model (no GPU) records what `generate` was called with."""
import os
import wave

import numpy as np
import pytest

from mlx_audio_amd import generate as G
from mlx_audio_amd.base import GenerationResult


class _Model:
    sample_rate = 24000

    def __init__(self):
        self.calls = []

    def generate(self, text, voice=None, speed=1.0, lang_code="a", ref_audio=None, ref_text=None, stream=False, streaming_interval=2.0, **kwargs):
        self.calls.append(dict(text=text, voice=voice, speed=speed, lang_code=lang_code, ref_audio=ref_audio, ref_text=ref_text, stream=stream,
                               streaming_interval=streaming_interval, **kwargs))
        for i in range(3):
            a = np.full(2400, 0.1 * (i + 1), np.float32)
            yield GenerationResult(audio=a, samples=a.size, sample_rate=24000, segment_idx=i, token_count=7, audio_duration="00:00:00.100",
                                   real_time_factor=0.01, prompt={"tokens": 7, "tokens-per-sec": 70.0},
                                   audio_samples={"samples": a.size, "samples-per-sec": 24000.0}, processing_time_seconds=0.1, peak_memory_usage=0.5)


def _read(path):
    with wave.open(path, "rb") as w:
        return np.frombuffer(w.readframes(w.getnframes()), "<i2").astype(np.float32) / 32767.0, w.getframerate()


def test_generate_audio_files_join_stream_and_forwarded_arguments(tmp_path, capsys):
    m = _Model()
    pre = str(tmp_path / "out")
    G.generate_audio("hello", model=m, voice="af_heart", speed=1.2, max_tokens=55, temperature=0.3, file_prefix=pre, verbose=True, top_k=5)
    for i in range(3):
        a, sr = _read(f"{pre}_{i:03d}.wav")
        assert sr == 24000 and a.size == 2400 and abs(a[0] - 0.1 * (i + 1)) < 1e-3
    c = m.calls[-1]
    assert (c["speed"], c["max_tokens"], c["temperature"], c["top_k"], c["stream"], c["streaming_interval"]) == (1.2, 55, 0.3, 5, False, 2.0)
    out = capsys.readouterr().out
    assert "Real-time factor:" in out and "Peak memory usage:" in out and "Prompt:                7 tokens" in out
    # one joined file
    pre2 = str(tmp_path / "joined")
    G.generate_audio("hello", model=m, file_prefix=pre2, join_audio=True, verbose=False)
    a, _ = _read(pre2 + ".wav")
    assert a.size == 7200 and not os.path.exists(pre2 + "_000.wav")
    # stream: partial results are not written one by one (generate.py:309: `elif not stream`)
    pre3 = str(tmp_path / "streamed")
    G.generate_audio("hello", model=m, file_prefix=pre3, stream=True, streaming_interval=0.5, verbose=False)
    assert m.calls[-1]["stream"] is True and m.calls[-1]["streaming_interval"] == 0.5
    assert not os.path.exists(pre3 + "_000.wav") and not os.path.exists(pre3 + ".wav")


def test_generate_audio_reference_clip_and_errors_are_printed_not_raised(tmp_path, capsys):
    m = _Model()
    # a stereo 16 kHz clip -> mono 24 kHz
    t = np.arange(16000) / 16000.0
    st = np.stack([np.sin(2 * np.pi * 220 * t), np.zeros_like(t)], 1)
    clip = str(tmp_path / "ref.wav")
    with wave.open(clip, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes((st * 32767 * 0.5).astype("<i2").tobytes())
    a = G.load_audio(clip, sample_rate=24000)
    assert a.shape == (24000,) and a.dtype == np.float32
    ref = 0.25 * np.sin(2 * np.pi * 220 * np.arange(24000) / 24000.0)  # mean of the two channels, Fourier-resampled
    assert np.abs(a[100:-100] - ref[100:-100]).max() < 5e-3
    assert G.load_audio(clip, sample_rate=24000, segment_duration=2).shape == (48000,)
    assert G.load_audio(clip, sample_rate=16000, length=16400).shape == (16400,)
    G.generate_audio("hi", model=m, ref_audio=clip, ref_text="a tone", file_prefix=str(tmp_path / "r"), verbose=False)
    c = m.calls[-1]
    assert c["ref_text"] == "a tone" and isinstance(c["ref_audio"], np.ndarray) and c["ref_audio"].shape == (24000,)
    # no caption: the reference transcribes with Whisper; here it is an error, printed like every other one (generate.py:353-357)
    n = len(m.calls)
    G.generate_audio("hi", model=m, ref_audio=clip, file_prefix=str(tmp_path / "r2"), verbose=False)
    assert len(m.calls) == n and "ref_text is required" in capsys.readouterr().out
    G.generate_audio("hi", model=m, ref_audio=str(tmp_path / "missing.wav"), ref_text="x", verbose=False)
    assert "Reference audio file not found" in capsys.readouterr().out
    try:
        import soundfile  # noqa: F401
    except ImportError:
        G.generate_audio("hi", model=m, audio_format="flac", file_prefix=str(tmp_path / "f"), verbose=False)
        assert "needs the soundfile package" in capsys.readouterr().out


def test_audio_volume_normalize_restates_generate_py_53_103():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(5000) * 0.02).astype(np.float32)  # quiet: first lifted to a 0.1 peak, then to the 0.2 level
    y = G.audio_volume_normalize(x)
    temp = np.sort(np.abs(x))
    x2 = x / max(temp[-1], 1e-3) * 0.1
    t2 = temp[temp > 0.01]
    vol = np.mean(t2[int(0.9 * len(t2)) : int(0.99 * len(t2))])
    want = x2 * np.clip(0.2 / vol, 0.1, 10)
    want = want / np.abs(want).max() if np.abs(want).max() > 1 else want
    np.testing.assert_allclose(y, want, rtol=1e-6)
    assert np.array_equal(G.audio_volume_normalize(np.zeros(100, np.float32) + 0.5)[:3], (np.zeros(3) + 0.5 * np.clip(0.2 / 0.5, 0.1, 10)).astype(np.float32))
