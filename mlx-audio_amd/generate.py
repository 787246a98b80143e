"""Host-side mirror of mlx_audio/tts/generate.py (`load_audio` :17-50, `generate_audio` :203-358): same arguments, same files written, same
printed statistics and the same catch-and-print error behaviour.  What this engine does not carry is said where it applies: audio playback
(`play`; there is no sound device beside an MI355X), Whisper transcription of a reference clip (`stt_model`; `ref_text` must be given), and
file formats other than 16-bit wav unless `soundfile` happens to be importable."""
from __future__ import annotations

import inspect
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


def _write_audio(path: str, audio: np.ndarray, sample_rate: int, audio_format: str) -> None:
    if audio_format == "wav":
        return _write_wav(path, audio, sample_rate)
    try:
        import soundfile as sf  # the reference's writer (generate.py:311,337)
    except ImportError as e:
        raise ValueError(f"audio_format={audio_format!r} needs the soundfile package (the reference's writer); without it only wav is written") from e
    sf.write(path, np.asarray(audio, np.float32), sample_rate)


def audio_volume_normalize(audio: np.ndarray, coeff: float = 0.2) -> np.ndarray:
    """generate.py:53-103 (used for the Spark model's reference clips): scale so that the mean of the loudest 90-99 % of the magnitudes
    above 0.1 sits at `coeff`, the scale kept within [0.1, 10], the peak within 1."""
    audio = np.asarray(audio, np.float32).copy()
    temp = np.sort(np.abs(audio))
    if temp[-1] < 0.1:
        audio = audio * (0.1 / max(float(temp[-1]), 1e-3))
    temp = temp[temp > 0.01]
    if len(temp) <= 10:
        return audio
    volume = float(np.mean(temp[int(0.9 * len(temp)) : int(0.99 * len(temp))]))
    audio = audio * float(np.clip(coeff / volume, 0.1, 10))
    peak = float(np.max(np.abs(audio)))
    return audio / peak if peak > 1 else audio


def load_audio(audio_path: str, sample_rate: int = 24000, length: Optional[int] = None, volume_normalize: bool = False,
               segment_duration: Optional[int] = None) -> np.ndarray:
    """generate.py:17-50: mono float32 at `sample_rate` (channels averaged, Fourier resampling as scipy.signal.resample does it there)."""
    try:
        import soundfile as sf

        samples, sr = sf.read(audio_path)
        samples = np.asarray(samples, np.float32)
        if samples.ndim > 1:
            samples = samples.sum(axis=1) / samples.shape[1]
    except ImportError:
        with wave.open(audio_path, "rb") as w:
            sr, n, ch, sw = w.getframerate(), w.getnframes(), w.getnchannels(), w.getsampwidth()
            raw = w.readframes(n)
        if sw != 2:
            raise ValueError(f"{audio_path}: only 16-bit PCM wav is read without the soundfile package")
        samples = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
        if ch > 1:
            samples = samples.reshape(-1, ch).sum(axis=1) / ch
    if sr != sample_rate:
        from scipy.signal import resample

        print(f"Resampling from {sr} to {sample_rate}")
        samples = resample(samples, int(samples.shape[0] / sr * sample_rate)).astype(np.float32)
    if segment_duration is not None:  # generate.py:106-127: a random window of that many seconds (padded when the clip is shorter)
        seg = int(sample_rate * segment_duration)
        if samples.shape[0] < seg:
            samples = np.pad(samples, (0, seg - samples.shape[0]))
        start = np.random.randint(0, samples.shape[0] - seg + 1)
        samples = samples[start : start + seg]
    if volume_normalize:
        samples = audio_volume_normalize(samples)
    if length is not None:
        assert abs(samples.shape[0] - length) < 1000
        samples = samples[:length] if samples.shape[0] > length else np.pad(samples, (0, int(length - samples.shape[0])))
    return samples


def _to_numpy(audio) -> np.ndarray:
    if hasattr(audio, "detach"):
        audio = audio.detach().float().cpu().numpy()
    return np.asarray(audio, np.float32).reshape(-1)


def generate_audio(text: str, model_path: str = "prince-canuma/Kokoro-82M", max_tokens: int = 1200, voice: str = "af_heart", speed: float = 1.0,
                   lang_code: str = "a", ref_audio: Optional[str] = None, ref_text: Optional[str] = None,
                   stt_model: str = "mlx-community/whisper-large-v3-turbo", file_prefix: str = "audio", audio_format: str = "wav",
                   join_audio: bool = False, play: bool = False, verbose: bool = True, temperature: float = 0.7, stream: bool = False,
                   streaming_interval: float = 2.0, model=None, **kwargs) -> None:
    """generate.py:203-358.  Writes `{file_prefix}_{i:03d}.{fmt}` per result, or one `{file_prefix}.{fmt}` with join_audio; with
    stream=True the results are partial segments every `streaming_interval` seconds and (as there) only a joined file is written.
    `model` (an addition) takes an already loaded model instead of `model_path`."""
    try:
        from .utils import load_model

        if play or stream:
            # the reference plays while it generates (AudioPlayer, generate.py:283,305-306); this host has no sound device
            print("play/stream: no audio device on this host; results are only written to files")
        model = model or load_model(model_path)
        accepts = inspect.signature(model.generate).parameters
        if ref_audio:
            if not os.path.exists(ref_audio):
                raise FileNotFoundError(f"Reference audio file not found: {ref_audio}")
            normalize = hasattr(model, "model_type") and callable(model.model_type) and model.model_type() == "spark"
            ref_audio = load_audio(ref_audio, sample_rate=model.sample_rate, volume_normalize=normalize)
            if not ref_text and "ref_text" in accepts:
                # generate.py:268-280 transcribes with Whisper (`stt_model`) here: speech-to-text is not part of this engine
                raise ValueError(f"ref_text is required with ref_audio (the reference would transcribe it with {stt_model}; no STT model here)")
        print(f"\n\033[94mModel:\033[0m {model_path}\n\033[94mText:\033[0m {text}\n\033[94mVoice:\033[0m {voice}\n"
              f"\033[94mSpeed:\033[0m {speed}x\n\033[94mLanguage:\033[0m {lang_code}")
        results = model.generate(text=text, voice=voice, speed=speed, lang_code=lang_code, ref_audio=ref_audio, ref_text=ref_text,
                                 temperature=temperature, max_tokens=max_tokens, verbose=verbose, stream=stream,
                                 streaming_interval=streaming_interval, **kwargs)
        audio_list = []
        file_name = f"{file_prefix}.{audio_format}"
        for i, result in enumerate(results):
            a = _to_numpy(result.audio)
            if join_audio:
                audio_list.append(a)
            elif not stream:
                file_name = f"{file_prefix}_{i:03d}.{audio_format}"
                _write_audio(file_name, a, result.sample_rate, audio_format)
                print(f"✅ Audio successfully generated and saving as: {file_name}")
            if verbose:
                print("==========")
                print(f"Duration:              {result.audio_duration}")
                print(f"Samples/sec:           {result.audio_samples['samples-per-sec']:.1f}")
                print(f"Prompt:                {result.token_count} tokens, {result.prompt['tokens-per-sec']:.1f} tokens-per-sec")
                print(f"Audio:                 {result.audio_samples['samples']} samples, {result.audio_samples['samples-per-sec']:.1f} samples-per-sec")
                print(f"Real-time factor:      {result.real_time_factor:.2f}x")
                print(f"Processing time:       {result.processing_time_seconds:.2f}s")
                print(f"Peak memory usage:     {result.peak_memory_usage:.2f}GB")
        if join_audio and not stream:
            if verbose:
                print(f"Joining {len(audio_list)} audio files")
            if audio_list:
                _write_audio(f"{file_prefix}.{audio_format}", np.concatenate(audio_list, axis=0), model.sample_rate, audio_format)
            if verbose:
                print(f"✅ Audio successfully generated and saving as: {file_name}")
    except ImportError as e:
        print(f"Import error: {e}")
        print("This might be due to incorrect Python path. Check your project structure.")
    except Exception as e:  # noqa: BLE001  (reference behaviour, generate.py:353-357)
        print(f"Error loading model: {e}")
        import traceback

        traceback.print_exc()
