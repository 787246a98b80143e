"""MI355X-native Kokoro-82M acoustic path (HIP kernels behind a C ABI).

Mirrors the reference's Python surface for the hot path only:
`load_model` (mlx_audio/tts/utils.py:150), `Model.__call__`/`Model.generate`
(mlx_audio/tts/models/kokoro/kokoro.py:120,269), `KokoroPipeline`
(mlx_audio/tts/models/kokoro/pipeline.py:66) and `generate_audio`
(mlx_audio/tts/generate.py:203).  Sub-modules are imported lazily so that
`import mlx_audio_amd.params` works without the HIP extension.
"""

__version__ = "0.1.0"
