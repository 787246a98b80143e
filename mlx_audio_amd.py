"""Import alias: the package directory is `mlx-audio_amd/` (not a valid Python identifier).

`import mlx_audio_amd` loads that directory as the package `mlx_audio_amd`.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_d = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mlx-audio_amd")
_spec = _ilu.spec_from_file_location("mlx_audio_amd", _os.path.join(_d, "__init__.py"), submodule_search_locations=[_d])
_m = _ilu.module_from_spec(_spec)
_sys.modules["mlx_audio_amd"] = _m
_spec.loader.exec_module(_m)
