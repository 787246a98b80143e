"""Host-side mirror of mlx_audio/tts/utils.py:150-268 (`load_model`) for the Kokoro path.

A local directory with `config.json` + `*.safetensors` is loaded as is; both checkpoint layouts (PyTorch-side
names / [O, I, K] convs, or MLX-side / [O, K, I]) are accepted.  There is no network here, so repo ids are only
resolved through the local Hugging Face cache (`snapshot_download(local_files_only=True)`)."""
from __future__ import annotations

import glob
import json
import logging
from pathlib import Path
from typing import Union

MODEL_REMAPPING = {"kokoro": "kokoro", "csm": "sesame", "sesame": "sesame"}  # (tts/utils.py:17-22 maps "csm" to the sesame package)


def get_model_path(path_or_hf_repo: str, revision=None) -> Path:
    p = Path(path_or_hf_repo)
    if p.exists():
        return p
    from huggingface_hub import snapshot_download

    return Path(snapshot_download(path_or_hf_repo, revision=revision, local_files_only=True,
                                  allow_patterns=["*.json", "*.safetensors", "*.txt"]))


def load_config(model_path: Union[str, Path]) -> dict:
    model_path = get_model_path(model_path) if isinstance(model_path, str) else model_path
    try:
        with open(model_path / "config.json", encoding="utf-8") as f:
            return json.load(f)
    except FileNotFoundError as exc:
        raise FileNotFoundError(f"Config not found at {model_path}") from exc


def get_model_and_args(model_type: str, model_name):
    """utils.py:77-121: the model families this engine serves: kokoro and sesame (CSM); anything else raises like the reference."""
    model_type = MODEL_REMAPPING.get(model_type, model_type)
    for part in model_name or []:
        if part in MODEL_REMAPPING:
            model_type = MODEL_REMAPPING[part]
    if model_type == "kokoro":
        from . import kokoro

        return kokoro, model_type
    if model_type == "sesame":
        from . import sesame

        return sesame, model_type
    msg = f"Model type {model_type} not supported."
    logging.error(msg)
    raise ValueError(msg)


def load_model(model_path, lazy: bool = False, strict: bool = True, compute_dtype: str = None, quantization_kernel: str = "exact", **kwargs):
    """Returns a ready `kokoro.Model` (weights folded, packed and resident in HBM).
    Raises FileNotFoundError when no safetensors are found, ValueError for an unsupported model type.
    quantization_kernel (only matters for a checkpoint with config["quantization"]): "exact" (default) multiplies by the dequantised
    weights `scale * q + bias` with the mode's ordinary kernels, i.e. the reference's arithmetic (tts/utils.py:241-260); "mxfp8" is the
    opt-in that re-quantises an 8-bit checkpoint's linears to e4m3 and runs them on the block-scaled fp8 matrix instruction."""
    if quantization_kernel not in ("exact", "mxfp8"):
        raise ValueError(f"quantization_kernel must be 'exact' or 'mxfp8', not {quantization_kernel!r}")
    if isinstance(model_path, str):
        model_name = model_path.lower().rstrip("/").split("/")[-1].split("-")
        path = get_model_path(model_path)
    elif isinstance(model_path, Path):
        path = model_path
        parts = path.parts
        model_name = parts[parts.index("hub") + 1].lower().split("--")[-1].split("-") if "hub" in parts else path.name.lower().split("-")
    else:
        raise ValueError(f"Invalid model path type: {type(model_path)}")
    config = load_config(path)
    model_type = config.get("model_type") or (model_name[0] if model_name else None)
    weight_files = glob.glob(str(path / "*.safetensors"))
    if not weight_files:
        logging.error(f"No safetensors found in {path}")
        raise FileNotFoundError(f"No safetensors found in {path}")
    arch, model_type = get_model_and_args(model_type, model_name)
    from safetensors import safe_open

    weights = {}
    for wf in weight_files:
        with safe_open(wf, framework="pt") as f:
            for k in f.keys():
                weights[k] = f.get_tensor(k)
    if compute_dtype is None:
        import torch

        any_bf16 = any(getattr(v, "dtype", None) == torch.bfloat16 for v in weights.values())
        compute_dtype = "bfloat16" if any_bf16 else "float32"  # the checkpoint dtype decides, as in the reference
    if model_type == "sesame":
        # sesame.Model(config) has no ModelConfig (tts/utils.py:226-230 passes the dict); the checkpoint dtype picks the weight storage
        model = arch.Model(dict(config, **{k: v for k, v in kwargs.items() if k in ("mimi_path", "text_tokenizer")}),
                           weight_dtype="bfloat16" if compute_dtype == "bfloat16" else "float32", mimi=kwargs.get("mimi"))
        model.load_weights(weights, strict=strict)
        return model
    quantization = config.get("quantization", None)
    if quantization is not None:  # utils.py:241-260: MLX affine group quantisation -> dequantised here, see quant.py
        import numpy as np
        import torch

        from .quant import dequantize_checkpoint

        as_np = {k: (v.float().numpy() if isinstance(v, torch.Tensor) and v.dtype in (torch.bfloat16, torch.float16) else
                     (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v))) for k, v in weights.items()}
        per_layer = {k: v for k, v in quantization.items() if k not in ("group_size", "bits")}  # custom per layer quantizations (utils.py:244-246)
        weights = dequantize_checkpoint(as_np, int(quantization["group_size"]), int(quantization["bits"]), per_layer)
    cfg = arch.ModelConfig.from_dict(config)
    # default: the dequantised weights on the ordinary kernels -- what the reference computes.  Opt-in "mxfp8": an 8-bit checkpoint in
    # bf16 mode runs its quantised linears on the fp8 matrix instruction (kk_set_quantization)
    q8 = None
    if quantization_kernel == "mxfp8":
        if quantization is None or int(quantization["bits"]) != 8 or int(quantization["group_size"]) % 32 != 0 or compute_dtype != "bfloat16":
            raise ValueError("quantization_kernel='mxfp8' needs an 8-bit checkpoint (group size a multiple of 32) in bfloat16 mode")
        q8 = {"group_size": int(quantization["group_size"]), "bits": 8}
    model = arch.Model(cfg, compute_dtype=compute_dtype, quantization=q8)
    model.load_weights(weights, strict=strict)
    return model
