"""Host mirror of the reference's CSM frame generator surface (mlx_audio/tts/models/sesame/sesame.py:276-415): `SesameModel` with
`setup_caches`, `reset_caches`, `generate_frame(tokens, tokens_mask, input_pos, ...)`.  The arithmetic runs in libkokoro_hip.so
(kk_csm_*, csrc/kk_csm.hip).  The text tokenizer, prompt building and the generation loop (sesame.py:484-817) are host code of the
reference that can call this class unchanged; they need the Llama-3.2 tokenizer files, which are not available offline."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import KokoroHipError, check


def _llama_args(d: dict) -> _lib.KKLlamaArgs:
    a = _lib.KKLlamaArgs()
    for k in ("num_layers", "num_heads", "num_kv_heads", "head_dim", "hidden", "intermediate"):
        setattr(a, k, int(d[k]))
    a.rope_theta, a.rope_factor, a.rms_eps = float(d["rope_theta"]), float(d["rope_factor"]), float(d["rms_eps"])
    return a


class SesameModel:
    def __init__(self, cfg: dict, weights: Optional[Dict[str, np.ndarray]] = None, device: str = "cuda:0", weight_dtype: str = "float32"):
        self.cfg = cfg
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise KokoroHipError("SesameModel needs a GPU: the frame generator has no CPU fallback")
        self.device = torch.device(device)
        kc = _lib.KKCsmConfig()
        kc.text_vocab_size, kc.audio_vocab_size = int(cfg["text_vocab_size"]), int(cfg["audio_vocab_size"])
        kc.audio_num_codebooks, kc.max_seq_len = int(cfg["audio_num_codebooks"]), int(cfg["max_seq_len"])
        kc.backbone, kc.decoder = _llama_args(cfg["backbone"]), _llama_args(cfg["decoder"])
        h = C.c_void_p()
        check(self.lib.kk_csm_create(C.byref(kc), C.byref(h)), "kk_csm_create")
        self._h = h
        # "bfloat16": what load_model does for a bf16 checkpoint (it keeps the checkpoint's dtype, tts/utils.py:217-262) -- the Linear
        # matrices are stored as bf16 and streamed as such by the single-token steps; arithmetic stays fp32
        self.weight_dtype = weight_dtype
        if weight_dtype != "float32":
            check(self.lib.kk_csm_set_weight_dtype(self._h, {"bfloat16": _lib.KK_BF16}[weight_dtype]), "kk_csm_set_weight_dtype")
        self._final = False
        self._ws = None
        self._enabled = False
        self.max_batch = 0
        self._graph = False
        self._gbuf = {}
        if weights is not None:
            self.load_weights(weights)

    def share(self) -> "SesameModel":
        """kk_csm_share: a second generator on the SAME device weights (own KV caches, positions, logits, graph cache) for another stream /
        thread in flight.  It keeps this one alive."""
        import copy

        if not self._final:
            raise KokoroHipError("SesameModel.share: load_weights first")
        other = copy.copy(self)
        h = C.c_void_p()
        check(self.lib.kk_csm_share(self._h, C.byref(h)), "kk_csm_share")
        other._h, other._parent = h, self
        other._ws, other._enabled, other.max_batch, other._graph, other._gbuf = None, False, 0, False, {}
        return other

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.kk_csm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_weights(self, weights: Dict[str, np.ndarray]) -> "SesameModel":
        with torch.cuda.device(self.device):
            for name, arr in weights.items():
                a = np.ascontiguousarray(np.asarray(arr, np.float32))
                shp = (C.c_int64 * a.ndim)(*a.shape)
                check(self.lib.kk_csm_load_tensor(self._h, name.encode(), shp, a.ndim, a.ctypes.data_as(C.c_void_p)), "kk_csm_load_tensor")
            check(self.lib.kk_csm_finalize(self._h, self._stream()), "kk_csm_finalize")
        self._final = True
        return self

    # ---- sesame.py:320-345
    def setup_caches(self, max_batch_size: int) -> None:
        with torch.cuda.device(self.device):
            check(self.lib.kk_csm_setup_caches(self._h, int(max_batch_size)), "kk_csm_setup_caches")
        self._gbuf = {}  # the library dropped its captured graphs with the old caches; their staging buffers go with them
        self._enabled = True
        self.max_batch = int(max_batch_size)

    def set_padding(self, pads) -> None:
        """kk_csm_set_padding: left padding (in frames) of every stream's prompt, for batches whose prompts differ in length.  Call on an
        empty cache, before the prompt block; reset_caches clears it."""
        arr = (C.c_int32 * len(pads))(*[int(p) for p in pads])
        check(self.lib.kk_csm_set_padding(self._h, len(pads), arr), "kk_csm_set_padding")

    def caches_are_enabled(self) -> bool:
        return self._enabled

    def reset_caches(self) -> None:
        check(self.lib.kk_csm_reset_caches(self._h), "kk_csm_reset_caches")

    def set_graph_mode(self, on: bool = True) -> None:
        """kk_csm_set_graph_mode: single-token frames are replayed as one hipGraph.  Inputs are staged in buffers that persist across
        calls (a graph is keyed on its pointers); the returned codes are a view that the next frame overwrites."""
        check(self.lib.kk_csm_set_graph_mode(self._h, 1 if on else 0), "kk_csm_set_graph_mode")
        self._graph = bool(on)

    @property
    def position(self) -> int:
        return int(self.lib.kk_csm_position(self._h))

    # ---- sesame.py:349-395
    def generate_frame(self, tokens, tokens_mask, input_pos=None, temperature: float = 0.0, top_k: int = 50, uniforms=None) -> torch.Tensor:
        """tokens [B, S, n_cb+1] int, tokens_mask same shape; `input_pos` (the reference's argument) is checked against the cache position.
        Returns codes [B, n_cb] int32 on the device."""
        assert self.caches_are_enabled(), "backbone caches are not enabled"
        tokens = torch.as_tensor(tokens).to(device=self.device, dtype=torch.int32).contiguous()
        mask = torch.as_tensor(tokens_mask).to(device=self.device, dtype=torch.float32).contiguous()
        B, S, W = tokens.shape
        ncb = self.cfg["audio_num_codebooks"]
        if W != ncb + 1 or tuple(mask.shape) != (B, S, W):
            raise ValueError(f"tokens / tokens_mask must be [B, S, {ncb + 1}]")
        if input_pos is not None:
            ip = np.array(input_pos.cpu() if isinstance(input_pos, torch.Tensor) else input_pos)
            if ip.shape != (B, S) or not np.array_equal(ip, np.broadcast_to(self.position + np.arange(S), (B, S))):
                raise ValueError("input_pos must continue the cache: position + arange(S) for every item")
        u = None
        if uniforms is not None:
            u = torch.as_tensor(uniforms).to(device=self.device, dtype=torch.float32).contiguous()
            if tuple(u.shape) != (B, ncb):
                raise ValueError(f"uniforms must be [B, {ncb}]")
        if self._graph and S == 1:
            key = (B, u is not None)
            if key not in self._gbuf:
                self._gbuf[key] = (torch.empty_like(tokens), torch.empty_like(mask), torch.empty((B, ncb), dtype=torch.float32, device=self.device),
                                   torch.empty((B, ncb), dtype=torch.int32, device=self.device))
            gt, gm, gu, gc = self._gbuf[key]
            gt.copy_(tokens)
            gm.copy_(mask)
            if u is not None:
                gu.copy_(u)
                u = gu
            tokens, mask = gt, gm
        with torch.cuda.device(self.device):
            need = int(self.lib.kk_csm_workspace_bytes(self._h, B, S))
            if need == 0:
                raise KokoroHipError("kk_csm_workspace_bytes failed")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            codes = self._gbuf[(B, u is not None)][3] if (self._graph and S == 1) else torch.empty((B, ncb), dtype=torch.int32, device=self.device)
            check(self.lib.kk_csm_generate_frame(self._h, self._stream(), B, S, C.c_void_p(tokens.data_ptr()), C.c_void_p(mask.data_ptr()),
                                                 float(temperature), int(top_k), C.c_void_p(u.data_ptr()) if u is not None else None,
                                                 C.c_void_p(self._ws.data_ptr()), need, C.c_void_p(codes.data_ptr())), "kk_csm_generate_frame")
        self._last_B = B
        return codes

    def debug_logits(self) -> torch.Tensor:
        B, ncb, V = self._last_B, self.cfg["audio_num_codebooks"], self.cfg["audio_vocab_size"]
        out = torch.empty((ncb, B, V), dtype=torch.float32, device=self.device)
        check(self.lib.kk_csm_debug_logits(self._h, self._stream(), B, C.c_void_p(out.data_ptr())), "kk_csm_debug_logits")
        return out
