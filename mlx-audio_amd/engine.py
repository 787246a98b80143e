"""Batched device engine over the C ABI.  PyTorch is used for device memory and streams only;
every arithmetic operation of the path runs in libkokoro_hip.so."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import KKConfig, check


def make_kk_config(cfg: dict, compute_dtype: str = "float32") -> KKConfig:
    ist, pb = cfg["istftnet"], cfg["plbert"]
    k = KKConfig()
    k.n_token, k.hidden_dim, k.style_dim, k.n_layer = cfg["n_token"], cfg["hidden_dim"], cfg["style_dim"], cfg["n_layer"]
    k.max_dur, k.text_encoder_kernel_size = cfg["max_dur"], cfg["text_encoder_kernel_size"]
    k.plbert_hidden, k.plbert_heads = pb["hidden_size"], pb["num_attention_heads"]
    k.plbert_intermediate, k.plbert_max_pos = pb["intermediate_size"], pb["max_position_embeddings"]
    k.plbert_layers, k.plbert_embedding = pb["num_hidden_layers"], pb.get("embedding_size", 128)
    k.decoder_hidden = cfg.get("decoder_hidden", 1024)
    k.upsample_initial_channel = ist["upsample_initial_channel"]
    k.n_upsamples = len(ist["upsample_rates"])
    for i, (u, ks) in enumerate(zip(ist["upsample_rates"], ist["upsample_kernel_sizes"])):
        k.upsample_rates[i], k.upsample_kernel_sizes[i] = u, ks
    k.n_resblock_kernels = len(ist["resblock_kernel_sizes"])
    for i, (ks, dil) in enumerate(zip(ist["resblock_kernel_sizes"], ist["resblock_dilation_sizes"])):
        k.resblock_kernel_sizes[i] = ks
        for j in range(3):
            k.resblock_dilations[i][j] = dil[j]
    k.gen_istft_n_fft, k.gen_istft_hop_size = ist["gen_istft_n_fft"], ist["gen_istft_hop_size"]
    k.compute_dtype = {"float32": _lib.KK_F32, "bfloat16": _lib.KK_BF16}[compute_dtype]
    return k


_NP2KK = {np.dtype(np.float32): _lib.KK_F32, np.dtype(np.float16): _lib.KK_F16}


class _SharedModel:
    """One finalized, IMMUTABLE kk_model (the weights on the device), shared by every engine / context made from it; destroyed when the last
    of them is gone."""

    def __init__(self, lib, cfg: dict, weights, compute_dtype: str, device, quantization: Optional[dict]):
        self.lib = lib
        self.h = C.c_void_p()
        kc = make_kk_config(cfg, compute_dtype)
        check(lib.kk_create(C.byref(kc), C.byref(self.h)), "kk_create")
        if quantization is not None:  # load_model's quantization branch (tts/utils.py:241-260): `weights` are the dequantised ones
            check(lib.kk_set_quantization(self.h, int(quantization["group_size"]), int(quantization["bits"])), "kk_set_quantization")
        for name, arr in weights.items():
            self._load(name, arr)
        with torch.cuda.device(device):
            check(lib.kk_finalize(self.h, C.c_void_p(torch.cuda.current_stream(device).cuda_stream)), "kk_finalize")

    def _load(self, name: str, arr) -> None:
        if isinstance(arr, torch.Tensor):
            if arr.dtype == torch.bfloat16:
                raw = arr.contiguous().view(torch.int16).cpu().numpy()
                dt = _lib.KK_BF16
            else:
                raw = arr.detach().cpu().contiguous().numpy()
                dt = _NP2KK.get(raw.dtype)
        else:
            raw = np.ascontiguousarray(arr)
            dt = _NP2KK.get(raw.dtype)
            if dt is None:
                raw = raw.astype(np.float32)
                dt = _lib.KK_F32
        shape = (C.c_int64 * raw.ndim)(*raw.shape)
        check(self.lib.kk_load_tensor(self.h, name.encode(), dt, shape, raw.ndim, raw.ctypes.data_as(C.c_void_p)), f"kk_load_tensor({name})")

    def __del__(self):
        try:
            if getattr(self, "h", None) and self.h.value:
                self.lib.kk_destroy(self.h)
                self.h = C.c_void_p()
        except Exception:
            pass


class KokoroEngine:
    """One kk_context (graph cache, side stream, debug / profile state) + its workspace on a finalized kk_model, and the batch helpers.
    `new_context()` gives a sibling engine on the SAME model (one copy of the weights): one engine per stream / thread in flight."""

    def __init__(self, cfg: dict, weights: Optional[Dict[str, np.ndarray]] = None, compute_dtype: str = "float32", device: Optional[torch.device] = None,
                 quantization: Optional[dict] = None, share: Optional["KokoroEngine"] = None):
        if not torch.cuda.is_available():
            raise _lib.KokoroHipError("KokoroEngine needs a GPU (torch.cuda.is_available() is False)")
        self.lib = _lib.load()
        if share is not None:
            cfg, compute_dtype, device, quantization = share.cfg, share.compute_dtype, share.device, share.quantization
        self.cfg = cfg
        self.compute_dtype = compute_dtype
        self.device = device or torch.device("cuda", torch.cuda.current_device())
        self.quantization = quantization
        self._model = share._model if share is not None else _SharedModel(self.lib, cfg, weights, compute_dtype, self.device, quantization)
        self._m = self._model.h  # kk_model*: immutable, shared
        self._h = C.c_void_p()   # kk_context*: this engine's own
        with torch.cuda.device(self.device):
            check(self.lib.kk_context_create(self._m, C.byref(self._h)), "kk_context_create")
        self._ws = None
        self._graph = False
        self._graph_bufs = {}
        self.upsample = int(np.prod(cfg["istftnet"]["upsample_rates"])) * cfg["istftnet"]["gen_istft_hop_size"] * 2  # samples / frame

    def new_context(self) -> "KokoroEngine":
        """A second context on the same weights (own graph cache, side stream, workspace): for another stream / thread in flight."""
        return KokoroEngine(self.cfg, share=self)

    def quantized_layers(self) -> int:
        return int(self.lib.kk_quantized_layers(self._m))

    def force(self, flags: int) -> None:
        """kk_debug_force_generic on this context (A/B tests)."""
        self.lib.kk_debug_force_generic(self._h, int(flags))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self.lib.kk_context_destroy(self._h)  # before the model: _model is released after this
                self._h = C.c_void_p()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, B: int, Tmax: int, Fmax: int) -> torch.Tensor:
        n = int(self.lib.kk_context_workspace_bytes(self._h, B, Tmax, Fmax))
        if n == 0:
            raise _lib.KokoroHipError("kk_workspace_bytes returned 0")
        if self._ws is None or self._ws.numel() < n:
            self._ws = None
            self._ws = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self._ws

    # ------------------------------------------------------------------ batch helpers
    def pack_ids(self, utterances: Sequence[Sequence[int]]):
        """[0, ids..., 0] rows (kokoro.py:135), zero padded to the longest."""
        lens = [len(u) + 2 for u in utterances]
        Tmax = max(lens)
        ids = np.zeros((len(utterances), Tmax), np.int32)
        for b, u in enumerate(utterances):
            ids[b, 1 : 1 + len(u)] = np.asarray(u, np.int32)
        return (
            torch.from_numpy(ids).to(self.device),
            torch.tensor(lens, dtype=torch.int32, device=self.device),
            Tmax,
        )

    def forward(
        self,
        ids: torch.Tensor,
        lens: torch.Tensor,
        ref_s: torch.Tensor,
        speed: torch.Tensor,
        Fmax: int,
        forced_dur: Optional[torch.Tensor] = None,
        noise_mode: int = _lib.NOISE_PHILOX,
        sine_noise: Optional[torch.Tensor] = None,
        seed: int = 0,
        out: Optional[torch.Tensor] = None,
    ):
        """One kk_forward call on the current stream.  Returns (wav [B, samples_per_frame*Fmax] float32,
        pred_dur [B, Tmax] int32, nframes [B] int32); nothing is synchronised."""
        B, Tmax = ids.shape
        self._last_B = B
        ws = self.workspace(B, Tmax, Fmax)
        if self._graph:
            # graph replay is keyed on every pointer: the outputs live in buffers that persist across calls
            key = (B, Tmax, Fmax)
            if key not in self._graph_bufs:
                self._graph_bufs[key] = (torch.empty((B, self.upsample * Fmax), dtype=torch.float32, device=self.device),
                                         torch.empty((B, Tmax), dtype=torch.int32, device=self.device),
                                         torch.empty((B,), dtype=torch.int32, device=self.device))
            gw, pred, nfr = self._graph_bufs[key]
            wav = out if out is not None else gw
        else:
            wav = out if out is not None else torch.empty((B, self.upsample * Fmax), dtype=torch.float32, device=self.device)
            pred = torch.empty((B, Tmax), dtype=torch.int32, device=self.device)
            nfr = torch.empty((B,), dtype=torch.int32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(
            self.lib.kk_forward(
                self._h, self._stream(), B, Tmax, p(ids), p(lens), p(ref_s), p(speed), p(forced_dur), Fmax, noise_mode, p(sine_noise),
                C.c_uint64(seed), p(ws), ws.numel(), p(wav), p(pred), p(nfr),
            ),
            "kk_forward",
        )
        return wav, pred, nfr

    def set_graph_mode(self, on: bool = True) -> None:
        """kk_set_graph_mode: repeated forward() calls with the same tensors become one hipGraphLaunch.  The returned
        wav / pred_dur / nframes are then views of buffers that the next call with the same shapes overwrites."""
        check(self.lib.kk_set_graph_mode(self._h, 1 if on else 0), "kk_set_graph_mode")
        self._graph = bool(on)

    def forward_text(self, ids, lens, ref_s, speed):
        B, Tmax = ids.shape
        self._last_B = B
        ws = self.workspace(B, Tmax, 0)
        pred = torch.empty((B, Tmax), dtype=torch.int32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr())
        check(self.lib.kk_forward_text(self._h, self._stream(), B, Tmax, p(ids), p(lens), p(ref_s), p(speed), p(ws), ws.numel(), p(pred)),
              "kk_forward_text")
        return pred

    def forward_audio(self, B, Tmax, lens, ref_s, dur, Fmax, noise_mode=_lib.NOISE_PHILOX, sine_noise=None, seed=0, out=None):
        # the workspace must be the one kk_forward_text just used (same B, Tmax).  The text stage's results live in its first
        # kk_workspace_bytes(B, Tmax, 0) bytes at offsets that do not depend on Fmax (bump allocation in a fixed order), so a workspace that
        # is too small for this Fmax -- the caller only learns Fmax from the predicted durations -- is re-allocated and that prefix copied
        need = int(self.lib.kk_context_workspace_bytes(self._h, B, Tmax, Fmax))
        if self._ws is None:
            raise _lib.KokoroHipError("forward_audio: run forward_text first (its results live in the workspace)")
        if self._ws.numel() < need:
            keep = min(int(self.lib.kk_context_workspace_bytes(self._h, B, Tmax, 0)), self._ws.numel())
            grown = torch.empty(need, dtype=torch.uint8, device=self.device)
            grown[:keep].copy_(self._ws[:keep])
            self._ws = grown
        ws = self._ws
        self._last_B = B
        wav = out if out is not None else torch.empty((B, self.upsample * Fmax), dtype=torch.float32, device=self.device)
        nfr = torch.empty((B,), dtype=torch.int32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        check(
            self.lib.kk_forward_audio(self._h, self._stream(), B, Tmax, p(lens), p(ref_s), p(dur), Fmax, noise_mode, p(sine_noise),
                                      C.c_uint64(seed), p(ws), ws.numel(), p(wav), p(nfr)),
            "kk_forward_audio",
        )
        return wav, nfr

    # ------------------------------------------------------------------ per-kernel-class timing (bench)
    PROFILE_CLASSES = ["conv_generic", "conv_mfma", "instnorm_stats", "adain_act", "lstm", "istft_head", "layernorm", "attention",
                       "source", "stft", "linear_mxfp8"]

    def profile_begin(self, max_launches: int = 200000) -> None:
        check(self.lib.kk_profile_begin(self._h, max_launches), "kk_profile_begin")

    def profile_end(self) -> dict:
        n = len(self.PROFILE_CLASSES)
        ms, fl, by, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)(), (C.c_int64 * n)()
        check(self.lib.kk_profile_end(self._h, n, ms, fl, by, cnt), "kk_profile_end")
        return {k: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "launches": cnt[i]} for i, k in enumerate(self.PROFILE_CLASSES)}

    # ------------------------------------------------------------------ debug hooks (tests)
    def debug_fetch(self, name: str) -> torch.Tensor:
        rows, ch = C.c_int64(), C.c_int64()
        check(self.lib.kk_debug_info(self._h, name.encode(), C.byref(rows), C.byref(ch)), "kk_debug_info")
        B = self._last_B
        out = torch.empty((B, rows.value, ch.value), dtype=torch.float32, device=self.device)
        check(self.lib.kk_debug_fetch(self._h, self._stream(), name.encode(), C.c_void_p(out.data_ptr())), "kk_debug_fetch")
        return out

    def debug_override(self, name: str, t: torch.Tensor) -> None:
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        self._overrides = getattr(self, "_overrides", {})
        self._overrides[name] = t  # keep alive
        check(self.lib.kk_debug_override(self._h, name.encode(), C.c_void_p(t.data_ptr())), "kk_debug_override")

    def debug_clear(self) -> None:
        self.lib.kk_debug_clear(self._h)
        self._overrides = {}
