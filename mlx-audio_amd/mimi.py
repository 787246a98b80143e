"""Host mirror of the reference's Mimi codec surface for the DECODE path (mlx_audio/codec/models/mimi/mimi.py): `mimi_202407`,
`Mimi(cfg)`, `Mimi.decode(codes)`, `.sample_rate`, `.frame_rate`.  The arithmetic runs in libkokoro_hip.so (kk_mimi_*, csrc/kk_mimi.hip);
PyTorch allocates device memory and provides the stream.  `Mimi.encode` (mimi.py:138-145) runs on the fp32 kernels; the streaming
`*_step` entry points are not built."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Dict, List

import numpy as np
import torch

from . import _lib
from ._lib import KokoroHipError, check


@dataclass
class MimiConfig:
    """The fields of MimiConfig / SeanetConfig / TransformerConfig (mimi.py:27-38) that the decode path reads."""
    dim: int = 512
    nq: int = 32
    bins: int = 2048
    qdim: int = 256
    num_heads: int = 8
    num_layers: int = 8
    dim_feedforward: int = 2048
    nfilters: int = 64
    ratios: List[int] = field(default_factory=lambda: [8, 6, 5, 4])
    ksize: int = 7
    residual_ksize: int = 3
    last_ksize: int = 3
    upsample_stride: int = 2
    compress: int = 2
    rope_base: float = 10000.0
    sample_rate: float = 24000.0
    frame_rate: float = 12.5

    @classmethod
    def from_dict(cls, d: dict) -> "MimiConfig":
        return cls(**{k: v for k, v in d.items() if k in cls.__dataclass_fields__})


def mimi_202407(num_codebooks: int) -> MimiConfig:
    """mimi.py:41-101."""
    return MimiConfig(nq=num_codebooks)


class Mimi:
    def __init__(self, cfg: MimiConfig, weights: Dict[str, np.ndarray] | None = None, device: str = "cuda:0", compute_dtype: str = "float32"):
        self.cfg = cfg
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise KokoroHipError("Mimi needs a GPU: the decode path has no CPU fallback")
        self.device = torch.device(device)
        kc = _lib.KKMimiConfig()
        for k in ("dim", "nq", "bins", "qdim", "num_heads", "num_layers", "dim_feedforward", "nfilters", "ksize", "residual_ksize", "last_ksize",
                  "upsample_stride", "compress"):
            setattr(kc, k, int(getattr(cfg, k)))
        kc.n_ratios = len(cfg.ratios)
        for i, r in enumerate(cfg.ratios):
            kc.ratios[i] = int(r)
        kc.rope_base = float(cfg.rope_base)
        if compute_dtype not in ("float32", "bfloat16"):
            raise ValueError("compute_dtype must be float32 or bfloat16")
        kc.compute_dtype = _lib.KK_BF16 if compute_dtype == "bfloat16" else _lib.KK_F32
        h = C.c_void_p()
        check(self.lib.kk_mimi_create(C.byref(kc), C.byref(h)), "kk_mimi_create")
        self._h = h
        self._final = False
        self._ws = None
        self._streams = {}
        if weights is not None:
            self.load_weights(weights)

    def share(self) -> "Mimi":
        """A second Mimi on the SAME kk_mimi (its weights are immutable after finalize; decode / encode take a caller-owned workspace and the
        streaming state lives in stream objects): own workspace and streams, for another HIP stream / host thread in flight.  Keeps this one
        alive; only the original destroys the codec."""
        import copy

        other = copy.copy(self)
        other._parent, other._ws, other._streams = self, None, {}
        return other

    def __del__(self):
        try:
            for st in getattr(self, "_streams", {}).values():
                self.lib.kk_mimi_stream_destroy(st["h"])
            self._streams = {}
            if getattr(self, "_h", None) and getattr(self, "_parent", None) is None:
                self.lib.kk_mimi_destroy(self._h)
            self._h = None
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def load_weights(self, weights: Dict[str, np.ndarray]) -> "Mimi":
        """MLX-side names and layouts (what Mimi.load_pytorch_weights produces, mimi.py:184-249).  Unknown names are ignored the way
        `strict=False` would; missing ones fail in finalize with the parameter's name."""
        with torch.cuda.device(self.device):
            for name, arr in weights.items():
                a = np.ascontiguousarray(np.asarray(arr, np.float32))
                shp = (C.c_int64 * max(a.ndim, 1))(*(a.shape if a.ndim else (1,)))
                check(self.lib.kk_mimi_load_tensor(self._h, name.encode(), shp, max(a.ndim, 1), a.ctypes.data_as(C.c_void_p)), "kk_mimi_load_tensor")
            check(self.lib.kk_mimi_finalize(self._h, self._stream()), "kk_mimi_finalize")
        self._final = True
        return self

    @property
    def frame_rate(self) -> float:
        return self.cfg.frame_rate

    @property
    def sample_rate(self) -> float:
        return self.cfg.sample_rate

    def decode(self, codes) -> torch.Tensor:
        """codes [B, nq, Nf] integer -> pcm [B, 1, 1920 * Nf] float32 on the device (mimi.py:147-154)."""
        if not self._final:
            raise KokoroHipError("Mimi.decode: load_weights first")
        codes = torch.as_tensor(codes).to(device=self.device, dtype=torch.int32).contiguous()
        if codes.ndim != 3 or codes.shape[1] != self.cfg.nq:
            raise ValueError(f"codes must be [B, {self.cfg.nq}, Nf], got {tuple(codes.shape)}")
        B, _, Nf = codes.shape
        self._last_B = B
        with torch.cuda.device(self.device):
            need = int(self.lib.kk_mimi_workspace_bytes(self._h, B, Nf))
            if need == 0:
                raise KokoroHipError("kk_mimi_workspace_bytes failed")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            spf = int(self.lib.kk_mimi_samples_per_frame(self._h))
            pcm = torch.empty((B, 1, spf * Nf), dtype=torch.float32, device=self.device)
            check(self.lib.kk_mimi_decode(self._h, self._stream(), B, Nf, C.c_void_p(codes.data_ptr()), C.c_void_p(self._ws.data_ptr()), need,
                                          C.c_void_p(pcm.data_ptr())), "kk_mimi_decode")
        return pcm

    def encode(self, xs) -> torch.Tensor:
        """pcm [B, 1, N] float -> codes [B, nq, ceil-chain(N)] int32 on the device (mimi.py:138-145)."""
        if not self._final:
            raise KokoroHipError("Mimi.encode: load_weights first")
        xs = torch.as_tensor(xs).to(device=self.device, dtype=torch.float32).contiguous()
        if xs.ndim != 3 or xs.shape[1] != 1:
            raise ValueError(f"pcm must be [B, 1, N], got {tuple(xs.shape)}")
        B, _, N = xs.shape
        self._last_B = B
        with torch.cuda.device(self.device):
            need = int(self.lib.kk_mimi_encode_workspace_bytes(self._h, B, N))
            if need == 0:
                raise KokoroHipError("kk_mimi_encode_workspace_bytes failed (no encoder parameters loaded?)")
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            nf = int(self.lib.kk_mimi_encode_frames(self._h, N))
            codes = torch.empty((B, self.cfg.nq, nf), dtype=torch.int32, device=self.device)
            check(self.lib.kk_mimi_encode(self._h, self._stream(), B, N, C.c_void_p(xs.data_ptr()), C.c_void_p(self._ws.data_ptr()), need,
                                          C.c_void_p(codes.data_ptr())), "kk_mimi_encode")
        return codes

    # ---- streaming (mimi.py:156-168): the state lives in library-owned stream objects, one per direction
    def _open_stream(self, slot: str, encoder: bool, B: int, chunk: int, max_batch: int, max_frames: int, max_chunk: int = 0):
        """The stream of one direction.  The reference's step functions take ANY number of frames per call and continue the conv / KV state
        (mimi.py:156-168), so a call with another F continues the open stream (kk_mimi_stream_set_chunk).  The state buffers are sized for the
        largest step, fixed when the stream is created: the first call's F, or `max_chunk`.  A fresh (or reset) stream is re-created when the
        batch or the step outgrows it; a stream that has consumed frames cannot grow -- that is an error, never a silent restart."""
        st = self._streams.get(slot)
        if st is not None:
            fresh = int(self.lib.kk_mimi_stream_frames(st["h"])) == 0
            if (st["maxb"] < B or st["maxchunk"] < chunk) and not fresh:
                what = f"batch {B} > {st['maxb']}" if st["maxb"] < B else f"{chunk} frames per step > {st['maxchunk']}"
                raise ValueError(f"Mimi stream: {what} on a stream that has already consumed frames; open it with max_batch= / max_chunk= large "
                                 "enough for every step, or call reset_stream() first")
            if fresh and (st["maxb"] < max(B, max_batch) or st["maxchunk"] < max(chunk, max_chunk)):  # a fresh / reset stream may be re-sized
                self._close_slot(slot)
                st = None
        if st is None:
            h = C.c_void_p()
            mb, mc = max(B, max_batch), max(chunk, max_chunk)
            check(self.lib.kk_mimi_stream_create_chunked(self._h, int(encoder), mb, max(max_frames, mc), mc, C.byref(h)), "kk_mimi_stream_create_chunked")
            st = self._streams[slot] = {"h": h, "maxb": mb, "maxchunk": mc, "chunk": mc, "ws": None}
        if st["chunk"] != chunk:
            check(self.lib.kk_mimi_stream_set_chunk(st["h"], chunk), "kk_mimi_stream_set_chunk")
            st["chunk"] = chunk
        need = int(self.lib.kk_mimi_stream_workspace_bytes(st["h"], B))
        if need == 0:
            raise KokoroHipError("kk_mimi_stream_workspace_bytes failed")
        if st["ws"] is None or st["ws"].numel() < need:
            st["ws"] = torch.empty(need, dtype=torch.uint8, device=self.device)
        return st

    def decode_step(self, codes, max_batch: int = 0, max_frames: int = 2048, max_chunk: int = 0) -> torch.Tensor:
        """codes [B, nq, F] -> pcm [B, 1, 1920 * F] (the reference feeds F = 1).  The first call (or the first after reset_stream) fixes B and
        the largest F (`max_chunk` raises it); later calls may carry fewer frames and CONTINUE the stream, as the reference's do."""
        if not self._final:
            raise KokoroHipError("Mimi.decode_step: load_weights first")
        codes = torch.as_tensor(codes).to(device=self.device, dtype=torch.int32)
        if codes.ndim != 3 or codes.shape[1] != self.cfg.nq or codes.shape[2] < 1:
            raise ValueError(f"codes must be [B, {self.cfg.nq}, F >= 1], got {tuple(codes.shape)}")
        B, _, F = codes.shape
        codes = codes.contiguous()
        with torch.cuda.device(self.device):
            st = self._open_stream("dec", False, B, F, max_batch, max_frames, max_chunk)
            spf = int(self.lib.kk_mimi_samples_per_frame(self._h))
            pcm = torch.empty((B, 1, spf * F), dtype=torch.float32, device=self.device)
            self._last_B = B
            check(self.lib.kk_mimi_decode_step(st["h"], self._stream(), B, C.c_void_p(codes.data_ptr()), C.c_void_p(st["ws"].data_ptr()), st["ws"].numel(),
                                               C.c_void_p(pcm.data_ptr())), "kk_mimi_decode_step")
        return pcm

    def encode_step(self, xs, max_batch: int = 0, max_frames: int = 2048, max_chunk: int = 0) -> torch.Tensor:
        """mimi.py:156-161: pcm [B, 1, 1920 * F] -> codes [B, nq, F], continuing the encoder's state.  Whole code frames only (the
        reference's modules also hold back a partial stride; a partial FRAME would return nothing until completed, so it is refused)."""
        if not self._final:
            raise KokoroHipError("Mimi.encode_step: load_weights first")
        xs = torch.as_tensor(xs).to(device=self.device, dtype=torch.float32)
        if xs.ndim != 3 or xs.shape[1] != 1:
            raise ValueError(f"xs must be [B, 1, N], got {tuple(xs.shape)}")
        spf = int(self.lib.kk_mimi_samples_per_frame(self._h))
        B, _, N = xs.shape
        if N < spf or N % spf:
            raise ValueError(f"encode_step takes whole code frames ({spf} samples each), got {N} samples")
        F = N // spf
        xs = xs.reshape(B, N).contiguous()
        with torch.cuda.device(self.device):
            st = self._open_stream("enc", True, B, F, max_batch, max_frames, max_chunk)
            codes = torch.empty((B, self.cfg.nq, F), dtype=torch.int32, device=self.device)
            self._last_B = B
            check(self.lib.kk_mimi_encode_step(st["h"], self._stream(), B, C.c_void_p(xs.data_ptr()), C.c_void_p(st["ws"].data_ptr()), st["ws"].numel(),
                                               C.c_void_p(codes.data_ptr())), "kk_mimi_encode_step")
        return codes

    def reset_stream(self) -> None:
        """Mimi.reset_state (mimi.py:131-137): both directions start over."""
        for st in self._streams.values():
            check(self.lib.kk_mimi_stream_reset(st["h"]), "kk_mimi_stream_reset")

    reset_state = reset_stream

    def _close_slot(self, slot: str) -> None:
        st = self._streams.pop(slot, None)
        if st is not None:
            self.lib.kk_mimi_stream_destroy(st["h"])

    def close_stream(self) -> None:
        for slot in list(self._streams):
            self._close_slot(slot)

    def debug_fetch(self, name: str) -> torch.Tensor:
        rows, ch = C.c_int64(0), C.c_int64(0)
        check(self.lib.kk_mimi_debug_info(self._h, name.encode(), C.byref(rows), C.byref(ch)), "kk_mimi_debug_info")
        B = self._last_B
        out = torch.empty((B, rows.value, ch.value), dtype=torch.float32, device=self.device)
        check(self.lib.kk_mimi_debug_fetch(self._h, self._stream(), name.encode(), C.c_void_p(out.data_ptr())), "kk_mimi_debug_fetch")
        return out


class MimiStreamingDecoder:
    """mimi.py:264-306: keeps the codec's decode state across calls and decodes tokens frame by frame with `decode_step`."""

    def __init__(self, mimi: Mimi) -> None:
        self._mimi = mimi
        self.reset()

    def reset(self) -> None:
        self._mimi.reset_stream()

    def decode_frames(self, tokens) -> torch.Tensor:
        tokens = torch.as_tensor(tokens)
        if tokens.ndim == 2:
            tokens = tokens[None]
        return torch.cat([self._mimi.decode_step(tokens[:, :, t : t + 1]) for t in range(tokens.shape[-1])], dim=-1)
