"""CPU restatement (PyTorch-CPU fp32) of the reference's Mimi codec DECODE path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench tools' cpu_baseline legs may import this file; the product path
(mlx-audio_amd/) never does.

Follows, as text (reference paths relative to the repository root):
  mlx_audio/codec/models/mimi/mimi.py:147-154            Mimi.decode
  mlx_audio/codec/models/mimi/modules/quantization.py    EuclideanCodebook.decode :41-43, embedding = embedding_sum /
                                                         max(cluster_usage, 1e-5) :25-28, split RVQ decode :97-101,135-139,178-182
  mlx_audio/codec/models/mimi/modules/conv.py            causal StreamableConv1d :244-263, StreamableConvTranspose1d :323-333,
                                                         depth-wise ConvTrUpsample1d :379-401 (eye-masked dense weight :82-95)
  mlx_audio/codec/models/mimi/modules/transformer.py     Attention :62-104 (RoPE traditional, base 10000; **no mask is passed in the
                                                         non-streaming call**, so decode() attends bidirectionally), MlpNoGating
                                                         :126-134 (gelu_approx), TransformerLayer :137-177 (LayerScale), ProjectedTransformer
  mlx_audio/codec/models/mimi/modules/seanet.py          SeanetResnetBlock :55-116, DecoderLayer :189-225, SeanetDecoder :228-283

PARITY WITH MLX IS UNPINNED: `mlx` is not installable here and the reference's only Mimi test
(mlx_audio/codec/tests/test_mimi.py) pins shapes only -- codes [1,32,63] -> pcm [1,1,120960]; that known answer is
checked in tests/test_mimi_oracle.py.  MLX op semantics assumed (upstream knowledge, not in tree): conv weights [O,K,I]
channels-last, conv_transpose1d without kernel flip (mimi.py:228-237 maps PyTorch [I,O,K] -> transpose(1,2,0)),
nn.gelu_approx = tanh form, nn.RoPE(traditional=True) rotates (x[2i], x[2i+1]) by pos * base^(-2i/D).
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32)


class MimiOracle:
    def __init__(self, w: dict, cfg: dict):
        self.w = {k: np.asarray(v, np.float32) for k, v in w.items()}
        self.cfg = cfg

    # ---- quantization.py ----------------------------------------------------------------------------------------
    def codebook(self, prefix: str) -> torch.Tensor:
        usage = np.maximum(self.w[prefix + ".cluster_usage"], 1e-5)[:, None]  # quantization.py:25-28
        return t(self.w[prefix + ".embedding_sum"] / usage)

    def rvq_decode(self, which: str, codes: np.ndarray) -> torch.Tensor:
        """codes [B, n, Nf] -> [B, dim, Nf]: sum of code-book rows (:97-101), then the 1x1 output projection (:135-139)."""
        q = None
        for i in range(codes.shape[1]):
            e = self.codebook(f"quantizer.{which}.vq.layers.{i}.codebook")[torch.as_tensor(codes[:, i].astype(np.int64))]  # [B,Nf,256]
            q = e if q is None else q + e
        q = q.transpose(1, 2)  # [B,256,Nf]
        w = t(self.w[f"quantizer.{which}.output_proj.weight"]).permute(0, 2, 1)  # MLX [O,K,I] -> torch [O,I,K]
        return F.conv1d(q, w)

    def quantizer_decode(self, codes: np.ndarray) -> torch.Tensor:
        q = self.rvq_decode("rvq_first", codes[:, :1])
        if codes.shape[1] > 1:
            q = q + self.rvq_decode("rvq_rest", codes[:, 1:])  # quantization.py:178-182
        return q

    # ---- conv.py -------------------------------------------------------------------------------------------------
    def causal_conv(self, x: torch.Tensor, prefix: str, dilation: int = 1) -> torch.Tensor:
        """StreamableConv1d.__call__ (:244-263), causal, stride 1: left pad (k-1)*d, pad_mode 'constant' (zeros)."""
        w = t(self.w[prefix + ".conv.conv.weight"])  # [O,K,I]
        b = t(self.w[prefix + ".conv.conv.bias"]) if prefix + ".conv.conv.bias" in self.w else None
        k = w.shape[1]
        pad = (k - 1) * dilation
        return F.conv1d(F.pad(x, (pad, 0)), w.permute(0, 2, 1), b, dilation=dilation)

    def causal_convtr(self, x: torch.Tensor, prefix: str, stride: int) -> torch.Tensor:
        """StreamableConvTranspose1d.__call__ (:323-333), causal: full transposed conv, then the last k - stride samples go."""
        w = t(self.w[prefix + ".convtr.convtr.weight"])  # MLX [O,K,I]
        b = t(self.w[prefix + ".convtr.convtr.bias"]) if prefix + ".convtr.convtr.bias" in self.w else None
        k = w.shape[1]
        y = F.conv_transpose1d(x, w.permute(2, 0, 1), b, stride=stride)
        return y[..., : y.shape[-1] - max(k - stride, 0)]

    def upsample(self, x: torch.Tensor) -> torch.Tensor:
        """ConvTrUpsample1d (:379-401): depth-wise transposed conv k = 2*stride, no bias, causal."""
        s = self.cfg["upsample_stride"]
        w = t(self.w["upsample.convtr.convtr.convtr.weight"])  # [1, 2s, C]
        C = w.shape[2]
        y = F.conv_transpose1d(x, w[0].transpose(0, 1)[:, None, :], None, stride=s, groups=C)
        return y[..., : y.shape[-1] - s]

    # ---- transformer.py ------------------------------------------------------------------------------------------
    def rope(self, x: torch.Tensor) -> torch.Tensor:
        """nn.RoPE(head_dim, traditional=True, base=max_period), offset 0; x [B,H,T,D]."""
        D = x.shape[-1]
        pos = torch.arange(x.shape[2], dtype=torch.float32)[:, None]
        inv = torch.tensor(float(self.cfg["rope_base"]), dtype=torch.float32) ** (-torch.arange(0, D // 2, dtype=torch.float32) / (D // 2))
        ang = pos * inv[None, :]
        c, s = torch.cos(ang), torch.sin(ang)
        x0, x1 = x[..., 0::2], x[..., 1::2]
        out = torch.empty_like(x)
        out[..., 0::2] = x0 * c - x1 * s
        out[..., 1::2] = x0 * s + x1 * c
        return out

    def transformer(self, x: torch.Tensor, prefix: str = "decoder_transformer") -> torch.Tensor:
        """ProjectedTransformer with conv_layout (:213-247): [B,C,T] -> [B,T,C] -> layers -> back."""
        cfg = self.cfg
        H = cfg["num_heads"]
        x = x.transpose(1, 2)
        Bn, T, C = x.shape
        hd = C // H
        for i in range(cfg["num_layers"]):
            p = f"{prefix}.transformer.layers.{i}"
            n1 = F.layer_norm(x, (C,), t(self.w[p + ".norm1.weight"]), t(self.w[p + ".norm1.bias"]), 1e-5)
            qkv = (n1 @ t(self.w[p + ".self_attn.in_proj.weight"]).T).reshape(Bn, T, 3, H, hd)
            q, k, v = [qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3)]
            q, k = self.rope(q), self.rope(k)
            att = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v  # mask=None in the reference call (:171)
            att = att.permute(0, 2, 1, 3).reshape(Bn, T, C) @ t(self.w[p + ".self_attn.out_proj.weight"]).T
            x = x + att * t(self.w[p + ".layer_scale_1.scale"])
            n2 = F.layer_norm(x, (C,), t(self.w[p + ".norm2.weight"]), t(self.w[p + ".norm2.bias"]), 1e-5)
            h = n2 @ t(self.w[p + ".gating.linear1.weight"]).T
            h = 0.5 * h * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (h + 0.044715 * h ** 3)))  # nn.gelu_approx
            x = x + (h @ t(self.w[p + ".gating.linear2.weight"]).T) * t(self.w[p + ".layer_scale_2.scale"])
        return x.transpose(1, 2)

    # ---- seanet.py -----------------------------------------------------------------------------------------------
    def seanet_decoder(self, x: torch.Tensor, inter: dict | None = None) -> torch.Tensor:
        x = self.causal_conv(x, "decoder.init_conv1d")
        for l, r in enumerate(self.cfg["ratios"]):
            p = f"decoder.layers.{l}"
            x = self.causal_convtr(F.elu(x), p + ".upsample", r)
            res = x
            y = self.causal_conv(F.elu(x), p + ".residuals.0.block.0", 1)
            y = self.causal_conv(F.elu(y), p + ".residuals.0.block.1", 1)
            x = y + res  # true_skip
            if inter is not None:
                inter[f"layer{l}"] = x.numpy()
        return self.causal_conv(F.elu(x), "decoder.final_conv1d")

    # ---- encode side: seanet.py:120-212 (encoder), conv.py:232-263,350-367 (strided causal conv, 'edge' resampler),
    #      quantization.py:35-39,62-66,84-92,128-133,170-176 (argmin of c2 - x.e, residual loop), mimi.py:138-145
    def strided_causal_conv(self, x: torch.Tensor, prefix: str, stride: int, pad_mode: str = "constant") -> torch.Tensor:
        w = t(self.w[prefix + ".conv.conv.weight"])
        b = t(self.w[prefix + ".conv.conv.bias"]) if prefix + ".conv.conv.bias" in self.w else None
        k = w.shape[1]
        padding_total = k - stride
        L = x.shape[-1]
        nframes = max(L + padding_total - k, 0) / stride + 1.0  # get_extra_padding_for_conv1d (conv.py:200-209)
        ideal = (int(math.ceil(nframes)) - 1) * stride + k - padding_total
        extra = max(0, ideal - L)
        x = F.pad(x, (padding_total, extra), mode="replicate" if pad_mode == "edge" else "constant")
        return F.conv1d(x, w.permute(0, 2, 1), b, stride=stride)

    def seanet_encoder(self, x: torch.Tensor) -> torch.Tensor:
        x = self.causal_conv(x, "encoder.init_conv1d")
        for l, r in enumerate(reversed(self.cfg["ratios"])):
            p = f"encoder.layers.{l}"
            res = x
            y = self.causal_conv(F.elu(x), p + ".residuals.0.block.0", 1)
            y = self.causal_conv(F.elu(y), p + ".residuals.0.block.1", 1)
            x = y + res
            x = self.strided_causal_conv(F.elu(x), p + ".downsample", r)
        return self.causal_conv(F.elu(x), "encoder.final_conv1d")

    def rvq_encode(self, which: str, x: torch.Tensor, n: int, trace: list | None = None) -> np.ndarray:
        """x [B, dim, T] -> codes [B, n, T].  Distances c2 - x.e with c2 = |e|^2 / 2 (quantization.py:27-28,35-39)."""
        w = t(self.w[f"quantizer.{which}.input_proj.weight"]).permute(0, 2, 1)
        residual = F.conv1d(x, w).transpose(1, 2)  # [B, T, qdim]
        codes = []
        for i in range(n):
            E = self.codebook(f"quantizer.{which}.vq.layers.{i}.codebook")
            c2 = (E * E).sum(-1) / 2
            dist = c2[None, None, :] - residual @ E.T
            idx = dist.argmin(-1)
            if trace is not None:
                trace.append((which, i, dist.numpy().copy(), idx.numpy().copy()))
            residual = residual - E[idx]
            codes.append(idx.numpy())
        return np.stack(codes, axis=1)

    def encode(self, pcm: np.ndarray, trace: list | None = None, return_inter: bool = False):
        """Mimi.encode (mimi.py:138-145): pcm [B, 1, N] -> codes [B, nq, ceil(N / 1920)]."""
        with torch.no_grad():
            inter = {}
            x = self.seanet_encoder(t(pcm))
            inter["seanet"] = x.numpy()
            x = self.transformer(x, prefix="encoder_transformer")
            inter["transformer"] = x.numpy()
            x = self.strided_causal_conv(x, "downsample.conv", self.cfg["upsample_stride"], pad_mode="edge")
            inter["downsampled"] = x.numpy()
            codes = self.rvq_encode("rvq_first", x, 1, trace)
            if self.cfg["nq"] > 1:
                codes = np.concatenate([codes, self.rvq_encode("rvq_rest", x, self.cfg["nq"] - 1, trace)], axis=1)
        return (codes, inter) if return_inter else codes

    # ---- mimi.py:147-154 -------------------------------------------------------------------------------------------
    def decode(self, codes: np.ndarray, return_inter: bool = False):
        with torch.no_grad():
            inter = {}
            x = self.quantizer_decode(np.asarray(codes))
            inter["quantized"] = x.numpy()
            x = self.upsample(x)
            inter["upsampled"] = x.numpy()
            x = self.transformer(x)
            inter["transformer"] = x.numpy()
            pcm = self.seanet_decoder(x, inter).numpy()
        return (pcm, inter) if return_inter else pcm


class MimiStreamOracle:
    """Streaming decode, restated with the reference's explicit per-module state (TEST INFRASTRUCTURE ONLY):
      mimi.py:163-168            Mimi.decode_step = quantizer.decode -> upsample.step -> decoder_transformer(cache) -> decoder.step
      mimi.py:264-306            MimiStreamingDecoder.reset / decode_frames (frame by frame, outputs concatenated)
      conv.py:265-293            StreamableConv1d.step: first call left-pads ksize - stride zeros, `_prev_xs` carries the unconsumed input
      conv.py:335-351            StreamableConvTranspose1d.step: the last ksize - stride outputs are held back (`_prev_ys`, bias removed
                                 before it is added to the next step's head)
      seanet.py:113-116,219-223,277-283   residual block / decoder layer / decoder .step (the skip add of equal lengths is a plain add)
      transformer.py:79-104      Attention with a KV cache: RoPE offset = cache.offset, keys limited to the last `context` cached ones
                                 plus the new block, NO mask (the positions of one step see each other)
    The cache is mlx_lm-style (append, fetch everything): not in the reference tree -- upstream knowledge, PARITY UNPINNED."""

    def __init__(self, w: dict, cfg: dict, context: int = 250):
        self.o = MimiOracle(w, cfg)
        self.cfg = cfg
        self.context = context
        self.reset()

    def reset(self):
        self.state = {}
        # mimi.py:129-130: one cache list per transformer (encoder_cache / decoder_cache), each with its own offset
        self.kv = {"decoder_transformer": [None] * self.cfg["num_layers"], "encoder_transformer": [None] * self.cfg["num_layers"]}
        self.offset = {"decoder_transformer": 0, "encoder_transformer": 0}

    # conv.py:265-293 (dilation 1 everywhere in Mimi): the first call left-pads ksize - stride (zeros, or the first sample for 'edge'),
    # `_prev_xs` carries the unconsumed input, nframes = (len + stride - ksize) // stride outputs leave
    def conv_step(self, x: torch.Tensor, prefix: str, stride: int = 1, pad_mode: str = "constant") -> torch.Tensor:
        w = t(self.o.w[prefix + ".conv.conv.weight"])
        b = t(self.o.w[prefix + ".conv.conv.bias"]) if prefix + ".conv.conv.bias" in self.o.w else None
        k = w.shape[1]
        if x.shape[-1] == 0:
            return x.new_zeros(x.shape[0], w.shape[0], 0)
        if prefix not in self.state:
            x = F.pad(x, (k - stride, 0), mode="replicate" if pad_mode == "edge" else "constant")
        else:
            x = torch.cat([self.state[prefix], x], dim=-1)
        nframes = max(x.shape[-1] + stride - k, 0) // stride
        self.state[prefix] = x[..., nframes * stride :]
        if nframes == 0:
            return x.new_zeros(x.shape[0], w.shape[0], 0)
        return F.conv1d(x[..., : (nframes - 1) * stride + k], w.permute(0, 2, 1), b, stride=stride)

    # conv.py:335-351
    def convtr_step(self, x: torch.Tensor, prefix: str, stride: int, depthwise: bool = False) -> torch.Tensor:
        w = t(self.o.w[prefix + ".convtr.convtr.weight"])
        bkey = prefix + ".convtr.convtr.bias"
        b = t(self.o.w[bkey]) if bkey in self.o.w else None
        k = w.shape[1]
        if depthwise:
            ys = F.conv_transpose1d(x, w[0].transpose(0, 1)[:, None, :], None, stride=stride, groups=w.shape[2])
        else:
            ys = F.conv_transpose1d(x, w.permute(2, 0, 1), b, stride=stride)
        if prefix in self.state:
            prev = self.state[prefix]
            if b is not None:
                prev = prev - b[None, :, None]
            pt = prev.shape[-1]
            ys = torch.cat([ys[..., :pt] + prev, ys[..., pt:]], dim=-1)
        inv = k - stride
        ot = ys.shape[-1]
        self.state[prefix] = ys[..., ot - inv :]
        return ys[..., : ot - inv]

    def transformer_step(self, x: torch.Tensor, which: str = "decoder_transformer") -> torch.Tensor:
        cfg, o = self.cfg, self.o
        kv = self.kv[which]
        H = cfg["num_heads"]
        x = x.transpose(1, 2)
        Bn, T, C = x.shape
        hd = C // H
        pos = torch.arange(self.offset[which], self.offset[which] + T, dtype=torch.float32)[:, None]
        inv = torch.tensor(float(cfg["rope_base"]), dtype=torch.float32) ** (-torch.arange(0, hd // 2, dtype=torch.float32) / (hd // 2))
        c, s = torch.cos(pos * inv[None]), torch.sin(pos * inv[None])

        def rope(v):
            out = torch.empty_like(v)
            out[..., 0::2] = v[..., 0::2] * c - v[..., 1::2] * s
            out[..., 1::2] = v[..., 0::2] * s + v[..., 1::2] * c
            return out

        for i in range(cfg["num_layers"]):
            p = f"{which}.transformer.layers.{i}"
            n1 = F.layer_norm(x, (C,), t(o.w[p + ".norm1.weight"]), t(o.w[p + ".norm1.bias"]), 1e-5)
            qkv = (n1 @ t(o.w[p + ".self_attn.in_proj.weight"]).T).reshape(Bn, T, 3, H, hd)
            q, k, v = [qkv[:, :, j].permute(0, 2, 1, 3) for j in range(3)]
            q, k = rope(q), rope(k)
            if kv[i] is not None:
                k = torch.cat([kv[i][0], k], dim=2)
                v = torch.cat([kv[i][1], v], dim=2)
            kv[i] = (k, v)
            klen = k.shape[2]
            tgt = T + min(self.context, klen - T)
            k, v = k[:, :, klen - tgt :], v[:, :, klen - tgt :]
            att = torch.softmax((q @ k.transpose(-1, -2)) * hd ** -0.5, dim=-1) @ v
            att = att.permute(0, 2, 1, 3).reshape(Bn, T, C) @ t(o.w[p + ".self_attn.out_proj.weight"]).T
            x = x + att * t(o.w[p + ".layer_scale_1.scale"])
            n2 = F.layer_norm(x, (C,), t(o.w[p + ".norm2.weight"]), t(o.w[p + ".norm2.bias"]), 1e-5)
            h = n2 @ t(o.w[p + ".gating.linear1.weight"]).T
            h = 0.5 * h * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (h + 0.044715 * h ** 3)))
            x = x + (h @ t(o.w[p + ".gating.linear2.weight"]).T) * t(o.w[p + ".layer_scale_2.scale"])
        self.offset[which] += T
        return x.transpose(1, 2)

    def seanet_step(self, x: torch.Tensor) -> torch.Tensor:
        x = self.conv_step(x, "decoder.init_conv1d")
        for l, r in enumerate(self.cfg["ratios"]):
            p = f"decoder.layers.{l}"
            x = self.convtr_step(F.elu(x), p + ".upsample", r)
            res = x
            y = self.conv_step(F.elu(x), p + ".residuals.0.block.0")
            y = self.conv_step(F.elu(y), p + ".residuals.0.block.1")
            x = y + res
        return self.conv_step(F.elu(x), "decoder.final_conv1d")

    def decode_step(self, codes: np.ndarray, return_inter: bool = False):
        """codes [B, nq, t] (t = 1 in MimiStreamingDecoder) -> pcm [B, 1, samples_per_frame * t]."""
        with torch.no_grad():
            x = self.o.quantizer_decode(np.asarray(codes))
            x = self.convtr_step(x, "upsample.convtr", self.cfg["upsample_stride"], depthwise=True)
            xt = self.transformer_step(x)
            pcm = self.seanet_step(xt).numpy()
        return (pcm, {"upsampled": x.numpy(), "transformer": xt.numpy()}) if return_inter else pcm

    # seanet.py:120-212 through each module's step (SeaNetEncoder.step: init conv, per layer residual block + ELU + strided conv, last conv)
    def seanet_encoder_step(self, x: torch.Tensor) -> torch.Tensor:
        x = self.conv_step(x, "encoder.init_conv1d")
        for l, r in enumerate(reversed(self.cfg["ratios"])):
            p = f"encoder.layers.{l}"
            res = x
            y = self.conv_step(F.elu(x), p + ".residuals.0.block.0")
            y = self.conv_step(F.elu(y), p + ".residuals.0.block.1")
            x = y + res
            x = self.conv_step(F.elu(x), p + ".downsample", stride=r)
        return self.conv_step(F.elu(x), "encoder.final_conv1d")

    def encode_step(self, pcm: np.ndarray, return_inter: bool = False):
        """Mimi.encode_step (mimi.py:156-161): pcm [B, 1, n] -> codes [B, nq, frames completed by this call] (whole strides only leave
        each module; the rest waits in its `_prev_xs`)."""
        with torch.no_grad():
            x = self.seanet_encoder_step(t(pcm))
            xt = self.transformer_step(x, "encoder_transformer") if x.shape[-1] else x
            xd = self.conv_step(xt, "downsample.conv", stride=self.cfg["upsample_stride"], pad_mode="edge")
            if xd.shape[-1] == 0:
                codes = np.zeros((pcm.shape[0], self.cfg["nq"], 0), np.int64)
            else:
                codes = self.o.rvq_encode("rvq_first", xd, 1)
                if self.cfg["nq"] > 1:
                    codes = np.concatenate([codes, self.o.rvq_encode("rvq_rest", xd, self.cfg["nq"] - 1)], axis=1)
        return (codes, {"seanet": x.numpy(), "transformer": xt.numpy(), "downsampled": xd.numpy()}) if return_inter else codes

    def decode_frames(self, codes: np.ndarray) -> np.ndarray:
        return np.concatenate([self.decode_step(codes[:, :, i : i + 1]) for i in range(codes.shape[-1])], axis=-1)
