"""CPU oracle for the Kokoro-82M acoustic path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This file is a CPU restatement (numpy + PyTorch-CPU fp32) of the reference's algorithm
for the hot path named in BASELINE.json:

    mlx_audio/tts/models/kokoro/kokoro.py:120-170      Model.__call__
    mlx_audio/tts/models/kokoro/modules.py             TextEncoder / LSTM / Albert / prosody
    mlx_audio/tts/models/kokoro/istftnet.py            AdaIN / resblocks / SineGen / Generator / Decoder
    mlx_audio/utils.py:10-158                          hanning / stft / istft
    mlx_audio/tts/models/interpolate.py:6-108          interpolate / interpolate1d

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it, and
only as the checker.  The product path (mlx-audio_amd/) never imports anything from here.

Pinning status.  The reference's tests hold exactly one family of numeric known answers
on this path (mlx_audio/tts/tests/test_interpolate.py:40-84); `tests/test_oracle_pins.py`
checks this file against them, against the hyper-parameter dict at
mlx_audio/tts/tests/test_models.py:92-122 and the derived 81.76 M parameter count.
No reference test pins a waveform, duration vector or intermediate activation, the
reference's runtime (`mlx`) is not installable here, and no checkpoint exists offline:
WAVEFORM-LEVEL PARITY WITH MLX IS THEREFORE **UNPINNED** (see DESIGN.md "Oracle").

Layout conventions.  Like the reference's outer code this file keeps activations as
[B, C, L] ("NCL") between modules.  Weights are kept in the reference's *MLX-side*
layout (post-`sanitize`, kokoro.py:172-252): conv weights [C_out, K, C_in/groups],
linear weights [out, in], LSTM `Wx_*/Wh_*/bias_*`.

The three random draws of the reference (istftnet.py:563, 620, 679) are explicit
inputs here: `rand_ini` (provably without effect on the output, see `sine_gen`),
`sine_noise` and the unused `noi_source`.
"""

from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# mlx_audio/utils.py
# ----------------------------------------------------------------------------------------


def hanning(size: int) -> np.ndarray:
    """Symmetric Hann, python-double then float32 (utils.py:10-14)."""
    return np.array(
        [0.5 * (1 - math.cos(2 * math.pi * n / (size - 1))) for n in range(size)],
        dtype=np.float32,
    )


def stft(x: np.ndarray, n_fft=800, hop_length=None, win_length=None, center=True) -> np.ndarray:
    """utils.py:52-101 with window="hann", pad_mode="reflect".  x: [N] float32 -> [frames, n_fft//2+1] complex64."""
    if hop_length is None:
        hop_length = n_fft // 4
    if win_length is None:
        win_length = n_fft
    w = hanning(win_length)
    if w.shape[0] < n_fft:
        w = np.concatenate([w, np.zeros(n_fft - w.shape[0], np.float32)])
    x = np.asarray(x, np.float32)
    if center:
        p = n_fft // 2
        prefix = x[1 : p + 1][::-1]
        suffix = x[-(p + 1) : -1][::-1]
        x = np.concatenate([prefix, x, suffix])
    num_frames = 1 + (x.shape[0] - n_fft) // hop_length
    if num_frames <= 0:
        raise ValueError("Input is too short")
    idx = np.arange(num_frames)[:, None] * hop_length + np.arange(n_fft)[None, :]
    frames = x[idx] * w[None, :]
    return np.fft.rfft(frames.astype(np.float32), axis=-1).astype(np.complex64)


def istft(x: np.ndarray, hop_length=None, win_length=None, center=True, length=None) -> np.ndarray:
    """utils.py:104-158 with window="hann".  x: [bins, frames] complex64 -> [samples] float32.

    The window is the *periodic* Hann `hanning(win_length + 1)[:-1]` (utils.py:121) and the
    overlap-add is normalised by sum(w), not sum(w^2) (utils.py:143-150).
    The scatter-add order of the reference (frame-major, utils.py:138-147) is kept so the
    float32 rounding of the 4-term sums is the same.
    """
    if win_length is None:
        win_length = (x.shape[1] - 1) * 2
    if hop_length is None:
        hop_length = win_length // 4
    w = hanning(win_length + 1)[:-1]
    num_frames = x.shape[1]
    t = (num_frames - 1) * hop_length + win_length
    frames_time = np.fft.irfft(x.astype(np.complex64), axis=0).astype(np.float32).T  # [frames, win]
    upd = (frames_time * w[None, :]).astype(np.float32)
    reconstructed = np.zeros(t, np.float32)
    window_sum = np.zeros(t, np.float32)
    # frame-major accumulation == sequential scatter-add; vectorised per within-hop phase
    nseg = win_length // hop_length
    assert nseg * hop_length == win_length
    # process frames in ascending order for every output sample: frame f contributes segment j
    # (offsets j*hop..j*hop+hop-1) to output block f+j.  For a fixed output block the
    # ascending-frame order is j descending.
    nblk = num_frames + nseg - 1
    rec_b = np.zeros((nblk, hop_length), np.float32)
    win_b = np.zeros((nblk, hop_length), np.float32)
    for j in range(nseg - 1, -1, -1):
        rec_b[j : j + num_frames] += upd[:, j * hop_length : (j + 1) * hop_length]
        win_b[j : j + num_frames] += w[None, j * hop_length : (j + 1) * hop_length]
    reconstructed = rec_b.reshape(-1)[:t]
    window_sum = win_b.reshape(-1)[:t]
    nz = window_sum != 0
    out = reconstructed.copy()
    out[nz] = reconstructed[nz] / window_sum[nz]
    if center and length is None:
        out = out[win_length // 2 : -win_length // 2]
    if length is not None:
        out = out[:length]
    return out


# ----------------------------------------------------------------------------------------
# mlx_audio/tts/models/interpolate.py
# ----------------------------------------------------------------------------------------


def interpolate1d(inp: np.ndarray, size: int, mode: str = "linear", align_corners=None) -> np.ndarray:
    """interpolate.py:57-108.  inp [N, C, W] float32.  Float32 op order follows the reference:
    `mx.arange(size)` is int32, python scalars are weakly typed (-> float32)."""
    inp = np.asarray(inp, np.float32)
    batch, channels, in_width = inp.shape
    if size < 1:
        size = 1
    if in_width < 1:
        in_width = 1
    if mode == "nearest":
        if size == 1:
            indices = np.array([0])
        else:
            scale = in_width / size
            indices = np.floor(np.arange(size, dtype=np.int32).astype(np.float32) * np.float32(scale)).astype(np.int32)
            indices = np.clip(indices, 0, in_width - 1)
        return inp[:, :, indices]
    if align_corners and size > 1:
        x = np.arange(size, dtype=np.int32).astype(np.float32) * np.float32((in_width - 1) / (size - 1))
    else:
        if size == 1:
            x = np.array([0.0], np.float32)
        else:
            x = np.arange(size, dtype=np.int32).astype(np.float32) * np.float32(in_width / size)
            if not align_corners:
                x = (x + np.float32(0.5 * (in_width / size))) - np.float32(0.5)
    if in_width == 1:
        return np.broadcast_to(inp, (batch, channels, size)).copy()
    x_low = np.floor(x).astype(np.int32)
    x_high = np.minimum(x_low + 1, in_width - 1)
    x_frac = (x - x_low.astype(np.float32)).astype(np.float32)
    # NOTE: x_low is *not* clamped at 0 (interpolate.py:96); a negative index wraps to the
    # end of the array exactly as MLX/numpy integer-array indexing does.
    y_low = inp[:, :, x_low]
    y_high = inp[:, :, x_high]
    return (y_low * (np.float32(1) - x_frac)[None, None, :] + y_high * x_frac[None, None, :]).astype(np.float32)


def interpolate(inp: np.ndarray, size=None, scale_factor=None, mode="nearest", align_corners=None) -> np.ndarray:
    """interpolate.py:6-54.  `scale_factor` reaches this function as an mx.array scalar on the hot
    path (istftnet.py:568-578), so `shape * scale_factor` is a float32 product."""
    inp = np.asarray(inp)
    ndim = inp.ndim
    if ndim < 3:
        raise ValueError(f"Expected at least 3D input (N, C, D1), got {ndim}D")
    spatial = ndim - 2
    if size is not None and scale_factor is not None:
        raise ValueError("Only one of size or scale_factor should be defined")
    if size is None and scale_factor is None:
        raise ValueError("One of size or scale_factor must be defined")
    if size is not None and not isinstance(size, (list, tuple)):
        size = [size] * spatial
    if scale_factor is not None and not isinstance(scale_factor, (list, tuple)):
        scale_factor = [scale_factor] * spatial
    if size is None:
        size = []
        for i in range(spatial):
            prod = np.float32(inp.shape[i + 2]) * np.float32(scale_factor[i])
            size.append(max(1, int(np.ceil(prod))))
    if spatial == 1:
        return interpolate1d(inp, size[0], mode, align_corners)
    raise ValueError(f"Only 1D interpolation currently supported, got {spatial}D")


# ----------------------------------------------------------------------------------------
# helpers (torch CPU fp32)
# ----------------------------------------------------------------------------------------


def _t(a) -> torch.Tensor:
    return torch.as_tensor(np.asarray(a, dtype=np.float32))


def weight_norm(v: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """istftnet.py:53-93 with dim=0 (as called at :130): norm over axes (1, 2), +1e-7 on the norm."""
    norm = torch.sqrt(torch.sum(v * v, dim=(1, 2), keepdim=True))
    return v / (norm + 1e-7) * g


def leaky_relu(x: torch.Tensor, slope: float) -> torch.Tensor:
    return torch.where(x > 0, x, x * slope)


class KokoroOracle:
    """Restatement of kokoro.py:Model with explicit noise inputs.  `w` maps MLX-side parameter
    names to numpy arrays (see oracle/synth.py for the inventory)."""

    def __init__(self, weights: Dict[str, np.ndarray], config: dict, dtype=torch.float32):
        self.cfg = config
        self.dtype = dtype
        self.w = {k: torch.as_tensor(np.asarray(v, dtype=np.float32)).to(dtype) for k, v in weights.items()}
        ist = config["istftnet"]
        self.ups_rates = list(ist["upsample_rates"])
        self.ups_k = list(ist["upsample_kernel_sizes"])
        self.rb_k = list(ist["resblock_kernel_sizes"])
        self.rb_d = [list(d) for d in ist["resblock_dilation_sizes"]]
        self.n_fft = int(ist["gen_istft_n_fft"])
        self.hop = int(ist["gen_istft_hop_size"])
        self.init_ch = int(ist["upsample_initial_channel"])
        self.upsample_scale = int(np.prod(self.ups_rates)) * self.hop  # istftnet.py:714
        self.nheads = int(config["plbert"]["num_attention_heads"])
        self.nlayers_bert = int(config["plbert"]["num_hidden_layers"])
        self.n_layer = int(config["n_layer"])

    # ---- primitive modules -------------------------------------------------------------

    def linear(self, x, p):
        """mlx nn.Linear: x @ W.T + b."""
        y = x @ self.w[p + ".weight"].T
        if p + ".bias" in self.w:
            y = y + self.w[p + ".bias"]
        return y

    def layer_norm(self, x, p, eps):
        """nn.LayerNorm over the last axis, population variance."""
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        return (x - mean) / torch.sqrt(var + eps) * self.w[p + ".weight"] + self.w[p + ".bias"]

    def conv_weighted(self, x_ncl, p, stride=1, padding=1, dilation=1, groups=1, transpose=False):
        """ConvWeighted.__call__ (istftnet.py:128-170) on an NCL tensor.

        weight = g * v / (||v|| + 1e-7) recomputed from weight_g/weight_v (istftnet.py:130).
        conv1d:            mx.conv1d(x_nlc, w[O,K,I])           == F.conv1d(x_ncl, w.permute(0,2,1))
        conv_transpose1d:  reference passes weight.T = [I',K,O'] (istftnet.py:161-166) for the
                           Generator `ups` (groups == 1) -> true transposed conv with
                           torch weight[in, out, k] = weight_v[in, k, out];
                           for groups > 1 (`pool`) the weight [C,K,1] is used as is ->
                           torch depthwise weight[c, 0, k] = w[c, k, 0].
        """
        w = weight_norm(self.w[p + ".weight_v"], self.w[p + ".weight_g"])
        b = self.w.get(p + ".bias")
        if not transpose:
            return F.conv1d(x_ncl, w.permute(0, 2, 1).contiguous(), b, stride, padding, dilation, groups)
        if groups > 1:
            wt = w.permute(0, 2, 1).contiguous()  # [C, 1, K]
        else:
            wt = w.permute(0, 2, 1).contiguous()  # v is [in, K, out] -> [in, out, K]
        return F.conv_transpose1d(x_ncl, wt, b, stride, padding, 0, groups, dilation)

    def instance_norm(self, x, eps=1e-5):
        """istftnet.py:216-268, affine=False: stats over L per (b, c), ddof 0."""
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        return (x - mean) / torch.sqrt(var + eps)

    def adain(self, x, s, p):
        """AdaIN1d (istftnet.py:327-338)."""
        h = self.linear(s, p + ".fc")[:, :, None]
        c = h.shape[1] // 2
        gamma, beta = h[:, :c], h[:, c:]
        return (1 + gamma) * self.instance_norm(x) + beta

    def adain_resblock1(self, x, s, p, k, dil):
        """AdaINResBlock1 (istftnet.py:341-396) with Snake1D."""
        for j in range(3):
            a1 = self.w[f"{p}.alpha1.{j}"]
            a2 = self.w[f"{p}.alpha2.{j}"]
            xt = self.adain(x, s, f"{p}.adain1.{j}")
            xt = xt + (1 / a1) * (torch.sin(a1 * xt) ** 2)
            xt = self.conv_weighted(xt, f"{p}.convs1.{j}", 1, (k * dil[j] - dil[j]) // 2, dil[j])
            xt = self.adain(xt, s, f"{p}.adain2.{j}")
            xt = xt + (1 / a2) * (torch.sin(a2 * xt) ** 2)
            xt = self.conv_weighted(xt, f"{p}.convs2.{j}", 1, (k - 1) // 2, 1)
            x = xt + x
        return x

    def adain_resblk1d(self, x, s, p, upsample=False):
        """AdainResBlk1d (istftnet.py:825-899)."""
        cin = x.shape[1]
        learned_sc = (p + ".conv1x1.weight_v") in self.w
        # shortcut (istftnet.py:863-872): nearest x2 then optional 1x1 (no bias)
        sc = x
        if upsample:
            sc = sc.repeat_interleave(2, dim=-1)
        if learned_sc:
            sc = self.conv_weighted(sc, p + ".conv1x1", 1, 0, 1)
        # residual (istftnet.py:874-894)
        r = self.adain(x, s, p + ".norm1")
        r = leaky_relu(r, 0.2)
        if upsample:
            r = self.conv_weighted(r, p + ".pool", 2, 1, 1, groups=cin, transpose=True)
            r = F.pad(r, (1, 0))  # zero pad at the FRONT of the time axis (istftnet.py:881)
        r = self.conv_weighted(r, p + ".conv1", 1, 1, 1)
        r = self.adain(r, s, p + ".norm2")
        r = leaky_relu(r, 0.2)
        r = self.conv_weighted(r, p + ".conv2", 1, 1, 1)
        return (r + sc) / math.sqrt(2)

    def lstm(self, x, p):
        """Bidirectional LSTM (modules.py:93-285).  x [B, L, I] -> [B, L, 2H]."""
        outs = []
        for d in ("forward", "backward"):
            Wx, Wh = self.w[f"{p}.Wx_{d}"], self.w[f"{p}.Wh_{d}"]
            bias = self.w[f"{p}.bias_ih_{d}"] + self.w[f"{p}.bias_hh_{d}"]
            x_proj = bias + x @ Wx.T  # mx.addmm(bias, x, Wx.T)
            B, L, _ = x.shape
            H = Wh.shape[1]
            h = torch.zeros(B, H, dtype=x.dtype)
            c = torch.zeros(B, H, dtype=x.dtype)
            hs = [None] * L
            order = range(L) if d == "forward" else range(L - 1, -1, -1)
            WhT = Wh.T.contiguous()
            for idx in order:
                ifgo = x_proj[:, idx, :] + h @ WhT
                i, f, g, o = torch.split(ifgo, H, dim=-1)
                i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
                c = f * c + i * g
                h = o * torch.tanh(c)
                hs[idx] = h
            outs.append(torch.stack(hs, dim=1))
        return torch.cat(outs, dim=-1)

    # ---- Albert (modules.py:438-649) ------------------------------------------------------

    def albert(self, ids: torch.Tensor) -> torch.Tensor:
        """CustomAlbert sequence output for a full-length (unmasked) B=1 sequence."""
        T = ids.shape[1]
        pe = "bert.embeddings."
        emb = (
            self.w[pe + "word_embeddings.weight"][ids]
            + self.w[pe + "position_embeddings.weight"][torch.arange(T)][None]
            + self.w[pe + "token_type_embeddings.weight"][torch.zeros_like(ids)]
        )
        x = self.layer_norm(emb, pe + "LayerNorm", 1e-12)
        x = self.linear(x, "bert.encoder.embedding_hidden_mapping_in")
        lp = "bert.encoder.albert_layer_groups.0.albert_layers.0."
        nh = self.nheads
        hd = x.shape[-1] // nh
        mask = torch.zeros(1, 1, 1, T, dtype=x.dtype)  # (1 - 1) * -10000 (modules.py:641-643)
        for _ in range(self.nlayers_bert):
            q = self.linear(x, lp + "attention.query").view(1, T, nh, hd).permute(0, 2, 1, 3)
            k = self.linear(x, lp + "attention.key").view(1, T, nh, hd).permute(0, 2, 1, 3)
            v = self.linear(x, lp + "attention.value").view(1, T, nh, hd).permute(0, 2, 1, 3)
            sc = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + mask
            pr = torch.softmax(sc, dim=-1)
            ctx = (pr @ v).permute(0, 2, 1, 3).reshape(1, T, nh * hd)
            ctx = self.linear(ctx, lp + "attention.dense")
            att = self.layer_norm(ctx + x, lp + "attention.LayerNorm", 1e-12)
            ff = self.linear(att, lp + "ffn")
            ff = ff * 0.5 * (1.0 + torch.erf(ff / math.sqrt(2.0)))  # nn.GELU() exact
            ff = self.linear(ff, lp + "ffn_output")
            x = self.layer_norm(ff + att, lp + "full_layer_layer_norm", 1e-12)
        return x

    # ---- prosody predictor (modules.py:288-411) ----------------------------------------------

    def ada_layer_norm(self, x, s, p, eps=1e-5):
        """AdaLayerNorm (modules.py:71-90).  x [1, T, C], s [1, 128]."""
        h = self.linear(s, p + ".fc")
        c = h.shape[1] // 2
        gamma, beta = h[:, None, :c], h[:, None, c:]
        mean = x.mean(-1, keepdim=True)
        var = ((x - mean) ** 2).mean(-1, keepdim=True)
        xn = (x - mean) / torch.sqrt(var + eps)
        return (1 + gamma) * xn + beta

    def duration_encoder(self, d_en, s):
        """DurationEncoder (modules.py:392-411), B=1, no masked positions.  d_en [1,512,T] -> [1,T,640]."""
        T = d_en.shape[-1]
        sb = s[:, :, None].expand(1, s.shape[-1], T)  # [1,128,T]
        x = torch.cat([d_en, sb], dim=1)  # [1,640,T]
        for i in range(self.n_layer):
            xl = self.lstm(x.transpose(1, 2), f"predictor.text_encoder.lstms.{2 * i}")  # [1,T,512]
            xn = self.ada_layer_norm(xl, s, f"predictor.text_encoder.lstms.{2 * i + 1}")
            x = torch.cat([xn.transpose(1, 2), sb], dim=1)
        return x.transpose(1, 2)

    def f0n_train(self, en, s):
        """ProsodyPredictor.F0Ntrain (modules.py:355-377).  en [1,640,F] -> F0, N [1, 2F]."""
        x = self.lstm(en.transpose(1, 2), "predictor.shared").transpose(1, 2)  # [1,512,F]
        outs = []
        for name in ("F0", "N"):
            y = x
            y = self.adain_resblk1d(y, s, f"predictor.{name}.0")
            y = self.adain_resblk1d(y, s, f"predictor.{name}.1", upsample=True)
            y = self.adain_resblk1d(y, s, f"predictor.{name}.2")
            wproj = self.w[f"predictor.{name}_proj.weight"]  # [1, 1, 256] (O,K,I)
            y = F.conv1d(y, wproj.permute(0, 2, 1).contiguous(), self.w[f"predictor.{name}_proj.bias"])
            outs.append(y[:, 0, :])
        return outs[0], outs[1]

    def text_encoder(self, ids):
        """TextEncoder (modules.py:41-68).  ids [1,T] -> [1,512,T]."""
        x = self.w["text_encoder.embedding.weight"][ids].transpose(1, 2)
        k = int(self.cfg["text_encoder_kernel_size"])
        for i in range(self.n_layer):
            x = self.conv_weighted(x, f"text_encoder.cnn.{i}.0", 1, (k - 1) // 2, 1)
            x = self.layer_norm(x.transpose(1, 2), f"text_encoder.cnn.{i}.1", 1e-5).transpose(1, 2)
            x = leaky_relu(x, 0.2)
        x = self.lstm(x.transpose(1, 2), "text_encoder.lstm").transpose(1, 2)
        return x

    # ---- source module (istftnet.py:531-680) ---------------------------------------------------

    def sine_gen(self, f0_up: np.ndarray, rand_ini: Optional[np.ndarray], sine_noise: Optional[np.ndarray]):
        """SineGen.__call__ (istftnet.py:606-623) for harmonic_num=8, sine_amp=0.1, noise_std=0.003,
        voiced_threshold=10.  f0_up [1, N, 1] float32 -> sine_waves [1, N, 9], uv [1, N, 1].

        `rand_ini` is added to sample 0 only (istftnet.py:563-565) and the 1/300 linear
        down-sampling that follows reads samples 300*i+149 and 300*i+150 only
        (interpolate.py:84-98: x = 300*i + 149.5), so it cannot influence the result; it is
        still applied here for faithfulness.
        """
        up = self.upsample_scale
        f0_up = np.asarray(f0_up, np.float32)
        harm = np.arange(1, 10, dtype=np.int32).astype(np.float32)[None, None, :]
        fn = (f0_up * harm).astype(np.float32)
        rad = np.mod((fn / np.float32(24000)).astype(np.float32), np.float32(1)).astype(np.float32)
        if rand_ini is not None:
            ri = np.array(rand_ini, np.float32).copy()
            ri[:, 0] = 0
            rad[:, 0, :] = rad[:, 0, :] + ri
        scale_down = np.float32(1) / np.float32(up)  # 1 / mx.array(int32 300) -> float32
        rad_dn = interpolate(rad.transpose(0, 2, 1), scale_factor=scale_down, mode="linear").transpose(0, 2, 1)
        phase = (np.cumsum(rad_dn, axis=1, dtype=np.float32) * np.float32(2)).astype(np.float32) * np.float32(math.pi)
        # `mx.cumsum(..) * 2 * mx.pi`: left-to-right float32 products
        phase_up = interpolate(
            (phase.transpose(0, 2, 1) * np.float32(up)).astype(np.float32), scale_factor=np.float32(up), mode="linear"
        ).transpose(0, 2, 1)
        sines = np.sin(phase_up.astype(np.float32)).astype(np.float32)
        sine_waves = (sines * np.float32(0.1)).astype(np.float32)
        uv = (f0_up > 10).astype(np.float32)
        noise_amp = (uv * np.float32(0.003) + (np.float32(1) - uv) * np.float32(0.1) / np.float32(3)).astype(np.float32)
        if sine_noise is None:
            noise = np.zeros_like(sine_waves)
        else:
            noise = (noise_amp * np.asarray(sine_noise, np.float32)).astype(np.float32)
        sine_waves = (sine_waves * uv + noise).astype(np.float32)
        return sine_waves, uv

    def source_module(self, f0_up, rand_ini, sine_noise):
        """SourceModuleHnNSF.__call__ (istftnet.py:668-680): tanh(Linear(9 -> 1))."""
        sw, uv = self.sine_gen(f0_up, rand_ini, sine_noise)
        W = self.w["decoder.generator.m_source.l_linear.weight"].float().numpy()
        b = self.w["decoder.generator.m_source.l_linear.bias"].float().numpy()
        merged = np.tanh((sw @ W.T + b).astype(np.float32)).astype(np.float32)
        return merged, uv

    def har_features(self, f0_curve: np.ndarray, rand_ini, sine_noise):
        """Generator front end (istftnet.py:770-775): F0 [1, 2F] -> har [1, 22, 120F+1] (and har_source)."""
        up = self.upsample_scale
        f0_up = np.repeat(np.asarray(f0_curve, np.float32), up, axis=1)[:, :, None]  # nn.Upsample nearest
        har_source, _ = self.source_module(f0_up, rand_ini, sine_noise)
        har_source = har_source[:, :, 0]  # [1, 600F]
        mags, phs = [], []
        for b in range(har_source.shape[0]):
            X = stft(har_source[b], self.n_fft, self.hop, self.n_fft).T  # [11, frames]
            mags.append(np.abs(X).astype(np.float32))
            phs.append(np.arctan2(X.imag, X.real).astype(np.float32))
        har = np.concatenate([np.stack(mags), np.stack(phs)], axis=1)
        return har, har_source

    # ---- Generator / Decoder (istftnet.py:696-963) -------------------------------------------------

    def istft_head(self, x: np.ndarray) -> np.ndarray:
        """istftnet.py:804-806 + MLXSTFT.inverse (:497-523).  x [B, 22, frames] (conv_post output)
        -> audio [B, 1, 5*(frames-1)].  `mlx_unwrap` on sin(.) in [-1, 1] is an exact no-op
        (|diff| <= 2 < pi, istftnet.py:441-442)."""
        nb = self.n_fft // 2 + 1
        x = np.asarray(x, np.float32)
        spec = np.exp(x[:, :nb]).astype(np.float32)
        phase = np.sin(x[:, nb:]).astype(np.float32)
        outs = []
        for b in range(x.shape[0]):
            re = (spec[b] * np.cos(phase[b])).astype(np.float32)
            im = (spec[b] * np.sin(phase[b])).astype(np.float32)
            outs.append(istft((re + 1j * im).astype(np.complex64), self.hop, self.n_fft))
        return np.stack(outs)[:, None, :]

    def generator(self, x, s, f0_curve, rand_ini, sine_noise, inter=None):
        """Generator.__call__ (istftnet.py:769-807).  x [1,512,2F] torch; returns audio np [1,1,600F]."""
        har_np, har_source = self.har_features(f0_curve.float().numpy(), rand_ini, sine_noise)
        har = torch.as_tensor(har_np).to(x.dtype)
        if inter is not None:
            inter["har_source"] = har_source
            inter["har"] = har_np
        gp = "decoder.generator."
        nk = len(self.rb_k)
        nu = len(self.ups_rates)
        for i in range(nu):
            x = leaky_relu(x, 0.1)
            u, k = self.ups_rates[i], self.ups_k[i]
            if i + 1 < nu:
                stride_f0 = int(np.prod(self.ups_rates[i + 1 :]))
                nc_stride, nc_pad = stride_f0, (stride_f0 + 1) // 2
            else:
                nc_stride, nc_pad = 1, 0
            wn = self.w[f"{gp}noise_convs.{i}.weight"]  # nn.Conv1d, [O, K, I]
            x_source = F.conv1d(har, wn.permute(0, 2, 1).contiguous(), self.w[f"{gp}noise_convs.{i}.bias"], nc_stride, nc_pad)
            x_source = self.adain_resblock1(x_source, s, f"{gp}noise_res.{i}", 7 if i + 1 < nu else 11, [1, 3, 5])
            x = self.conv_weighted(x, f"{gp}ups.{i}", u, (k - u) // 2, 1, transpose=True)
            if i == nu - 1:
                x = F.pad(x, (1, 0))  # "ReflectionPad1d" is a zero left pad (istftnet.py:688-689)
            x = x + x_source
            if inter is not None:
                inter[f"gen_pre_res{i}"] = x.float().numpy().copy()
            xs = None
            for j in range(nk):
                r = self.adain_resblock1(x, s, f"{gp}resblocks.{i * nk + j}", self.rb_k[j], self.rb_d[j])
                xs = r if xs is None else xs + r
            x = xs / nk
            if inter is not None:
                inter[f"gen_stage{i}"] = x.float().numpy().copy()
        x = leaky_relu(x, 0.01)
        x = self.conv_weighted(x, gp + "conv_post", 1, 3, 1)
        if inter is not None:
            inter["conv_post"] = x.float().numpy().copy()
        return self.istft_head(x.float().numpy())

    def decoder(self, asr, f0_curve, n_curve, s, rand_ini, sine_noise, inter=None):
        """Decoder.__call__ (istftnet.py:947-963)."""
        F0 = self.conv_weighted(f0_curve[:, None, :], "decoder.F0_conv", 2, 1, 1)
        N = self.conv_weighted(n_curve[:, None, :], "decoder.N_conv", 2, 1, 1)
        x = torch.cat([asr, F0, N], dim=1)
        x = self.adain_resblk1d(x, s, "decoder.encode")
        if inter is not None:
            inter["dec_encode"] = x.float().numpy().copy()
        asr_res = self.conv_weighted(asr, "decoder.asr_res.0", 1, 0, 1)
        res = True
        for i in range(4):
            if res:
                x = torch.cat([x, asr_res, F0, N], dim=1)
            up = i == 3
            x = self.adain_resblk1d(x, s, f"decoder.decode.{i}", upsample=up)
            if up:
                res = False
        if inter is not None:
            inter["dec_out"] = x.float().numpy().copy()
        return self.generator(x, s, f0_curve, rand_ini, sine_noise, inter)

    # ---- Model.__call__ (kokoro.py:120-170) ---------------------------------------------------------

    def text_stage(self, input_ids, ref_s, speed: float = 1.0) -> np.ndarray:
        """kokoro.py:135-150 only: pred_dur [T] int32 (cheap; used by tests to size buffers)."""
        ids = torch.tensor([[0, *[int(i) for i in input_ids], 0]], dtype=torch.long)
        ref_s = torch.as_tensor(np.asarray(ref_s, np.float32)).reshape(1, 256).to(self.dtype)
        d_en = self.linear(self.albert(ids), "bert_encoder").transpose(1, 2)
        d = self.duration_encoder(d_en, ref_s[:, 128:])
        x = self.lstm(d, "predictor.lstm")
        duration = torch.sigmoid(self.linear(x, "predictor.duration_proj.linear_layer")).sum(-1) / speed
        return torch.clamp(torch.round(duration), min=1).to(torch.int32)[0].numpy()

    def forward(
        self,
        input_ids,
        ref_s,
        speed: float = 1.0,
        forced_dur=None,
        rand_ini=None,
        sine_noise=None,
        return_inter: bool = False,
        f0n_override=None,
    ):
        """input_ids: python list of token ids WITHOUT the BOS/EOS zeros (kokoro.py:135 adds them).
        ref_s [1, 256].  Returns (audio [600F] float32, pred_dur [T] int32[, inter])."""
        inter = {} if return_inter else None
        dt = self.dtype
        ids = torch.tensor([[0, *[int(i) for i in input_ids], 0]], dtype=torch.long)
        T = ids.shape[1]
        assert T <= int(self.cfg["plbert"]["max_position_embeddings"])  # kokoro.py:131-134
        ref_s = torch.as_tensor(np.asarray(ref_s, np.float32)).reshape(1, 256).to(dt)
        bert_dur = self.albert(ids)
        d_en = self.linear(bert_dur, "bert_encoder").transpose(1, 2)  # [1,512,T]
        s = ref_s[:, 128:]
        d = self.duration_encoder(d_en, s)  # [1,T,640]
        x = self.lstm(d, "predictor.lstm")
        dur_logits = self.linear(x, "predictor.duration_proj.linear_layer")
        duration = torch.sigmoid(dur_logits).sum(-1) / speed
        pred_dur = torch.clamp(torch.round(duration), min=1).to(torch.int32)[0]  # torch.round = half-to-even
        if inter is not None:
            inter["bert_dur"] = bert_dur.float().numpy().copy()
            inter["d"] = d.float().numpy().copy()
            inter["duration"] = duration.float().numpy().copy()
        use_dur = pred_dur if forced_dur is None else torch.as_tensor(np.asarray(forced_dur, np.int32))
        idx = torch.repeat_interleave(torch.arange(T), use_dur.to(torch.long))
        Fr = idx.shape[0]
        aln = torch.zeros(T, Fr, dtype=dt)
        aln[idx, torch.arange(Fr)] = 1
        en = d.transpose(1, 2) @ aln[None]  # [1,640,F]
        F0_pred, N_pred = self.f0n_train(en, s)
        if f0n_override is not None:  # tests: condition the vocoder on given F0 / N curves (see DESIGN.md "conditioning")
            F0_pred = torch.as_tensor(np.asarray(f0n_override[0], np.float32)).reshape(1, -1).to(dt)
            N_pred = torch.as_tensor(np.asarray(f0n_override[1], np.float32)).reshape(1, -1).to(dt)
        t_en = self.text_encoder(ids)
        asr = t_en @ aln[None]
        if inter is not None:
            inter["t_en"] = t_en.float().numpy().copy()
            inter["en"] = en.float().numpy().copy()
            inter["asr"] = asr.float().numpy().copy()
            inter["F0_pred"] = F0_pred.float().numpy().copy()
            inter["N_pred"] = N_pred.float().numpy().copy()
        audio = self.decoder(asr, F0_pred, N_pred, ref_s[:, :128], rand_ini, sine_noise, inter)[0]  # [1, 600F]
        out = (np.asarray(audio[0], np.float32), pred_dur.numpy().astype(np.int32))
        return out + (inter,) if return_inter else out
