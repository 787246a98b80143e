"""CPU restatement (PyTorch-CPU fp32) of the reference's CSM-1B frame generator -- TEST INFRASTRUCTURE ONLY (same rule as the other
oracles: only tests/, smoke and cpu_baseline legs may import it).

Follows, as text:
  mlx_audio/tts/models/sesame/sesame.py:276-415   SesameModel: embeddings, masked sum, backbone, codebook0 head, 31 depth-decoder
                                                  steps (projection, decoder, audio_head[i-1]), _embed_audio / _embed_tokens
  mlx_audio/tts/models/sesame/sesame.py:37-48     create_causal_mask / index_causal_mask (causal inside the new block; every cached key visible)
  mlx_audio/tts/models/sesame/attention.py:10-195 Llama3ScaledRoPE (interleaved pairs, cos/sin cache, llama3 frequency scaling), GQA attention
  mlx_lm (NOT in the reference tree, pinned there as a dependency): LlamaModel = per layer x + attn(rms(x)); h + down(silu(gate(rms h)) * up(rms h));
                                                  final RMSNorm; KVCache; make_sampler(temp, top_k)

PARITY WITH MLX IS UNPINNED (mlx / mlx_lm absent, no fixtures for this path).  Sampling: mlx_lm draws with MLX's RNG, which cannot be
reproduced; here and in the HIP path a frame is drawn either greedily (temp = 0 -> argmax, the same rule make_sampler applies) or by inverse CDF
over the top-k set ordered by descending logit with INJECTED uniforms -- the same distribution, a reproducible draw.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


def t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32)


def llama3_theta(dim: int, base: float, scale_factor: float, low: float = 1.0, high: float = 4.0, old_ctx: int = 8192) -> np.ndarray:
    """attention.py:33-82 (rope_init + apply_scaling), float32 like the reference."""
    freqs = (1.0 / (np.float32(base) ** (np.arange(0, dim, 2)[: dim // 2].astype(np.float32) / np.float32(dim)))).astype(np.float32)
    low_w, high_w = old_ctx / low, old_ctx / high
    out = []
    for f in freqs:
        wl = 2 * math.pi / float(f)
        if wl < high_w:
            out.append(float(f))
        elif wl > low_w:
            out.append(float(f) / scale_factor)
        else:
            smooth = (old_ctx / wl - low) / (high - low)
            out.append((1 - smooth) * float(f) / scale_factor + smooth * float(f))
    return np.asarray(out, np.float32)


class LlamaStack:
    """mlx_lm LlamaModel with the reference's Attention swapped in (sesame.py:296-299) and embed_tokens = Identity."""

    def __init__(self, w: dict, prefix: str, a: dict):
        self.w, self.p, self.a = w, prefix, a
        self.theta = t(llama3_theta(a["head_dim"], a["rope_theta"], a["rope_factor"]))
        self.reset()

    def reset(self):
        self.k = [None] * self.a["num_layers"]
        self.v = [None] * self.a["num_layers"]
        self.offset = 0

    def rms(self, x, name):
        v = x.pow(2).mean(-1, keepdim=True)
        return x * torch.rsqrt(v + self.a["rms_eps"]) * t(self.w[name])

    def rope(self, x, offset):  # x [B, S, H, D]
        S = x.shape[1]
        ang = torch.arange(offset, offset + S, dtype=torch.float32)[:, None] * self.theta[None, :]
        c, s = torch.cos(ang)[None, :, None, :], torch.sin(ang)[None, :, None, :]
        x0, x1 = x[..., 0::2], x[..., 1::2]
        out = torch.empty_like(x)
        out[..., 0::2] = x0 * c - x1 * s
        out[..., 1::2] = x1 * c + x0 * s
        return out

    def __call__(self, h: torch.Tensor) -> torch.Tensor:
        a = self.a
        B, S, _ = h.shape
        H, KV, D = a["num_heads"], a["num_kv_heads"], a["head_dim"]
        off = self.offset
        for i in range(a["num_layers"]):
            p = f"{self.p}.layers.{i}"
            x = self.rms(h, p + ".input_layernorm.weight")
            q = (x @ t(self.w[p + ".self_attn.q_proj.weight"]).T).reshape(B, S, H, D)
            k = (x @ t(self.w[p + ".self_attn.k_proj.weight"]).T).reshape(B, S, KV, D)
            v = (x @ t(self.w[p + ".self_attn.v_proj.weight"]).T).reshape(B, S, KV, D)
            q, k = self.rope(q, off), self.rope(k, off)
            self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], 1)
            self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], 1)
            kk = self.k[i].repeat_interleave(H // KV, dim=2)  # explicit K/V broadcast, attention.py:175-188
            vv = self.v[i].repeat_interleave(H // KV, dim=2)
            sc = torch.einsum("bshd,bthd->bhst", q, kk) * D ** -0.5
            klen = self.k[i].shape[1]  # == off + S unless a test moved `offset` (RoPE positions) without filling the cache
            qpos = torch.arange(klen - S, klen)[:, None]
            kpos = torch.arange(0, klen)[None, :]
            sc = sc.masked_fill(~(kpos <= qpos)[None, None], float("-inf"))  # causal in the new block, all cached keys visible
            o = torch.einsum("bhst,bthd->bshd", torch.softmax(sc, -1), vv).reshape(B, S, H * D)
            h = h + o @ t(self.w[p + ".self_attn.o_proj.weight"]).T
            x = self.rms(h, p + ".post_attention_layernorm.weight")
            g = x @ t(self.w[p + ".mlp.gate_proj.weight"]).T
            u = x @ t(self.w[p + ".mlp.up_proj.weight"]).T
            h = h + (F.silu(g) * u) @ t(self.w[p + ".mlp.down_proj.weight"]).T
        self.offset += S
        return self.rms(h, f"{self.p}.norm.weight")


def sample(logits: torch.Tensor, temp: float, top_k: int, u: np.ndarray | None) -> np.ndarray:
    """logits [B, V].  temp == 0 or u is None: argmax (first index on ties).  Else inverse CDF over the top_k logits in descending order
    (ties: lower index first) of softmax(logit / temp), with one injected uniform per row."""
    lg = logits.numpy().astype(np.float32)
    if u is None or temp == 0:
        return lg.argmax(-1).astype(np.int64)
    out = np.zeros(lg.shape[0], np.int64)
    for b in range(lg.shape[0]):
        order = np.lexsort((np.arange(lg.shape[1]), -lg[b]))[:top_k]
        z = (lg[b, order] / np.float32(temp)).astype(np.float32)
        p = np.exp(z - z.max()).astype(np.float32)
        c = np.cumsum(p, dtype=np.float32)
        j = int(np.searchsorted(c, np.float32(u[b]) * c[-1], side="left"))
        out[b] = order[min(j, top_k - 1)]
    return out


class CsmOracle:
    def __init__(self, w: dict, cfg: dict):
        self.w = {k: np.asarray(v, np.float32) for k, v in w.items()}
        self.cfg = cfg
        self.backbone = LlamaStack(self.w, "backbone", cfg["backbone"])
        self.decoder = LlamaStack(self.w, "decoder", cfg["decoder"])

    def reset_caches(self):
        self.backbone.reset()
        self.decoder.reset()

    def embed_audio(self, codebook: int, tokens: np.ndarray) -> torch.Tensor:
        return t(self.w["audio_embeddings.weight"])[torch.as_tensor(tokens + codebook * self.cfg["audio_vocab_size"])]

    def embed_tokens(self, tokens: np.ndarray) -> torch.Tensor:
        """tokens [B, S, n_cb + 1] -> [B, S, n_cb + 1, D] (sesame.py:400-415)."""
        ncb = self.cfg["audio_num_codebooks"]
        text = t(self.w["text_embeddings.weight"])[torch.as_tensor(tokens[:, :, -1])][:, :, None]
        aud = t(self.w["audio_embeddings.weight"])[torch.as_tensor(tokens[:, :, :-1] + np.arange(ncb)[None, None] * self.cfg["audio_vocab_size"])]
        return torch.cat([aud, text], dim=-2)

    def generate_frame(self, tokens: np.ndarray, tokens_mask: np.ndarray, temp: float = 0.0, top_k: int = 50, uniforms: np.ndarray | None = None,
                       trace: dict | None = None) -> np.ndarray:
        """sesame.py:349-395.  tokens [B, S, n_cb+1] int, tokens_mask same shape (0/1); positions continue from the backbone cache.
        uniforms [B, n_cb] or None (greedy).  Returns codes [B, n_cb]."""
        with torch.no_grad():
            ncb = self.cfg["audio_num_codebooks"]
            emb = self.embed_tokens(np.asarray(tokens)) * t(tokens_mask)[..., None]
            h = emb.sum(2)
            h = self.backbone(h)
            last_h = h[:, -1]
            c0_logits = last_h @ t(self.w["codebook0_head.weight"]).T
            c0 = sample(c0_logits, temp, top_k, None if uniforms is None else uniforms[:, 0])
            if trace is not None:
                trace["last_h"] = last_h.numpy().copy()
                trace["c0_logits"] = c0_logits.numpy().copy()
                trace["ci_logits"] = []
            samples = [c0]
            curr = torch.cat([last_h[:, None], self.embed_audio(0, c0)[:, None]], 1)
            self.decoder.reset()  # fresh decoder cache every frame (sesame.py:374)
            proj = t(self.w["projection.weight"])
            for i in range(1, ncb):
                dh = self.decoder(curr @ proj.T)
                lg = dh[:, -1] @ t(self.w["audio_head"][i - 1])
                if trace is not None:
                    trace["ci_logits"].append(lg.numpy().copy())
                ci = sample(lg, temp, top_k, None if uniforms is None else uniforms[:, i])
                samples.append(ci)
                curr = self.embed_audio(i, ci)[:, None]
            return np.stack(samples, 1)
