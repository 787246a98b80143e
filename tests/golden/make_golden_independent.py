"""Fixtures that pin the CPU oracles (oracle/*.py) against INDEPENDENT implementations importable in the build container
("next" #1).  Run here only:  python tests/golden/make_golden_independent.py
Nothing from `transformers` travels: the outputs are small .npz files under tests/golden/independent/, read by
tests/test_oracle_independent.py (CPU, `-m "not gpu"`), which rebuilds the same seeded weights through mlx-audio_amd/params.py
(their SHA-256 is stored in the fixture) and runs the ORACLE on the stored inputs.  This script never imports an oracle.

What is the independent implementation of what (and which reference lines the oracle block restates):
  albert        transformers.AlbertModel (hidden_act="gelu": the reference's nn.GELU() is the exact erf form, modules.py:538; HF's default for
                Albert is gelu_new)                        <-> KokoroOracle.albert          (modules.py:438-649)
  lstm          torch.nn.LSTM(bidirectional=True), weights through params.to_torch_layout (the inverse of `sanitize`)
                                                           <-> KokoroOracle.lstm            (modules.py:93-285)
  stft / istft  torch.stft (symmetric Hann, reflect pad, center) and torch.istft (periodic Hann).  torch.istft divides by sum(w^2), the
                reference by sum(w) (utils.py:143-150): the fixture holds torch's output AND the sum(w^2)/sum(w) envelope built with
                torch.nn.functional.fold                    <-> oracle stft / istft / istft_head (utils.py:52-158, istftnet.py:497-523)
  convs, norms  torch.nn.utils.parametrizations.weight_norm on Conv1d / ConvTranspose1d (no +1e-7 on the norm: istftnet.py:88 adds it, the
                difference is 1e-7 relative), torch.nn.InstanceNorm1d, torch.nn.LayerNorm, F.interpolate(nearest)
                                                           <-> conv_weighted / instance_norm / adain / ada_layer_norm / interpolate
  llama         transformers.LlamaModel, rope llama3 (factor 32, low 1, high 4, 8192), prompt block + two cached steps; q/k rows re-ordered from
                the reference's interleaved-pair RoPE (attention.py:96-110) to HF's half-split form
                                                           <-> csm_oracle.LlamaStack        (attention.py:10-195, sesame.py:296-299; mlx_lm)
  csm           transformers.CsmForConditionalGeneration: [text | audio frames] prompt through the backbone with its KV cache, lm_head,
                the depth decoder greedy over the code books   <-> CsmOracle.generate_frame  (sesame.py:349-415)
  mimi          transformers.MimiModel sub-modules: quantizer decode / encode, upsample / downsample, SEANet encoder / decoder, both
                transformers (HF applies a causal sliding-window mask, the reference's non-streaming call passes none, transformer.py:171:
                the layers are therefore run through HF's own layer modules with attention_mask=None)
                                                           <-> MimiOracle                   (codec/models/mimi/**)
"""
import hashlib
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "independent")
sys.path[:0] = [ROOT]

import mlx_audio_amd.params as P  # noqa: E402  (weights + configs only; no oracle import in this file)

torch.manual_seed(0)
torch.set_grad_enabled(False)


def wdigest(w: dict) -> str:
    h = hashlib.sha256()
    for k in sorted(w):
        a = np.ascontiguousarray(np.asarray(w[k], np.float32))
        h.update(k.encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def save(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    p = os.path.join(OUT, name + ".npz")
    np.savez_compressed(p, **kw)
    print(f"{name}: {os.path.getsize(p) / 1024:.0f} KiB  " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in kw.items() if np.ndim(v) > 0))


def T(a):
    return torch.as_tensor(np.asarray(a, np.float32))


# --------------------------------------------------------------------------------------------------------------------
# Kokoro blocks
# --------------------------------------------------------------------------------------------------------------------
def albert_cfg():
    cfg = P.tiny_config()
    cfg["plbert"] = dict(hidden_size=96, num_attention_heads=4, intermediate_size=160, max_position_embeddings=64, num_hidden_layers=5, dropout=0.1)
    return cfg


def make_albert():
    from transformers import AlbertConfig, AlbertModel

    cfg = albert_cfg()
    w = P.synth_checkpoint(cfg, 11)
    pb = cfg["plbert"]
    hf = AlbertModel(AlbertConfig(vocab_size=cfg["n_token"], embedding_size=128, hidden_size=pb["hidden_size"], num_hidden_layers=pb["num_hidden_layers"],
                                  num_attention_heads=pb["num_attention_heads"], intermediate_size=pb["intermediate_size"], hidden_act="gelu",
                                  hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, max_position_embeddings=pb["max_position_embeddings"],
                                  type_vocab_size=2, layer_norm_eps=1e-12), add_pooling_layer=False).eval()
    sd = hf.state_dict()
    mine = {k[len("bert."):]: v for k, v in w.items() if k.startswith("bert.") and not k.startswith("bert.pooler")}
    missing = [k for k in sd if k not in mine and "position_ids" not in k]
    assert not missing, missing
    hf.load_state_dict({k: T(mine[k]) for k in sd if k in mine}, strict=False)
    rng = np.random.default_rng(5)
    ids = rng.integers(0, cfg["n_token"], (1, 23))
    out = hf(input_ids=torch.as_tensor(ids), attention_mask=torch.ones(1, 23, dtype=torch.long)).last_hidden_state.numpy()
    save("albert", ids=ids, out=out, seed=np.int64(11), wsha=np.array(wdigest(w)))


def make_lstm():
    cfg = P.tiny_config()
    w = P.synth_checkpoint(cfg, 12)
    wt = P.to_torch_layout(w)
    cases = {}
    for name in ("text_encoder.lstm", "predictor.lstm", "predictor.shared"):
        Wih = wt[name + ".weight_ih_l0"]
        H, I = Wih.shape[0] // 4, Wih.shape[1]
        m = torch.nn.LSTM(I, H, batch_first=True, bidirectional=True).eval()
        m.load_state_dict({k: T(wt[f"{name}.{k}"]) for k in m.state_dict()})
        x = torch.randn(2, 37, I)
        y, _ = m(x)
        cases[name.replace(".", "__") + "__x"] = x.numpy()
        cases[name.replace(".", "__") + "__y"] = y.numpy()
    save("lstm", seed=np.int64(12), wsha=np.array(wdigest(w)), **cases)


def make_stft():
    n_fft, hop = 20, 5
    x = torch.randn(3, 600 * 4)
    k = torch.arange(n_fft, dtype=torch.float64)
    w_sym = (0.5 * (1 - torch.cos(2 * torch.pi * k / (n_fft - 1)))).float()  # symmetric Hann, the reference's stft window (utils.py:10-14,73)
    w_per = torch.hann_window(n_fft, periodic=True)  # the reference's istft window hanning(n+1)[:-1] (utils.py:121)
    X = torch.stft(x, n_fft, hop, n_fft, window=w_sym, center=True, pad_mode="reflect", return_complex=True)  # [3, 11, frames]
    # inverse: a random 22-channel head input, the reference's exp / sin parametrisation done in float64 torch here
    frames = 4 * 120 + 1
    head_in = 0.5 * torch.randn(2, 22, frames)
    mag, ph = torch.exp(head_in[:, :11].double()), torch.sin(head_in[:, 11:].double())
    spec = torch.complex(mag * torch.cos(ph), mag * torch.sin(ph)).to(torch.complex64)
    y_torch = torch.istft(spec, n_fft, hop, n_fft, window=w_per, center=True)  # = OLA(w * frame) / OLA(w^2), trimmed by n_fft // 2
    full = (frames - 1) * hop + n_fft
    ola = lambda v: F.fold(v.reshape(1, n_fft, 1).expand(1, n_fft, frames).contiguous(), (1, full), (1, n_fft), stride=(1, hop)).reshape(full)
    env = (ola(w_per * w_per) / ola(w_per))[n_fft // 2: full - n_fft // 2]  # sum(w^2) / sum(w): reference = torch * env
    save("stft", x=x.numpy(), X_re=X.real.numpy(), X_im=X.imag.numpy(), head_in=head_in.numpy(), spec_re=spec.real.numpy(), spec_im=spec.imag.numpy(),
         y_torch=y_torch.numpy(), env=env.numpy())


def make_primitives():
    """weight-normed conv / transposed conv / depth-wise transposed conv, instance norm + AdaIN, AdaLayerNorm, nearest interpolation."""
    from torch.nn.utils.parametrizations import weight_norm

    out = {}
    rng = np.random.default_rng(21)

    def wn_conv(tag, cin, cout, k, stride, pad, dil, groups=1, transpose=False, L=41):
        if transpose:
            m = torch.nn.ConvTranspose1d(cin, cout, k, stride, pad, groups=groups, dilation=dil)
        else:
            m = torch.nn.Conv1d(cin, cout, k, stride, pad, dilation=dil, groups=groups)
        m = weight_norm(m, dim=0).eval()
        g = (1.0 + 0.3 * rng.standard_normal(m.parametrizations.weight.original0.shape)).astype(np.float32)
        v = (rng.standard_normal(m.parametrizations.weight.original1.shape) / np.sqrt(k * max(1, cin // groups))).astype(np.float32)
        b = (0.1 * rng.standard_normal(m.bias.shape)).astype(np.float32)
        m.parametrizations.weight.original0.copy_(T(g))
        m.parametrizations.weight.original1.copy_(T(v))
        m.bias.copy_(T(b))
        x = torch.randn(2, cin, L)
        out.update({f"{tag}__g": g, f"{tag}__v_torch": v, f"{tag}__b": b, f"{tag}__x": x.numpy(), f"{tag}__y": m(x).numpy()})

    wn_conv("conv_k3", 24, 16, 3, 1, 1, 1)
    wn_conv("conv_k7_d3", 16, 16, 7, 1, 9, 3)
    wn_conv("conv_k11_d5", 8, 8, 11, 1, 25, 5, L=97)
    wn_conv("conv_s2", 1, 1, 3, 2, 1, 1, L=40)  # Decoder.F0_conv / N_conv (istftnet.py:923-938)
    wn_conv("ups_k20_s10", 12, 6, 20, 10, 5, 1, transpose=True, L=13)  # Generator.ups[0] (istftnet.py:737-748)
    wn_conv("ups_k12_s6", 6, 4, 12, 6, 3, 1, transpose=True, L=29)
    wn_conv("pool_dw", 10, 10, 3, 2, 1, 1, groups=10, transpose=True, L=17)  # UpSample1d pool, depth-wise (istftnet.py:853-861)
    x = torch.randn(2, 12, 50)
    out["in__x"], out["in__y"] = x.numpy(), torch.nn.InstanceNorm1d(12, eps=1e-5, affine=False)(x).numpy()
    fc = torch.nn.Linear(128, 24)
    s = torch.randn(2, 128)
    h = fc(s)[:, :, None]
    out.update(adain__fc_w=fc.weight.numpy(), adain__fc_b=fc.bias.numpy(), adain__s=s.numpy(),
               adain__y=((1 + h[:, :12]) * torch.nn.InstanceNorm1d(12, eps=1e-5)(x) + h[:, 12:]).numpy())
    xl = torch.randn(1, 9, 20)
    fc2 = torch.nn.Linear(128, 40)
    h2 = fc2(s[:1])
    out.update(adaln__x=xl.numpy(), adaln__fc_w=fc2.weight.numpy(), adaln__fc_b=fc2.bias.numpy(),
               adaln__y=((1 + h2[:, None, :20]) * F.layer_norm(xl, (20,), eps=1e-5) + h2[:, None, 20:]).numpy())
    xi = torch.randn(1, 2, 7)
    out.update(nearest__x=xi.numpy(), nearest__y=F.interpolate(xi, scale_factor=300, mode="nearest").numpy(),
               linear_up__y=F.interpolate(xi, scale_factor=300, mode="linear", align_corners=False).numpy())
    save("primitives", **out)


# --------------------------------------------------------------------------------------------------------------------
# Llama / CSM
# --------------------------------------------------------------------------------------------------------------------
ROPE = {"rope_type": "llama3", "rope_theta": 500000.0, "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
        "original_max_position_embeddings": 8192}


def half_split_rows(wq: np.ndarray, heads: int, hd: int) -> np.ndarray:
    """Rows of a q / k projection from interleaved-pair order (x[2i], x[2i+1] rotate together: attention.py:96-110) to HF's half-split
    order (x[i], x[i + hd/2] rotate together).  The same permutation on q and k leaves every q.k score unchanged."""
    perm = np.concatenate([np.arange(0, hd, 2), np.arange(1, hd, 2)])
    return wq.reshape(heads, hd, -1)[:, perm].reshape(heads * hd, -1)


def csm_small_config():
    rope = dict(rope_theta=500000.0, rope_factor=32.0, rms_eps=1e-5)
    return dict(text_vocab_size=50, audio_vocab_size=19, audio_num_codebooks=4, max_seq_len=2048,
                backbone=dict(num_layers=3, num_heads=4, num_kv_heads=2, head_dim=32, hidden=64, intermediate=96, **rope),
                decoder=dict(num_layers=2, num_heads=2, num_kv_heads=1, head_dim=48, hidden=40, intermediate=56, **rope))


def llama_state(w, prefix, a):
    sd = {}
    for i in range(a["num_layers"]):
        p, q = f"{prefix}.layers.{i}", f"layers.{i}"
        sd[f"{q}.self_attn.q_proj.weight"] = T(half_split_rows(w[f"{p}.self_attn.q_proj.weight"], a["num_heads"], a["head_dim"]))
        sd[f"{q}.self_attn.k_proj.weight"] = T(half_split_rows(w[f"{p}.self_attn.k_proj.weight"], a["num_kv_heads"], a["head_dim"]))
        for nm in ("self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj", "input_layernorm", "post_attention_layernorm"):
            sd[f"{q}.{nm}.weight"] = T(w[f"{p}.{nm}.weight"])
    sd["norm.weight"] = T(w[f"{prefix}.norm.weight"])
    return sd


def make_llama():
    from transformers import LlamaConfig, LlamaModel

    cfg = csm_small_config()
    w = P.csm_synth_checkpoint(cfg, 13)
    a = cfg["backbone"]
    hf = LlamaModel(LlamaConfig(vocab_size=4, hidden_size=a["hidden"], intermediate_size=a["intermediate"], num_hidden_layers=a["num_layers"],
                                num_attention_heads=a["num_heads"], num_key_value_heads=a["num_kv_heads"], head_dim=a["head_dim"],
                                max_position_embeddings=2048, rms_norm_eps=a["rms_eps"], rope_parameters=dict(ROPE), attention_bias=False,
                                mlp_bias=False, attention_dropout=0.0)).eval()
    sd = llama_state(w, "backbone", a)
    sd["embed_tokens.weight"] = hf.state_dict()["embed_tokens.weight"]
    hf.load_state_dict(sd)
    x0, x1, x2 = torch.randn(2, 9, a["hidden"]), torch.randn(2, 1, a["hidden"]), torch.randn(2, 1, a["hidden"])
    o0 = hf(inputs_embeds=x0, use_cache=True)
    o1 = hf(inputs_embeds=x1, past_key_values=o0.past_key_values, use_cache=True)
    o2 = hf(inputs_embeds=x2, past_key_values=o1.past_key_values, use_cache=True)
    # positions far into the scaled region: the llama3 frequency scaling only shows at large positions
    xf = torch.randn(1, 3, a["hidden"])
    pos = torch.tensor([[1500, 1501, 1502]])
    of = hf(inputs_embeds=xf, position_ids=pos, use_cache=False)
    save("llama", seed=np.int64(13), wsha=np.array(wdigest(w)), x0=x0.numpy(), x1=x1.numpy(), x2=x2.numpy(), y0=o0.last_hidden_state.numpy(),
         y1=o1.last_hidden_state.numpy(), y2=o2.last_hidden_state.numpy(), xf=xf.numpy(), yf=of.last_hidden_state.numpy(), posf=pos.numpy(),
         inv_freq=hf.rotary_emb.inv_freq.numpy())


def make_csm():
    from transformers import CsmConfig, CsmDepthDecoderConfig, CsmForConditionalGeneration

    cfg = csm_small_config()
    w = P.csm_synth_checkpoint(cfg, 14)
    a, d = cfg["backbone"], cfg["decoder"]
    ncb, V = cfg["audio_num_codebooks"], cfg["audio_vocab_size"]
    dd = CsmDepthDecoderConfig(num_codebooks=ncb, backbone_hidden_size=a["hidden"], vocab_size=V, hidden_size=d["hidden"], intermediate_size=d["intermediate"],
                               num_hidden_layers=d["num_layers"], num_attention_heads=d["num_heads"], num_key_value_heads=d["num_kv_heads"],
                               head_dim=d["head_dim"], max_position_embeddings=33, rope_parameters=dict(ROPE), rms_norm_eps=d["rms_eps"])
    hc = CsmConfig(num_codebooks=ncb, vocab_size=V, text_vocab_size=cfg["text_vocab_size"], hidden_size=a["hidden"], intermediate_size=a["intermediate"],
                   num_hidden_layers=a["num_layers"], num_attention_heads=a["num_heads"], num_key_value_heads=a["num_kv_heads"], head_dim=a["head_dim"],
                   max_position_embeddings=2048, rope_parameters=dict(ROPE), rms_norm_eps=a["rms_eps"], depth_decoder_config=dd, tie_codebooks_embeddings=True,
                   pad_token_id=None, codebook_pad_token_id=None, bos_token_id=None, audio_token_id=None, audio_eos_token_id=None,
                   codec_config={"model_type": "mimi", "num_hidden_layers": 1, "hidden_size": 32, "num_attention_heads": 2, "num_key_value_heads": 2,
                                 "head_dim": 16, "intermediate_size": 32, "num_filters": 4, "codebook_size": 16, "codebook_dim": 8,
                                 "vector_quantization_hidden_dimension": 8, "num_quantizers": 4, "upsample_groups": 32})
    hf = CsmForConditionalGeneration(hc).eval()
    sd = {}
    sd.update({"backbone_model." + k: v for k, v in llama_state(w, "backbone", a).items()})
    sd.update({"depth_decoder.model." + k: v for k, v in llama_state(w, "decoder", d).items()})
    sd["embed_text_tokens.weight"] = T(w["text_embeddings.weight"])
    sd["backbone_model.embed_tokens.embed_audio_tokens.weight"] = T(w["audio_embeddings.weight"])
    sd["depth_decoder.model.embed_tokens.weight"] = T(w["audio_embeddings.weight"])  # tied (sesame.py:397-399 reads the same table)
    sd["depth_decoder.model.inputs_embeds_projector.weight"] = T(w["projection.weight"])
    sd["depth_decoder.codebooks_head.weight"] = T(w["audio_head"])
    sd["lm_head.weight"] = T(w["codebook0_head.weight"])
    cur = hf.state_dict()
    unset = [k for k in cur if k not in sd and not k.startswith("codec_model.")]
    assert not unset, unset
    for k in cur:
        if k.startswith("codec_model."):
            sd[k] = cur[k]
    hf.load_state_dict(sd)

    rng = np.random.default_rng(6)
    B, S_text, S_audio = 2, 5, 4
    text = rng.integers(0, cfg["text_vocab_size"], (B, S_text))
    audio = rng.integers(0, V, (B, S_audio, ncb))
    o_text = hf(input_ids=torch.as_tensor(text), use_cache=True)  # text positions: embed_text_tokens
    o_aud = hf(input_ids=torch.as_tensor(audio), past_key_values=o_text.past_key_values, use_cache=True, output_hidden_states=True)
    c0_logits = o_aud.logits[:, -1]
    # the final-normed state the lm_head saw: recompute through the backbone module (documented output [0])
    bo = hf.backbone_model(input_ids=torch.as_tensor(audio), past_key_values=hf(input_ids=torch.as_tensor(text), use_cache=True).past_key_values, use_cache=True)
    last_h = bo.last_hidden_state[:, -1]
    assert torch.allclose(hf.lm_head(last_h), c0_logits, atol=1e-6)
    codes = [c0_logits.argmax(-1)]
    ci_logits = []
    past = None
    for i in range(1, ncb):  # greedy over the code books with the depth decoder's own KV cache (fresh per frame, sesame.py:374)
        ids = torch.cat([torch.zeros(B, 1, dtype=torch.long), codes[0][:, None]], 1) if i == 1 else codes[-1][:, None]
        o = hf.depth_decoder(input_ids=ids, backbone_last_hidden_state=last_h if i == 1 else None, past_key_values=past, use_cache=True, logits_to_keep=1)
        past = o.past_key_values
        lg = o.logits[:, -1]
        ci_logits.append(lg.numpy())
        codes.append(lg.argmax(-1))
    save("csm", seed=np.int64(14), wsha=np.array(wdigest(w)), text=text, audio=audio, c0_logits=c0_logits.numpy(), last_h=last_h.numpy(),
         ci_logits=np.stack(ci_logits, 1), codes=torch.stack(codes, 1).numpy(), text_logits_last=o_text.logits[:, -1].numpy())


# --------------------------------------------------------------------------------------------------------------------
# Mimi
# --------------------------------------------------------------------------------------------------------------------
def mimi_hf_state(w: dict, cfg: dict, hf_keys) -> dict:
    """MLX-side names (what the reference's load_pytorch_weights produces, mimi.py:184-249) -> transformers.MimiModel names / layouts."""
    sd = {}
    oik = lambda a: T(np.transpose(a, (0, 2, 1)))  # MLX conv [O, K, I] -> torch [O, I, K]
    H, D = cfg["num_heads"], cfg["dim"]
    hd = D // H
    for side in ("encoder", "decoder"):
        pre = f"{side}.init_conv1d.conv.conv"
        if pre + ".weight" not in w:
            continue
        sd[f"{side}.layers.0.conv.weight"], sd[f"{side}.layers.0.conv.bias"] = oik(w[pre + ".weight"]), T(w[pre + ".bias"])
        for l in range(len(cfg["ratios"])):
            p = f"{side}.layers.{l}"
            if side == "decoder":  # [ELU, convtr, resblock] per ratio after the first conv
                up, rb = 3 * l + 2, 3 * l + 3
                sd[f"decoder.layers.{up}.conv.weight"] = T(np.transpose(w[p + ".upsample.convtr.convtr.weight"], (2, 0, 1)))  # [O,K,I] -> [I,O,K]
                sd[f"decoder.layers.{up}.conv.bias"] = T(w[p + ".upsample.convtr.convtr.bias"])
            else:  # [resblock, ELU, strided conv] per ratio
                rb, dn = 3 * l + 1, 3 * l + 3
                sd[f"encoder.layers.{dn}.conv.weight"] = oik(w[p + ".downsample.conv.conv.weight"])
                sd[f"encoder.layers.{dn}.conv.bias"] = T(w[p + ".downsample.conv.conv.bias"])
            for j, hfj in ((0, 1), (1, 3)):
                sd[f"{side}.layers.{rb}.block.{hfj}.conv.weight"] = oik(w[f"{p}.residuals.0.block.{j}.conv.conv.weight"])
                sd[f"{side}.layers.{rb}.block.{hfj}.conv.bias"] = T(w[f"{p}.residuals.0.block.{j}.conv.conv.bias"])
        last = 3 * len(cfg["ratios"]) + 2
        sd[f"{side}.layers.{last}.conv.weight"] = oik(w[f"{side}.final_conv1d.conv.conv.weight"])
        sd[f"{side}.layers.{last}.conv.bias"] = T(w[f"{side}.final_conv1d.conv.conv.bias"])
    for which, hfq, n in (("rvq_first", "semantic", 1), ("rvq_rest", "acoustic", cfg["nq"] - 1)):
        q = f"quantizer.{hfq}_residual_vector_quantizer"
        for i in range(n):
            sd[f"{q}.layers.{i}.codebook.embed_sum"] = T(w[f"quantizer.{which}.vq.layers.{i}.codebook.embedding_sum"])
            sd[f"{q}.layers.{i}.codebook.cluster_usage"] = T(w[f"quantizer.{which}.vq.layers.{i}.codebook.cluster_usage"])
            sd[f"{q}.layers.{i}.codebook.initialized"] = torch.ones(1)
        for io in ("input_proj", "output_proj"):
            if f"quantizer.{which}.{io}.weight" in w:
                sd[f"{q}.{io}.weight"] = oik(w[f"quantizer.{which}.{io}.weight"])
    sd["upsample.conv.weight"] = T(np.transpose(w["upsample.convtr.convtr.convtr.weight"][0], (1, 0))[:, None, :])  # [1, K, C] -> [C, 1, K]
    if "downsample.conv.conv.conv.weight" in w:
        sd["downsample.conv.weight"] = oik(w["downsample.conv.conv.conv.weight"])
    for tr in ("encoder_transformer", "decoder_transformer"):
        for i in range(cfg["num_layers"]):
            p, q = f"{tr}.transformer.layers.{i}", f"{tr}.layers.{i}"
            if p + ".self_attn.in_proj.weight" not in w:
                continue
            ip = w[p + ".self_attn.in_proj.weight"]
            sd[f"{q}.self_attn.q_proj.weight"] = T(half_split_rows(ip[:D], H, hd))  # nn.RoPE(traditional=True) pairs -> HF half-split pairs
            sd[f"{q}.self_attn.k_proj.weight"] = T(half_split_rows(ip[D: 2 * D], H, hd))
            sd[f"{q}.self_attn.v_proj.weight"] = T(ip[2 * D:])
            sd[f"{q}.self_attn.o_proj.weight"] = T(w[p + ".self_attn.out_proj.weight"])
            sd[f"{q}.mlp.fc1.weight"], sd[f"{q}.mlp.fc2.weight"] = T(w[p + ".gating.linear1.weight"]), T(w[p + ".gating.linear2.weight"])
            for a, b in (("norm1", "input_layernorm"), ("norm2", "post_attention_layernorm")):
                sd[f"{q}.{b}.weight"], sd[f"{q}.{b}.bias"] = T(w[f"{p}.{a}.weight"]), T(w[f"{p}.{a}.bias"])
            sd[f"{q}.self_attn_layer_scale.scale"], sd[f"{q}.mlp_layer_scale.scale"] = T(w[p + ".layer_scale_1.scale"]), T(w[p + ".layer_scale_2.scale"])
    missing = [k for k in hf_keys if k not in sd]
    assert not missing, missing
    return sd


def make_mimi():
    from transformers import MimiConfig, MimiModel

    cfg = P.mimi_tiny_config()
    w = P.mimi_synth_checkpoint(cfg, 15, encode=True)
    # hidden_act: the reference's MlpNoGating applies nn.gelu_approx (tanh form, transformer.py:132); HF's Mimi default is the exact erf GELU
    hc = MimiConfig(hidden_size=cfg["dim"], num_filters=cfg["nfilters"], upsampling_ratios=list(cfg["ratios"]), kernel_size=cfg["ksize"],
                    last_kernel_size=cfg["last_ksize"], residual_kernel_size=cfg["residual_ksize"], compress=cfg["compress"], num_residual_layers=1,
                    codebook_size=cfg["bins"], codebook_dim=cfg["qdim"], vector_quantization_hidden_dimension=cfg["qdim"], num_quantizers=cfg["nq"],
                    num_semantic_quantizers=1, num_hidden_layers=cfg["num_layers"], num_attention_heads=cfg["num_heads"],
                    num_key_value_heads=cfg["num_heads"], head_dim=cfg["dim"] // cfg["num_heads"], intermediate_size=cfg["dim_feedforward"],
                    hidden_act="gelu_pytorch_tanh", upsample_groups=cfg["dim"], rope_theta=float(cfg["rope_base"]), sliding_window=250,
                    use_causal_conv=True, pad_mode="constant", use_conv_shortcut=False, trim_right_ratio=1.0, norm_eps=1e-5, attn_implementation="eager")
    hf = MimiModel(hc).eval()
    hf.load_state_dict(mimi_hf_state(w, cfg, hf.state_dict().keys()))
    rng = np.random.default_rng(8)
    B, Nf = 2, 7
    codes = rng.integers(0, cfg["bins"], (B, cfg["nq"], Nf))
    codes[0, :, 0] = 0  # entry 0 has cluster_usage 0: the 1e-5 floor (quantization.py:25-28)

    def layers_no_mask(tr, x):  # the reference's non-streaming call passes mask=None (transformer.py:171): HF's layer modules, no causal mask
        pos = torch.arange(x.shape[1])[None]
        pe = tr.rotary_emb(x, pos)
        for layer in tr.layers:
            x = layer(x, attention_mask=None, position_embeddings=pe)[0]
        return x

    out = dict(seed=np.int64(15), wsha=np.array(wdigest(w)), codes=codes)
    q = hf.quantizer.decode(torch.as_tensor(codes))
    up = hf.upsample(q)
    trd = layers_no_mask(hf.decoder_transformer, up.transpose(1, 2)).transpose(1, 2)
    trd_causal = hf.decoder_transformer(up.transpose(1, 2), return_dict=True).last_hidden_state.transpose(1, 2)  # HF's own forward: causal
    pcm = hf.decoder(trd)
    out.update(quantized=q.numpy(), upsampled=up.numpy(), dec_tr_nomask=trd.numpy(), dec_tr_causal=trd_causal.numpy(), pcm_from_nomask=pcm.numpy(),
               hf_decode_causal=hf.decode(torch.as_tensor(codes)).audio_values.numpy())
    wav = (0.3 * torch.randn(B, 1, 1920 * 3 + 777))
    e = hf.encoder(wav)
    tre = layers_no_mask(hf.encoder_transformer, e.transpose(1, 2)).transpose(1, 2)
    dn = hf.downsample(tre)
    ecodes = hf.quantizer.encode(dn).transpose(0, 1)
    out.update(wav=wav.numpy(), seanet_enc=e.numpy(), enc_tr_nomask=tre.numpy(), downsampled=dn.numpy(), enc_codes=ecodes.numpy())
    save("mimi", **out)


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, fn in list(globals().items()):
        if name.startswith("make_") and callable(fn) and (not only or name[5:] in only):
            fn()
