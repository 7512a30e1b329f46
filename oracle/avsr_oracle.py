"""CPU restatement (fp32, plain torch tensor ops) of the src/clip_whisper hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the reference
file:line it restates.  `HF:` = transformers 5.15.0 (the third-party library where the
arithmetic lives).  Parity of this restatement is pinned by
`tests/golden/*.npz`, produced by oracle/make_golden.py from the reference's own
`ClipWhisperModel.encode/forward/generate` driving config-instantiated HF modules.
LoRA (peft) and WER (jiwer) are restated from their published definitions because neither
library is installed here: those two are "parity unpinned" against the real libraries.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- block-scaled fp8 mode (BASELINE config 5)
# The reference has no fp8 mode; the HIP path's `precision="fp8"` runs the frozen-weight projections of the encoders and of the LLM's
# TRAINING forward on the fp8 matrix pipe (oracle/mxfp8.py restates that arithmetic).  `fp8_mode()` switches the projections below to it:
# `_proj_enc` = encoder q/k/v/out/fc1/fc2, `_proj_llm` = decoder q/k/v/o base terms, gate/up/down and lm_head of the training forward.
_FP8 = {"enc": False, "llm": False}


class fp8_mode:
    def __init__(self, encoders=True, llm=True):
        self.new = {"enc": encoders, "llm": llm}

    def __enter__(self):
        self.old = dict(_FP8)
        _FP8.update(self.new)

    def __exit__(self, *a):
        _FP8.update(self.old)


class _LinearFp8(torch.autograd.Function):
    """y = fq(bf16(x)) fq(W)^T; backward dx = dy W: the HIP backward pass multiplies by the bf16 weights, gradients are not quantised."""

    @staticmethod
    def forward(ctx, x, w):
        from . import mxfp8
        ctx.save_for_backward(w)
        return mxfp8.linear_fp8(x, w)

    @staticmethod
    def backward(ctx, dy):
        (w,) = ctx.saved_tensors
        return dy @ w, None


def _proj_enc(x, w):
    return _LinearFp8.apply(x, w) if _FP8["enc"] else x @ w.T


def _proj_llm(x, w):
    return _LinearFp8.apply(x, w) if _FP8["llm"] else x @ w.T


# --------------------------------------------------------------------------- primitives
def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * w + b


def gelu_erf(x):
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


def quick_gelu(x):
    return x * torch.sigmoid(1.702 * x)


def softmax_attention(q, k, v, scale, causal):
    """q,k,v [B,H,T,hd] -> [B,T,H*hd].  HF eager_attention_forward (whisper :215-238, clip :259-277,
    llama eager path): softmax(QK^T*scale + mask) V in fp32."""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if causal:
        T, S = s.shape[-2:]
        mask = torch.ones(T, S, dtype=torch.bool).tril(diagonal=S - T)
        s = s.masked_fill(~mask, float("-inf"))
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v)
    return o.transpose(1, 2).reshape(o.shape[0], o.shape[2], -1)


# --------------------------------------------------------------------------- Whisper encoder
def whisper_encoder(sd, c, mel):
    """HF:models/whisper/modeling_whisper.py:592-646 (layer :379-413, attention :284-357).
    mel [B,80,3000] -> [B,1500,d]."""
    if mel.shape[-1] != 2 * c.n_ctx:
        raise ValueError(f"Whisper expects the mel input features to be of length {2 * c.n_ctx}, but found {mel.shape[-1]}")
    x = gelu_erf(F.conv1d(mel, sd["encoder.conv1.weight"], sd["encoder.conv1.bias"], padding=1))
    x = gelu_erf(F.conv1d(x, sd["encoder.conv2.weight"], sd["encoder.conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd["encoder.embed_positions.weight"]
    B, T, d = x.shape
    H = c.heads
    hd = d // H
    for i in range(c.layers):
        L = f"encoder.layers.{i}."
        h = layer_norm(x, sd[L + "self_attn_layer_norm.weight"], sd[L + "self_attn_layer_norm.bias"])
        q = (_proj_enc(h, sd[L + "self_attn.q_proj.weight"]) + sd[L + "self_attn.q_proj.bias"]) * (hd ** -0.5)
        k = _proj_enc(h, sd[L + "self_attn.k_proj.weight"])
        v = _proj_enc(h, sd[L + "self_attn.v_proj.weight"]) + sd[L + "self_attn.v_proj.bias"]
        sp = lambda t: t.view(B, T, H, hd).transpose(1, 2)
        a = softmax_attention(sp(q), sp(k), sp(v), 1.0, causal=False)
        x = x + (_proj_enc(a, sd[L + "self_attn.out_proj.weight"]) + sd[L + "self_attn.out_proj.bias"])
        h = layer_norm(x, sd[L + "final_layer_norm.weight"], sd[L + "final_layer_norm.bias"])
        h = gelu_erf(_proj_enc(h, sd[L + "fc1.weight"]) + sd[L + "fc1.bias"])
        x = x + (_proj_enc(h, sd[L + "fc2.weight"]) + sd[L + "fc2.bias"])
    return layer_norm(x, sd["encoder.layer_norm.weight"], sd["encoder.layer_norm.bias"])


# --------------------------------------------------------------------------- CLIP vision tower
def clip_vision_cls(sd, c, frames):
    """HF:models/clip/modeling_clip.py:641-656 (embeddings :202-218, layer :362-384, attn :297-335,
    MLP :346-350) followed by the reference's CLS pick WITHOUT post_layernorm
    (clip_whisper_model.py:1138-1142).  frames [N,3,H,W] -> [N,d]."""
    N = frames.shape[0]
    d, H = c.hidden, c.heads
    hd = d // H
    if frames.shape[-1] != c.image or frames.shape[-2] != c.image:
        raise ValueError(f"Input image size ({frames.shape[-2]}*{frames.shape[-1]}) doesn't match model ({c.image}*{c.image}).")
    pe = F.conv2d(frames, sd["embeddings.patch_embedding.weight"], stride=c.patch)  # [N,d,g,g]
    pe = pe.flatten(2).transpose(1, 2)
    cls = sd["embeddings.class_embedding"].expand(N, 1, d)
    x = torch.cat([cls, pe], dim=1) + sd["embeddings.position_embedding.weight"]
    x = layer_norm(x, sd["pre_layrnorm.weight"], sd["pre_layrnorm.bias"], c.eps)
    T = x.shape[1]
    for i in range(c.layers):
        L = f"encoder.layers.{i}."
        h = layer_norm(x, sd[L + "layer_norm1.weight"], sd[L + "layer_norm1.bias"], c.eps)
        # fp8 mode: the last block's out_proj / MLP run on the CLS rows only, in bf16 (csrc/engine.hip encoder_layers)
        pe_ = _proj_enc if i + 1 < c.layers else (lambda xx, ww: xx @ ww.T)
        q = _proj_enc(h, sd[L + "self_attn.q_proj.weight"]) + sd[L + "self_attn.q_proj.bias"]
        k = _proj_enc(h, sd[L + "self_attn.k_proj.weight"]) + sd[L + "self_attn.k_proj.bias"]
        v = _proj_enc(h, sd[L + "self_attn.v_proj.weight"]) + sd[L + "self_attn.v_proj.bias"]
        sp = lambda t: t.view(N, T, H, hd).transpose(1, 2)
        a = softmax_attention(sp(q), sp(k), sp(v), hd ** -0.5, causal=False)
        x = x + (pe_(a, sd[L + "self_attn.out_proj.weight"]) + sd[L + "self_attn.out_proj.bias"])
        h = layer_norm(x, sd[L + "layer_norm2.weight"], sd[L + "layer_norm2.bias"], c.eps)
        h = quick_gelu(pe_(h, sd[L + "mlp.fc1.weight"]) + sd[L + "mlp.fc1.bias"])
        x = x + (pe_(h, sd[L + "mlp.fc2.weight"]) + sd[L + "mlp.fc2.bias"])
    return x[:, 0]


# --------------------------------------------------------------------------- reference glue
def connector(sd, x):
    """SimpleModalityConnector.forward, modality_connector.py:16-20,43-44; DeepModalityConnector._forward_impl :91-108 when the state dict
    carries its keys (input_proj / hidden_layers.N.{0,1} / output_proj / *_norm; nn.GELU() = erf form, LayerNorm eps 1e-5)."""
    if "linear.weight" in sd:
        return x @ sd["linear.weight"].T + sd["linear.bias"]
    h = gelu_erf(layer_norm(x @ sd["input_proj.weight"].T + sd["input_proj.bias"], sd["input_norm.weight"], sd["input_norm.bias"]))
    i = 0
    while f"hidden_layers.{i}.0.weight" in sd:
        p = f"hidden_layers.{i}."
        h = gelu_erf(layer_norm(h @ sd[p + "0.weight"].T + sd[p + "0.bias"], sd[p + "1.weight"], sd[p + "1.bias"])) + h
        i += 1
    return layer_norm(h @ sd["output_proj.weight"].T + sd["output_proj.bias"], sd["output_norm.weight"], sd["output_norm.bias"])


def _conv1d_k3(x, w, b, stride=1):
    """nn.Conv1d(kernel_size=3, padding=1, stride) on token-major x [B,T,C]: y[b,t',o] = b[o] + sum_{kw,c} w[o,c,kw] x[b, stride t' + kw - 1, c]."""
    B, T, C = x.shape
    xp = torch.cat([x.new_zeros(B, 1, C), x, x.new_zeros(B, 1, C)], 1)
    To = (T + 2 - 3) // stride + 1
    idx = torch.arange(To) * stride
    cols = torch.cat([xp[:, idx + kw] for kw in range(3)], -1)                 # [B,To,3C], column kw*C + c
    return cols @ w.permute(0, 2, 1).reshape(w.shape[0], -1).T + b


def _group_norm_tokens(x, w, b, groups, eps=1e-5):
    """nn.GroupNorm(groups, C) applied to [B,C,T], written for token-major x [B,T,C]: statistics over (T, C/groups) per (batch, group)."""
    B, T, C = x.shape
    g = x.view(B, T, groups, C // groups)
    mu = g.mean((1, 3), keepdim=True)
    var = ((g - mu) ** 2).mean((1, 3), keepdim=True)
    return ((g - mu) * torch.rsqrt(var + eps)).view(B, T, C) * w + b


def _mha(x, sd, prefix, heads):
    """nn.MultiheadAttention(batch_first=True) self-attention in eval mode (its dropout 0.1 is off): packed in_proj, scaled dot product,
    out_proj.  The reference runs it in train() mode during training, i.e. WITH dropout on the attention weights (torch's RNG stream: not
    reproducible elsewhere); the restatement and the HIP path are the eval-mode function."""
    B, T, E = x.shape
    hd = E // heads
    qkv = x @ sd[prefix + "in_proj_weight"].T + sd[prefix + "in_proj_bias"]
    q, k, v = (t.view(B, T, heads, hd).transpose(1, 2) for t in qkv.split(E, -1))
    a = softmax_attention(q, k, v, hd ** -0.5, causal=False)                  # [B,T,E]
    return a @ sd[prefix + "out_proj.weight"].T + sd[prefix + "out_proj.bias"]


def connector_conv(sd, x):
    """ConvModalityConnector._forward_impl, modality_connector.py:158-172: Conv1d(k3) -> GroupNorm(8) -> GELU -> Conv1d(k3) -> GroupNorm(8)
    over the sequence, then Linear -> LayerNorm."""
    h = _group_norm_tokens(_conv1d_k3(x, sd["conv_layers.0.weight"], sd["conv_layers.0.bias"]), sd["conv_layers.1.weight"], sd["conv_layers.1.bias"], 8)
    h = _group_norm_tokens(_conv1d_k3(gelu_erf(h), sd["conv_layers.3.weight"], sd["conv_layers.3.bias"]), sd["conv_layers.4.weight"], sd["conv_layers.4.bias"], 8)
    return layer_norm(h @ sd["final_proj.weight"].T + sd["final_proj.bias"], sd["norm.weight"], sd["norm.bias"])


def connector_attention(sd, x, heads=8):
    """AttentionModalityConnector._forward_impl, modality_connector.py:218-238 (eval mode, see _mha)."""
    h = layer_norm(x @ sd["input_proj.weight"].T + sd["input_proj.bias"], sd["norm1.weight"], sd["norm1.bias"])
    h = layer_norm(_mha(h, sd, "attention.", heads) + h, sd["norm2.weight"], sd["norm2.bias"])
    f = gelu_erf(h @ sd["ff.0.weight"].T + sd["ff.0.bias"]) @ sd["ff.2.weight"].T + sd["ff.2.bias"]
    return layer_norm(f + h, sd["norm3.weight"], sd["norm3.bias"])


def connector_adaptive(sd, x):
    """AdaptiveModalityConnector._forward_impl, modality_connector.py:285-302, with PositionalEncoding :304-326 (the `pe` buffer of the state
    dict) and AdaptiveSequencePooling.forward :362-380: sequences longer than 512 go through Conv1d(k3,s2) -> GELU -> Conv1d(k3,s2) (length / 4),
    then self-attention (8 heads, eval mode) + residual + LayerNorm for every length."""
    h = gelu_erf(layer_norm(x @ sd["input_proj.weight"].T + sd["input_proj.bias"], sd["norm1.weight"], sd["norm1.bias"]))
    h = h + sd["pos_encoder.pe"][: h.shape[1]]
    if h.shape[1] > 512:
        h = _conv1d_k3(h, sd["adaptive_pool.long_adapter.0.weight"], sd["adaptive_pool.long_adapter.0.bias"], stride=2)
        h = _conv1d_k3(gelu_erf(h), sd["adaptive_pool.long_adapter.2.weight"], sd["adaptive_pool.long_adapter.2.bias"], stride=2)
    h = layer_norm(_mha(h, sd, "adaptive_pool.attn.", 8) + h, sd["adaptive_pool.norm.weight"], sd["adaptive_pool.norm.bias"])
    return layer_norm(h @ sd["output_proj.weight"].T + sd["output_proj.bias"], sd["norm2.weight"], sd["norm2.bias"])


def pad_or_truncate(x, target_len):
    """clip_whisper_model.py:320-374 (3-D branch)."""
    cur = x.shape[1]
    if cur == target_len:
        return x
    if cur > target_len:
        return x[:, :target_len]
    pad = torch.zeros(x.shape[0], target_len - cur, x.shape[2], dtype=x.dtype)
    return torch.cat([x, pad], dim=1)


def adaptive_projection(x, target_len, training=True):
    """clip_whisper_model.py:621-707.  Longer -> AdaptiveAvgPool1d(target) (window i =
    [floor(i*L/T), ceil((i+1)*L/T))); shorter -> linear interpolation, align_corners=True
    (train: F.interpolate; eval: explicit floor/ceil gather -- same formula)."""
    B, L, D = x.shape
    if L == target_len:
        return x
    if L > target_len:
        out = torch.empty(B, target_len, D, dtype=x.dtype)
        for i in range(target_len):
            s = (i * L) // target_len
            e = -((-(i + 1) * L) // target_len)
            out[:, i] = x[:, s:e].mean(dim=1)
        return out
    if training:
        # F.interpolate(mode='linear', align_corners=True): src = i*(L-1)/(T-1)
        if target_len == 1:
            return x[:, :1]
        scale = (L - 1) / (target_len - 1)
        idx = torch.arange(target_len, dtype=torch.float32) * scale
    else:
        idx = torch.linspace(0, L - 1, target_len)
    lo = idx.floor().long()
    hi = torch.clamp(idx.ceil().long(), max=L - 1) if not training else torch.clamp(lo + 1, max=L - 1)
    a = (idx - lo.float()).view(1, -1, 1)
    return x[:, lo] * (1 - a) + x[:, hi] * a


def adapt_mask(mask, target_len):
    """clip_whisper_model.py:709-736."""
    B, L = mask.shape
    if L == target_len:
        return mask
    if L > target_len:
        return mask[:, :target_len]
    return torch.cat([mask, torch.ones(B, target_len - L, dtype=mask.dtype)], dim=1)


def encode(W, cfg, audio=None, video=None, prompt=None):
    """ClipWhisperModel.encode, clip_whisper_model.py:407-462 with encode_audio :1067-1106,
    encode_video :1108-1146, _embed_prompt :464-487."""
    a = v = None
    if audio is not None:
        nm = cfg.whisper.n_mels              # the reference hard-codes 80 (:1074); 128 only for the whisper-large-v3 family
        if audio.dim() != 3 or audio.shape[1] != nm:
            raise ValueError(f"Audio input should have shape [batch_size, {nm}, time_steps], but got {tuple(audio.shape)}")
        a = connector(W["audio_connector"], whisper_encoder(W["whisper"], cfg.whisper, audio.float()))
    if video is not None:
        if video.dim() != 5 or video.shape[2] != 3:
            raise ValueError(f"Video input should have shape [batch_size, frames, 3, height, width], but got {tuple(video.shape)}")
        B, Fr = video.shape[:2]
        cls = clip_vision_cls(W["clip"], cfg.clip, video.float().reshape(B * Fr, 3, video.shape[3], video.shape[4]))
        v = connector(W["video_connector"], cls.view(B, Fr, -1))
    if a is not None and v is not None:
        L = min(cfg.max_seq_len, max(a.shape[1], v.shape[1]))
        out = cfg.fusion_scale * pad_or_truncate(a, L) + (1 - cfg.fusion_scale) * pad_or_truncate(v, L)
    elif a is not None:
        out = a
    elif v is not None:
        out = v
    else:
        raise ValueError("No valid inputs provided - both audio and video are None")
    if prompt is not None:
        ids = prompt[:, : cfg.max_prompt_len]
        out = torch.cat([W["llama"]["model.embed_tokens.weight"][ids], out], dim=1)
    mask = torch.ones(out.shape[0], out.shape[1], dtype=torch.long)
    return out, mask


# --------------------------------------------------------------------------- Llama + LoRA
def rms_norm(x, w, eps):
    """HF:models/llama/modeling_llama.py:62-67."""
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


def rope_cos_sin(positions, hd, theta, scaling=()):
    """HF:models/llama/modeling_llama.py:70-126 (default rope); `scaling` = (factor, low_freq_factor, high_freq_factor,
    original_max_position_embeddings) applies the "llama3" rule of HF:modeling_rope_utils.py _compute_llama3_parameters (Llama-3.1 / 3.2)."""
    inv = 1.0 / (theta ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    if scaling:
        factor, low, high, octx = scaling
        wavelen = 2 * math.pi / inv
        inv_l = torch.where(wavelen > octx / low, inv / factor, inv)
        smooth = (octx / wavelen - low) / (high - low)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        medium = ~(wavelen < octx / high) & ~(wavelen > octx / low)
        inv = torch.where(medium, smoothed, inv_l)
    fr = positions.float()[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], dim=-1)
    return emb.cos(), emb.sin()


def apply_rope(x, cos, sin):
    """x [B,H,T,hd]; HF:...modeling_llama.py:129-160 (rotate_half convention)."""
    h = x.shape[-1] // 2
    rot = torch.cat([-x[..., h:], x[..., :h]], dim=-1)
    return x * cos + rot * sin


def lora_linear(x, w, lora, key, scale, masks=None):
    """peft lora.Linear: W x + scale * B(A dropout(x)).  `masks[key]` (same shape as x, already scaled by 1/(1-p)) is the
    dropout mask of this module; None = eval / p=0.  Reference wrap: clip_whisper_model.py:961-1005.
    Parity unpinned against real peft (not installed)."""
    y = _proj_llm(x, w)
    if lora is not None and (key + ".lora_A") in lora:
        xl = x * masks[key].view_as(x) if masks is not None and key in masks else x
        y = y + scale * ((xl @ lora[key + ".lora_A"].T) @ lora[key + ".lora_B"].T)
    return y


def llama_hidden(sd, lora, c, lc, x, past=None, pos0=0, masks=None):
    """LlamaModel.forward, HF:models/llama/modeling_llama.py:366-419; layer :284-324; attn :236-281.
    x [B,T,d] inputs_embeds.  `past` = optional list of (k,v) per layer (KV cache), updated in place."""
    B, T, d = x.shape
    H, hd = c.heads, c.head_dim
    cos, sin = rope_cos_sin(torch.arange(pos0, pos0 + T), hd, c.theta, tuple(getattr(c, "rope_scaling", ()) or ()))
    scale = lc.scale if lc is not None else 0.0
    for i in range(c.layers):
        L = f"model.layers.{i}."
        K = f"layers.{i}."
        h = rms_norm(x, sd[L + "input_layernorm.weight"], c.eps)
        q = lora_linear(h, sd[L + "self_attn.q_proj.weight"], lora, K + "q_proj", scale, masks)
        k = lora_linear(h, sd[L + "self_attn.k_proj.weight"], lora, K + "k_proj", scale, masks)
        v = lora_linear(h, sd[L + "self_attn.v_proj.weight"], lora, K + "v_proj", scale, masks)
        Hkv = getattr(c, "kv_heads", 0) or H                 # grouped-query attention: k/v carry Hkv heads (modeling_llama.py:203-212 repeat_kv)
        sp = lambda t, n: t.view(B, T, n, hd).transpose(1, 2)
        q, k, v = apply_rope(sp(q, H), cos, sin), apply_rope(sp(k, Hkv), cos, sin), sp(v, Hkv)
        if past is not None:
            if past[i] is not None:
                k = torch.cat([past[i][0], k], dim=2)
                v = torch.cat([past[i][1], v], dim=2)
            past[i] = (k, v)
        if Hkv != H:
            k, v = k.repeat_interleave(H // Hkv, dim=1), v.repeat_interleave(H // Hkv, dim=1)
        a = softmax_attention(q, k, v, hd ** -0.5, causal=True)
        x = x + lora_linear(a, sd[L + "self_attn.o_proj.weight"], lora, K + "o_proj", scale, masks)
        h = rms_norm(x, sd[L + "post_attention_layernorm.weight"], c.eps)
        g = _proj_llm(h, sd[L + "mlp.gate_proj.weight"])
        u = _proj_llm(h, sd[L + "mlp.up_proj.weight"])
        x = x + _proj_llm(F.silu(g) * u, sd[L + "mlp.down_proj.weight"])
    return rms_norm(x, sd["model.norm.weight"], c.eps)


def causal_lm_loss(logits, labels):
    """HF:loss/loss_utils.py:49-71: labels padded with -100 then shifted by one; mean CE over
    non-ignored positions in fp32."""
    B, T, V = logits.shape
    shift = torch.cat([labels[:, 1:], torch.full((B, 1), -100, dtype=labels.dtype)], dim=1)
    return F.cross_entropy(logits.float().view(-1, V), shift.reshape(-1), ignore_index=-100, reduction="mean")


def prepare_llm_inputs(W, cfg, audio, video, prompt, labels, training):
    """The part of ClipWhisperModel.forward before the LLM call, clip_whisper_model.py:489-598."""
    x, mask = encode(W, cfg, audio, video, prompt)
    if labels is not None:
        labels = labels.clone()
        labels[labels == cfg.pad_token_id] = -100
        if labels.shape[1] != x.shape[1]:
            if training:
                x = adaptive_projection(x, labels.shape[1], training=True)
                mask = adapt_mask(mask, labels.shape[1])
            elif labels.shape[1] > x.shape[1]:
                labels = labels[:, : x.shape[1]]
            else:
                pad = torch.full((labels.shape[0], x.shape[1] - labels.shape[1]), -100, dtype=labels.dtype)
                labels = torch.cat([labels, pad], dim=1)
    return x, mask, labels


def forward(W, cfg, audio=None, video=None, prompt=None, labels=None, training=True, return_loss=True):
    """ClipWhisperModel.forward, clip_whisper_model.py:489-619.  Returns {"loss","logits"}."""
    x, mask, labels = prepare_llm_inputs(W, cfg, audio, video, prompt, labels if return_loss else None, training)
    h = llama_hidden(W["llama"], W.get("lora"), cfg.llama, cfg.lora, x)
    logits = h @ W["llama"]["lm_head.weight"].T
    out = {"logits": logits}
    if return_loss and labels is not None:
        out["loss"] = causal_lm_loss(logits, labels)
    return out


def train_step_grads(W, cfg, audio, video, prompt, labels, masks=None):
    """One forward+backward of the reference's training step (trainer/clip_whisper_trainer.py:433-456)
    with freeze_encoders=True: encoders+connectors run under no_grad (clip_whisper_model.py:1096-1106),
    so only the LoRA tensors receive gradients."""
    with torch.no_grad():
        x, mask, lab = prepare_llm_inputs(W, cfg, audio, video, prompt, labels, training=True)
    lora = {k: v.clone().requires_grad_(True) for k, v in W["lora"].items()}
    h = llama_hidden(W["llama"], lora, cfg.llama, cfg.lora, x, masks=masks)
    logits = _proj_llm(h, W["llama"]["lm_head.weight"])
    loss = causal_lm_loss(logits, lab)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in lora.items()}


def generate(W, cfg, audio=None, video=None, prompt=None, max_new_tokens=100, eos_token_id=None, return_margins=False):
    """ClipWhisperModel.generate, clip_whisper_model.py:1240-1348 -> HF GenerationMixin greedy search
    (do_sample=False) on inputs_embeds with a KV cache; returns new tokens only [B, <=max_new_tokens].
    After a row emits eos it is padded with pad_token_id (HF greedy `unfinished_sequences` rule).
    return_margins: also return the top-1 minus top-2 logit of every step [B, steps] (a reduced-precision implementation can only be
    held to the same token where this margin exceeds its logit error; tests/test_decode_gpu.py)."""
    with torch.no_grad():
        x, mask = encode(W, cfg, audio, video, prompt)
        sd, c = W["llama"], cfg.llama
        past = [None] * c.layers
        h = llama_hidden(sd, W.get("lora"), c, cfg.lora, x, past=past, pos0=0)
        pos = x.shape[1]
        B = x.shape[0]
        unfinished = torch.ones(B, dtype=torch.bool)
        out, margins = [], []
        for _ in range(max_new_tokens):
            logits = h[:, -1] @ sd["lm_head.weight"].T
            nxt = logits.argmax(-1)
            top2 = logits.topk(2, dim=-1).values
            margins.append(top2[:, 0] - top2[:, 1])
            if eos_token_id is not None:
                nxt = torch.where(unfinished, nxt, torch.full_like(nxt, cfg.pad_token_id))
                unfinished = unfinished & (nxt != eos_token_id)
            out.append(nxt)
            if eos_token_id is not None and not unfinished.any():
                break
            h = llama_hidden(sd, W.get("lora"), c, cfg.lora, sd["model.embed_tokens.weight"][nxt][:, None], past=past, pos0=pos)
            pos += 1
        if return_margins:
            return torch.stack(out, dim=1), torch.stack(margins, dim=1)
        return torch.stack(out, dim=1)


# --------------------------------------------------------------------------- optimizer
def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ as used at trainer/clip_whisper_trainer.py:458:
    total L2 norm; coef = max_norm/(norm+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def adamw_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.01):
    """torch.optim.AdamW (non-amsgrad) single-tensor rule; hyper-parameters from
    trainer/clip_whisper_trainer.py:171-232.  `step` is 1-based."""
    p.mul_(1 - lr * wd)
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def cosine_lr(base_lr, step, t_max, eta_min=0.0):
    """CosineAnnealingLR closed form (trainer/clip_whisper_trainer.py:210-230, no-warmup branch)."""
    return eta_min + (base_lr - eta_min) * (1 + math.cos(math.pi * step / t_max)) / 2


# --------------------------------------------------------------------------- WER
def wer(refs, hyps):
    """jiwer.wer(refs, hyps) restated (scripts/clip_whisper/decode.py:30-37, use sites :597,:658):
    corpus-level (S+D+I)/N over whitespace-split words, no normalisation.  Parity unpinned
    against jiwer (not installed; known-answer tests in tests/test_oracle.py)."""
    if isinstance(refs, str):
        refs, hyps = [refs], [hyps]
    errs = n = 0
    for r, h in zip(refs, hyps):
        rw, hw = r.split(), h.split()
        n += len(rw)
        prev = list(range(len(hw) + 1))
        for i in range(1, len(rw) + 1):
            cur = [i] + [0] * len(hw)
            for j in range(1, len(hw) + 1):
                cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (rw[i - 1] != hw[j - 1]))
            prev = cur
        errs += prev[len(hw)]
    return errs / n if n else float("inf")
