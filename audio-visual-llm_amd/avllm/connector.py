"""Modality connectors (reference: src/clip_whisper/models/modality_connector.py:6-402).
`simple` (one nn.Linear, xavier-uniform W, zero bias) is the hot path's default.  `deep` is the factory's fallback for EVERY unknown name
(`--connector_type qformer|perceiver|cross_modal|...` silently means `deep`, :394-396).  `conv`, `attention` and `adaptive` (:111-380) are
built from the same library calls (round 3): avllm_gemm for every Linear / Conv1d (im2col), avllm_layernorm, avllm_groupnorm_tokens,
avllm_act_residual, avllm_attention_fwd.  Parameters live in torch modules of the reference's own layout, so its state dicts load as they are.
One deliberate difference: the nn.MultiheadAttention inside `attention` / `adaptive` carries dropout 0.1, which the reference applies whenever
the model is in train() mode (torch's RNG stream, not reproducible elsewhere); here the connectors always compute the eval-mode function
(no gradient reaches them anyway: SURVEY.md fact 4)."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops


class SimpleModalityConnector(nn.Module):
    def __init__(self, input_dim, output_dim, device="cuda", dtype=torch.float32, max_seq_len=None, **kwargs):
        super().__init__()
        self.input_dim, self.output_dim, self.device, self.dtype = input_dim, output_dim, device, dtype
        self.linear = nn.Linear(input_dim, output_dim)            # parameter storage; compute is ops.gemm
        nn.init.xavier_uniform_(self.linear.weight)
        nn.init.zeros_(self.linear.bias)
        self.linear = self.linear.to(device=device, dtype=dtype)
        for p in self.parameters():
            p.requires_grad_(False)                               # no gradient reaches the connectors

    def forward(self, x):
        if x.dtype != self.dtype:
            x = ops.cast(x, self.dtype)
        shp = x.shape
        if x.dim() == 3 and not x.is_contiguous():
            x = x.contiguous()                                    # row-sliced view (first L frames of each item): one strided copy, then ONE projection
        y = ops.gemm(x.reshape(-1, shp[-1]), self.linear.weight, bias=self.linear.bias)
        return y.view(*shp[:-1], self.output_dim)


class DeepModalityConnector(nn.Module):
    """modality_connector.py:46-110: Linear -> LayerNorm -> GELU -> (num_layers - 2) x [Linear -> LayerNorm -> GELU, + residual] ->
    Linear -> LayerNorm.  Parameters live in the same torch modules under the same names (state_dict compatible); the arithmetic is
    avllm_gemm (bias epilogue) / avllm_layernorm / avllm_act_residual."""

    def __init__(self, input_dim, output_dim, device="cuda", dtype=torch.float32, hidden_dim=None, num_layers=2, **kwargs):
        super().__init__()
        self.input_dim, self.output_dim, self.device, self.dtype = input_dim, output_dim, device, dtype
        hidden_dim = hidden_dim or max(input_dim, output_dim)
        self.input_proj, self.input_norm, self.input_act = nn.Linear(input_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU()
        self.hidden_layers = nn.ModuleList(nn.ModuleList([nn.Linear(hidden_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.GELU()])
                                           for _ in range(num_layers - 2))
        self.output_proj, self.output_norm = nn.Linear(hidden_dim, output_dim), nn.LayerNorm(output_dim)
        for mod in self.modules():
            if isinstance(mod, nn.Linear):
                nn.init.xavier_uniform_(mod.weight)
                nn.init.zeros_(mod.bias)
        self.to(device=device, dtype=dtype)
        for p in self.parameters():
            p.requires_grad_(False)                               # no gradient reaches the connectors

    def forward(self, x):
        from . import lib as L
        if x.dtype != self.dtype:
            x = ops.cast(x, self.dtype)
        shp = x.shape
        h = x.reshape(-1, shp[-1])
        if not h.is_contiguous():
            h = h.contiguous()
        lin = lambda m, t: ops.gemm(t, m.weight, bias=m.bias)
        ln = lambda m, t: ops.layernorm(t, m.weight, m.bias, m.eps)
        h = ops.act_residual(ln(self.input_norm, lin(self.input_proj, h)), L.ACT_GELU)
        for linear, norm, _ in self.hidden_layers:
            h = ops.act_residual(ln(norm, lin(linear, h)), L.ACT_GELU, r=h)
        h = ln(self.output_norm, lin(self.output_proj, h))
        return h.view(*shp[:-1], self.output_dim)


def _freeze(mod, device, dtype):
    mod.to(device=device, dtype=dtype)
    for p in mod.parameters():
        p.requires_grad_(False)                                   # no gradient reaches the connectors


def _xavier(mod):
    for m in mod.modules():
        if isinstance(m, (nn.Linear, nn.Conv1d)):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)


def _conv_w2d(conv):
    """Conv1d weight [out, C, 3] -> GEMM weight [out, kw*C + c], K zero-padded to a multiple of 64 (avllm_gemm's K granularity)."""
    w = conv.weight.permute(0, 2, 1).reshape(conv.weight.shape[0], -1)
    K = (w.shape[1] + 63) // 64 * 64
    if K != w.shape[1]:
        w = torch.cat([w, w.new_zeros(w.shape[0], K - w.shape[1])], 1)
    return w.contiguous()


_lin = lambda m, t: ops.gemm(t.reshape(-1, t.shape[-1]), m.weight, bias=m.bias).view(*t.shape[:-1], m.weight.shape[0])
_ln = lambda m, t: ops.layernorm(t.contiguous(), m.weight, m.bias, m.eps)


class ConvModalityConnector(nn.Module):
    """modality_connector.py:111-172: Conv1d(k3) -> GroupNorm(8) -> GELU -> Conv1d(k3) -> GroupNorm(8) along the sequence, Linear, LayerNorm."""

    def __init__(self, input_dim, output_dim, device="cuda", dtype=torch.float32, kernel_size=3, **kwargs):
        super().__init__()
        if kernel_size != 3:
            raise NotImplementedError("ConvModalityConnector: kernel_size 3 (the reference's default and only configured value)")
        self.input_dim, self.output_dim, self.device, self.dtype = input_dim, output_dim, device, dtype
        self.conv_layers = nn.Sequential(nn.Conv1d(input_dim, output_dim, 3, padding=1), nn.GroupNorm(8, output_dim), nn.GELU(),
                                         nn.Conv1d(output_dim, output_dim, 3, padding=1), nn.GroupNorm(8, output_dim))
        self.final_proj, self.norm = nn.Linear(output_dim, output_dim), nn.LayerNorm(output_dim)
        _xavier(self)
        _freeze(self, device, dtype)

    def forward(self, x):
        from . import lib as L
        if x.dtype != self.dtype:
            x = ops.cast(x, self.dtype)
        c0, g0, _, c1, g1 = self.conv_layers
        h = ops.groupnorm_tokens(ops.conv1d_k3(x, _conv_w2d(c0), c0.bias), g0.weight, g0.bias, 8, g0.eps, act=L.ACT_GELU)
        h = ops.groupnorm_tokens(ops.conv1d_k3(h, _conv_w2d(c1), c1.bias), g1.weight, g1.bias, 8, g1.eps)
        return _ln(self.norm, _lin(self.final_proj, h))


class AttentionModalityConnector(nn.Module):
    """modality_connector.py:174-238: Linear -> LN -> self-attention (8 heads) + residual -> LN -> Linear(4x) GELU Linear + residual -> LN."""

    def __init__(self, input_dim, output_dim, device="cuda", dtype=torch.float32, heads=8, **kwargs):
        super().__init__()
        self.input_dim, self.output_dim, self.device, self.dtype, self.heads = input_dim, output_dim, device, dtype, heads
        self.input_proj, self.norm1 = nn.Linear(input_dim, output_dim), nn.LayerNorm(output_dim)
        self.attention = nn.MultiheadAttention(embed_dim=output_dim, num_heads=heads, dropout=0.1, batch_first=True)
        self.norm2 = nn.LayerNorm(output_dim)
        self.ff = nn.Sequential(nn.Linear(output_dim, output_dim * 4), nn.GELU(), nn.Linear(output_dim * 4, output_dim))
        self.norm3 = nn.LayerNorm(output_dim)
        _xavier(self)
        _freeze(self, device, dtype)

    def forward(self, x):
        from . import lib as L
        if x.dtype != self.dtype:
            x = ops.cast(x, self.dtype)
        a = self.attention
        h = _ln(self.norm1, _lin(self.input_proj, x))
        h = _ln(self.norm2, ops.act_residual(ops.mha_self(h, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, self.heads), L.ACT_NONE, r=h))
        f = _lin(self.ff[2], ops.act_residual(_lin(self.ff[0], h), L.ACT_GELU))
        return _ln(self.norm3, ops.act_residual(f, L.ACT_NONE, r=h))


class _PositionalEncoding(nn.Module):
    def __init__(self, d_model, max_len):
        super().__init__()
        import math
        pe = torch.zeros(max_len, d_model)
        position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)


class _AdaptiveSequencePooling(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.long_adapter = nn.Sequential(nn.Conv1d(dim, dim, 3, stride=2, padding=1), nn.GELU(), nn.Conv1d(dim, dim, 3, stride=2, padding=1))
        self.attn = nn.MultiheadAttention(embed_dim=dim, num_heads=8, dropout=0.1, batch_first=True)
        self.norm = nn.LayerNorm(dim)


class AdaptiveModalityConnector(nn.Module):
    """modality_connector.py:240-380: Linear -> LN -> GELU -> + sinusoidal positions -> [T > 512: Conv1d(k3,s2) GELU Conv1d(k3,s2)] ->
    self-attention (8 heads) + residual -> LN -> Linear -> LN.  Output length = T, or ((T-1)//2)//2 + 1 rows for T > 512."""

    def __init__(self, input_dim, output_dim, device="cuda", dtype=torch.float32, max_seq_len=1536, **kwargs):
        super().__init__()
        self.input_dim, self.output_dim, self.device, self.dtype, self.max_seq_len = input_dim, output_dim, device, dtype, max_seq_len
        mid = (input_dim + output_dim) // 2
        self.input_proj, self.norm1, self.act = nn.Linear(input_dim, mid), nn.LayerNorm(mid), nn.GELU()
        self.pos_encoder = _PositionalEncoding(mid, max_seq_len)
        self.adaptive_pool = _AdaptiveSequencePooling(mid)
        self.output_proj, self.norm2 = nn.Linear(mid, output_dim), nn.LayerNorm(output_dim)
        _xavier(self)
        _freeze(self, device, dtype)

    def forward(self, x):
        from . import lib as L
        if x.dtype != self.dtype:
            x = ops.cast(x, self.dtype)
        B, T, _ = x.shape
        ap, pe = self.adaptive_pool, self.pos_encoder.pe
        if T > pe.shape[0]:
            raise RuntimeError(f"The size of tensor a ({T}) must match the size of tensor b ({pe.shape[0]}) at non-singleton dimension 1")   # torch's own message for x + pe[:T]
        h = ops.act_residual(_ln(self.norm1, _lin(self.input_proj, x)), L.ACT_GELU, r=pe[:T].unsqueeze(0).expand(B, T, -1))
        if T > 512:
            c0, _, c1 = ap.long_adapter
            h = ops.conv1d_k3(ops.act_residual(ops.conv1d_k3(h, _conv_w2d(c0), c0.bias, stride=2), L.ACT_GELU), _conv_w2d(c1), c1.bias, stride=2)
        a = ap.attn
        h = _ln(ap.norm, ops.act_residual(ops.mha_self(h, a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, 8), L.ACT_NONE, r=h))
        return _ln(self.norm2, _lin(self.output_proj, h))


ModalityConnector = SimpleModalityConnector


def create_modality_connector(connector_type, input_dim, output_dim, device="cuda", dtype=torch.float32, **kwargs):
    """modality_connector.py:383-399."""
    import logging
    if connector_type == "simple":
        return SimpleModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
    if connector_type == "conv":
        return ConvModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
    if connector_type == "attention":
        return AttentionModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
    if connector_type == "adaptive":
        return AdaptiveModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
    if connector_type != "deep":
        logging.warning(f"Unknown connector type '{connector_type}', using 'deep' instead")            # the reference's fallback, :394-396
    return DeepModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
