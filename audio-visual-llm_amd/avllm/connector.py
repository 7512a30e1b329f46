"""Modality connectors (reference: src/clip_whisper/models/modality_connector.py:6-110,383-402).
`simple` (one nn.Linear, xavier-uniform W, zero bias) is the hot path's default.  `deep` is here too because the reference's factory
maps EVERY unknown name to it (`--connector_type qformer|perceiver|cross_modal|...` silently means `deep`, :394-396).  `conv`,
`attention` and `adaptive` (:112-380) stay out of scope and are refused by name."""
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


ModalityConnector = SimpleModalityConnector


def create_modality_connector(connector_type, input_dim, output_dim, device="cuda", dtype=torch.float32, **kwargs):
    """modality_connector.py:383-399."""
    import logging
    if connector_type == "simple":
        return SimpleModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
    if connector_type in ("conv", "attention", "adaptive"):
        raise NotImplementedError(f"connector_type='{connector_type}' is out of scope of the MI355X hot path (SURVEY.md §8f N4); use 'simple' or 'deep'")
    if connector_type != "deep":
        logging.warning(f"Unknown connector type '{connector_type}', using 'deep' instead")            # the reference's fallback, :394-396
    return DeepModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
