"""Modality connectors (reference: src/clip_whisper/models/modality_connector.py:6-44,383-402).
Only the default `simple` connector (one nn.Linear, xavier-uniform W, zero bias) is on the hot path; the other
types of the reference's factory are out of scope and are refused by name."""
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
            out = torch.empty(shp[0], shp[1], self.output_dim, device=x.device, dtype=self.dtype)
            for b in range(shp[0]):                               # row-sliced view (first L frames of each item)
                ops.gemm(x[b], self.linear.weight, out=out[b], bias=self.linear.bias)
            return out
        y = ops.gemm(x.reshape(-1, shp[-1]), self.linear.weight, bias=self.linear.bias)
        return y.view(*shp[:-1], self.output_dim)


ModalityConnector = SimpleModalityConnector


def create_modality_connector(connector_type, input_dim, output_dim, device="cuda", dtype=torch.float32, **kwargs):
    if connector_type != "simple":
        raise NotImplementedError(f"connector_type='{connector_type}' is out of scope; only 'simple' is on the hot path")
    return SimpleModalityConnector(input_dim, output_dim, device, dtype, **kwargs)
