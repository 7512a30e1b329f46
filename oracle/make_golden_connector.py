#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only; the reference does not exist on the GPU box): drives the REFERENCE's own
DeepModalityConnector (src/clip_whisper/models/modality_connector.py:46-110, loaded by file path) and its factory's fallback
(:383-399: an unknown connector name means `deep`), checks the CPU restatement (oracle/avsr_oracle.py `connector`) against it and
writes tests/golden/g9_deep_connector.npz: parameters, input, output.  Pure data; nothing of the reference's text is stored."""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import avsr_oracle as O  # noqa: E402

REF = "/root/reference/src/clip_whisper/models/modality_connector.py"


def main():
    spec = importlib.util.spec_from_file_location("ref_modality_connector", REF)
    mc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mc)
    out = {}
    for tag, layers, name in (("a", 2, "deep"), ("b", 4, "qformer")):          # "qformer" is not in the factory's map -> deep
        torch.manual_seed(11 + layers)
        kw = dict(device="cpu", dtype=torch.float32, num_layers=layers)
        conn = mc.create_modality_connector(name, 64, 128, **kw)
        assert type(conn).__name__ == "DeepModalityConnector"
        for p in conn.parameters():                                            # non-trivial biases / norm weights so every term is exercised
            if p.dim() == 1:
                p.data.add_(0.1 * torch.randn_like(p))
        x = torch.randn(2, 9, 64)
        with torch.no_grad():
            y = conn(x)
        sd = {k: v.detach().clone() for k, v in conn.state_dict().items()}
        mine = O.connector(sd, x)
        assert (mine - y).abs().max() < 2e-6, (mine - y).abs().max()
        for k, v in sd.items():
            out[f"{tag}.sd.{k}"] = v.numpy()
        out[f"{tag}.x"], out[f"{tag}.y"], out[f"{tag}.layers"] = x.numpy(), y.numpy(), np.array(layers)
        print(f"case {tag}: reference DeepModalityConnector ({layers} layers, requested as '{name}') == oracle, max diff {(mine - y).abs().max():.1e}")
    path = os.path.join(ROOT, "tests", "golden", "g9_deep_connector.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    # ---- round 3: conv / attention / adaptive (modality_connector.py:111-380), eval mode (their nn.MultiheadAttention carries dropout 0.1:
    # a train()-mode forward of the reference draws from torch's RNG and is not reproducible outside it)
    out = {}
    cases = (("conv", "conv", 64, 128, 21, {}, O.connector_conv), ("attn", "attention", 64, 128, 19, {}, O.connector_attention),
             ("adapt_short", "adaptive", 128, 128, 40, {"max_seq_len": 1536}, O.connector_adaptive),
             ("adapt_long", "adaptive", 128, 128, 600, {"max_seq_len": 1536}, O.connector_adaptive))      # > 512 tokens: the strided-conv branch
    for tag, name, din, dout, T, kw, fn in cases:
        torch.manual_seed(100 + T)
        conn = mc.create_modality_connector(name, din, dout, device="cpu", dtype=torch.float32, **kw).eval()
        for p_ in conn.parameters():
            if p_.dim() == 1:
                p_.data.add_(0.1 * torch.randn_like(p_))
        x = torch.randn(2, T, din)
        with torch.no_grad():
            y = conn(x)
        sd = {k: v.detach().clone() for k, v in conn.state_dict().items()}
        mine = fn(sd, x)
        assert mine.shape == y.shape and (mine - y).abs().max() < 5e-5, (tag, mine.shape, y.shape, (mine - y).abs().max())
        for k, v in sd.items():
            if k == "pos_encoder.pe":
                v = v[:T]                                            # only the rows a T-token input reads (the buffer has max_seq_len rows)
            out[f"{tag}.sd.{k}"] = v.numpy()
        out[f"{tag}.x"], out[f"{tag}.y"] = x.numpy(), y.numpy()
        print(f"case {tag}: reference {type(conn).__name__} == oracle, max diff {(mine - y).abs().max():.1e}, out {tuple(y.shape)}")
    path = os.path.join(ROOT, "tests", "golden", "g10_connectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
