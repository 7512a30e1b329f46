#!/usr/bin/env python3
"""Llama attention forward + backward at the bench shape (T = 256, 32 heads x 128, causal) as a function of the batch: does the backward scale with
the number of workgroups (throughput-bound) or stay flat (latency-bound per workgroup)?  Times avllm_attention_fwd and avllm_attention_bwd (dq + dk/dv)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops

T, H, hd = 256, 32, 128
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for B in (1, 2, 4, 8, 16, 32, 64):
    qkv = torch.randn(B * T, 3 * H * hd, device="cuda", dtype=torch.bfloat16)
    dout = torch.randn(B * T, H * hd, device="cuda", dtype=torch.bfloat16)
    o, lse = ops.attention_fwd(qkv, B, T, H, hd, True)
    res = {}
    for name, fn in (("fwd", lambda: ops.attention_fwd(qkv, B, T, H, hd, True)), ("bwd", lambda: ops.attention_bwd(qkv, o, dout, lse, B, T, H, hd, True))):
        for _ in range(3): fn()
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    fl = 4.0 * B * H * T * T * hd / 2
    print(f"B={B:3d} ({B * H * 2:5d} workgroups per kernel): fwd {res['fwd']:7.1f} us ({fl / res['fwd'] / 1e6:6.1f} TF/s)   bwd {res['bwd']:7.1f} us ({2.5 * fl / res['bwd'] / 1e6:6.1f} TF/s)", flush=True)
