#!/usr/bin/env python3
"""Times the attention kernels on the bench shapes (bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000


dev = "cuda"
for name, B, T, H, hd, causal, bwd in (("llama  B16 T256 H32 hd128 causal", 16, 256, 32, 128, True, True), ("clip   2000x197 H12 hd64", 2000, 197, 12, 64, False, False),
                                       ("whisper B16 T1500 H12 hd64", 16, 1500, 12, 64, False, False), ("vit-l  3000x257 H16 hd64", 3000, 257, 16, 64, False, False)):
    qkv = torch.randn(B * T, 3 * H * hd, device=dev, dtype=torch.bfloat16)
    us = timed(lambda: ops.attention_fwd(qkv, B, T, H, hd, causal))
    fl = 4.0 * B * H * T * T * hd * (0.5 if causal else 1.0)
    line = f"{name:36s} fwd {us:8.1f} us  {fl / us / 1e6:7.1f} TF/s  {(qkv.numel() * 2 + B * T * H * hd * 2) / us / 1e6:5.2f} TB/s"
    if bwd:
        o, lse = ops.attention_fwd(qkv, B, T, H, hd, causal)
        do = torch.randn_like(o)
        ub = timed(lambda: ops.attention_bwd(qkv, o, do, lse, B, T, H, hd, causal))
        line += f"   bwd {ub:8.1f} us  {2.5 * fl / ub / 1e6:7.1f} TF/s"
    print(line)
