#!/usr/bin/env python3
"""Times the normalisation kernels on the bench shapes (bf16)."""
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
    return e0.elapsed_time(e1) / n


dev = "cuda"
for rows, d, what in ((394000, 768, "clip layernorm"), (24000, 768, "whisper layernorm"), (4096, 4096, "llama rmsnorm")):
    x = torch.randn(rows, d, device=dev, dtype=torch.bfloat16)
    w = torch.randn(d, device=dev, dtype=torch.bfloat16); b = torch.randn(d, device=dev, dtype=torch.bfloat16)
    if "layernorm" in what:
        ms = timed(lambda: ops.layernorm(x, w, b))
        print(f"{what:20s} [{rows},{d}]  fwd {ms*1000:8.1f} us  {2*x.numel()*2/ms/1e9:6.2f} TB/s")
    else:
        ms = timed(lambda: ops.rmsnorm_fwd(x, w, 1e-5))
        y, rstd = ops.rmsnorm_fwd(x, w, 1e-5)
        dy = torch.randn_like(x); dres = torch.randn_like(x)
        ms2 = timed(lambda: ops.rmsnorm_bwd(dy, x, w, rstd, dres))
        print(f"{what:20s} [{rows},{d}]  fwd {ms*1000:8.1f} us  {2*x.numel()*2/ms/1e9:6.2f} TB/s   bwd {ms2*1000:8.1f} us  {4*x.numel()*2/ms2/1e9:6.2f} TB/s")
