#!/usr/bin/env python3
"""Per-tile fixed cost of the 256x256 GEMM kernels: time the CLIP qkv shape (M=394000, N=2304) at K = 128..1536 and fit
t = tiles/256 * (fixed + K/64 * per_kstep).  AVLLM_GEMM_VARIANT picks the kernel (5 = 16 waves, 7 = 4 waves)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops

M, N = int(os.environ.get("M", 394000)), int(os.environ.get("N", 2304))
rounds = ((M + 255) // 256) * ((N + 255) // 256) / 256.0
pts = []
for K in (128, 256, 512, 768, 1536):
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16); B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(2): ops.gemm(A, B, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6): ops.gemm(A, B, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 6 * 1000
    pts.append((K // 64, us / rounds))
    print(f"K={K:5d}  {us:9.1f} us  {us / rounds:7.2f} us per tile round  {2*M*N*K/us/1e6:8.1f} TF/s", flush=True)
n = len(pts); sx = sum(p[0] for p in pts); sy = sum(p[1] for p in pts); sxx = sum(p[0] ** 2 for p in pts); sxy = sum(p[0] * p[1] for p in pts)
b = (n * sxy - sx * sy) / (n * sxx - sx * sx); a = (sy - b * sx) / n
print(f"fit: fixed {a:.2f} us per tile + {b:.3f} us per K-step")
