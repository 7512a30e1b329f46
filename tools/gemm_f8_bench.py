#!/usr/bin/env python3
"""Block-scaled fp8 GEMM against the bf16 GEMM on the projection shapes of BASELINE config 5 (CLIP ViT-L/14, Whisper-large-v3,
Mistral-7B) and of the headline config; plus the activation quantiser."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops

SHAPES = [(4096, 4096, 4096, "llm q/o"), (4096, 6144, 4096, "mistral qkv"), (4096, 28672, 4096, "mistral gate+up"), (4096, 4096, 14336, "mistral down"),
          (192750, 3072, 1024, "vit-l qkv (750 fr)"), (192750, 4096, 1024, "vit-l fc1"), (192750, 1024, 4096, "vit-l fc2"),
          (24000, 3840, 1280, "whisper-l qkv"), (394000, 2304, 768, "vit-b qkv")]


def timed(fn, n):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for M, N, K, tag in SHAPES:
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    W = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    Aq, As = ops.mx_quantize(A, 0)
    Wq, Ws = ops.mx_quantize(W, 1)
    n = 10 if M < 100000 else 4
    t8 = timed(lambda: ops.gemm_f8(Aq, As, Wq, Ws, out=out), n)
    t16 = timed(lambda: ops.gemm(A, W, out=out), n)
    tq = timed(lambda: ops.mx_quantize(A, 0), n)
    fl = 2.0 * M * N * K
    print(f"{tag:20s} M={M:6d} N={N:5d} K={K:5d}  fp8 {t8*1000:8.1f} us {fl/t8/1e9:7.1f} TF/s | bf16 {t16*1000:8.1f} us {fl/t16/1e9:7.1f} TF/s | "
          f"quantise A {tq*1000:7.1f} us ({M*K*3/tq/1e9:5.2f} TB/s)", flush=True)
