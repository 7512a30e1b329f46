#!/usr/bin/env python3
"""GEMM micro-benchmark on the shapes of the hot path (random bf16 data).  AVLLM_GEMM_VARIANT=1 forces the 128x128 kernel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops, lib as L

MB = int(os.environ.get("ROWS", "4096"))       # B*S of the bench batch (16 x 256)
SHAPES = [  # (M, N, K, tag)
    (MB, 4096, 4096, "llama q/k/v/o"), (MB, 12288, 4096, "llama qkv fused"), (MB, 22016, 4096, "llama gate+up"),
    (MB, 4096, 11008, "llama down"), (MB, 4096, 22016, "llama d(gate,up)"), (MB, 11008, 4096, "llama d(down)"),
    (MB, 32000, 4096, "lm_head"), (MB, 4096, 32000, "d(lm_head)"),
    (394000, 2304, 768, "clip qkv"), (394000, 768, 768, "clip out"), (394000, 3072, 768, "clip fc1"), (394000, 768, 3072, "clip fc2"),
    (24000, 2304, 768, "whisper qkv"), (24000, 3072, 768, "whisper fc1"), (24000, 768, 3072, "whisper fc2"),
]
CALIBRATE = os.environ.get("CALIBRATE", "0") == "1"     # also time torch.matmul (hipBLASLt) on the same operands: a yardstick only

def main():
    dev = "cuda"
    act = int(os.environ.get("ACT", "0"))
    resid = os.environ.get("RESID", "0")                  # 1: bias + in-place residual (out-projection / fc2 / o / down as the model calls them); 2: residual from a second tensor
    for M, N, K, tag in SHAPES:
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        B = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev, dtype=torch.bfloat16) if act or resid != "0" else None
        R = out if resid == "1" else torch.zeros_like(out) if resid == "2" else None
        out.zero_()
        for _ in range(3):
            ops.gemm(A, B, out=out, bias=bias, act=act, R=R)
        torch.cuda.synchronize()
        n = 20 if M < 100000 else 8
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            ops.gemm(A, B, out=out, bias=bias, act=act, R=R)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        extra = ""
        if CALIBRATE and not act:
            Bt = B.t()
            for _ in range(3):
                torch.matmul(A, Bt, out=out)
            e0.record()
            for _ in range(n):
                torch.matmul(A, Bt, out=out)
            e1.record(); torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1) / n
            extra = f"   | hipBLASLt {ms2*1000:9.1f} us {2*M*N*K/ms2/1e9:8.1f} TF/s"
        print(f"{tag:18s} M={M:6d} N={N:5d} K={K:5d}  {ms*1000:9.1f} us  {2*M*N*K/ms/1e9:8.1f} TF/s{extra}", flush=True)

if __name__ == "__main__":
    main()
