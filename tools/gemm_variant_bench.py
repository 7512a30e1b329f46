#!/usr/bin/env python3
"""Two GEMM kernels side by side on the hot path's shapes, alternating launch by launch on one box:  VARIANTS=8,9 (default: the one-wave-per-SIMD
persistent kernel against the two-workgroup kernel).  EPI=plain|bias|act|res per shape as the model calls it (default: as the model)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops, lib as L

MB = int(os.environ.get("ROWS", "4096"))
CL = int(os.environ.get("CLIP_ROWS", "394000"))
SHAPES = [  # (M, N, K, K2, epilogue, tag)
    (CL, 2304, 768, 0, "bias", "clip qkv"), (CL, 768, 768, 0, "res", "clip out+res"), (CL, 3072, 768, 0, "act", "clip fc1+qgelu"), (CL, 768, 3072, 0, "res", "clip fc2+res"),
    (24000, 2304, 768, 0, "bias", "whisper qkv"), (24000, 768, 768, 0, "res", "whisper out+res"), (24000, 3072, 768, 0, "act", "whisper fc1"), (24000, 768, 3072, 0, "res", "whisper fc2+res"),
    (MB, 4096, 4096, 64, "plain", "llama q/k/v +lora"), (MB, 4096, 4096, 64, "res0", "llama o +lora +res"), (MB, 22016, 4096, 64, "plain", "llama gate|up +lora"),
    (MB, 4096, 11008, 64, "res0", "llama down +lora +res"), (MB, 32000, 4096, 0, "plain", "lm_head"), (MB, 4096, 32000, 0, "plain", "d(lm_head)"),
    (MB, 4096, 22016, 0, "plain", "d(gate,up)"), (MB, 11008, 4096, 0, "plain", "d(down)"),
]

def main():
    dev = "cuda"
    lib = L.load()
    variants = [int(v) for v in os.environ.get("VARIANTS", "8,9").split(",")]
    only = os.environ.get("ONLY")
    for M, N, K, K2, epi, tag in SHAPES:
        if only and only not in tag:
            continue
        A = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
        B = torch.randn(N, K, device=dev, dtype=torch.bfloat16) * K ** -0.5
        A2 = torch.randn(M, K2, device=dev, dtype=torch.bfloat16) if K2 else None
        B2 = torch.randn(N, K2, device=dev, dtype=torch.bfloat16) * 0.05 if K2 else None
        out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
        bias = torch.randn(N, device=dev, dtype=torch.bfloat16) if epi in ("bias", "act", "res") else None
        act = (L.ACT_QUICK_GELU if "clip" in tag else L.ACT_GELU) if epi == "act" else L.ACT_NONE
        R = out if epi in ("res", "res0") else None
        n = 16 if M < 100000 else 6
        res = {v: [] for v in variants}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        try:
            for rep in range(3):
                for v in variants:
                    lib.avllm_set_gemm_variant(v)
                    ops.gemm(A, B, out=out, bias=bias, act=act, R=R, A2=A2, B2=B2)
                    e0.record()
                    for _ in range(n):
                        ops.gemm(A, B, out=out, bias=bias, act=act, R=R, A2=A2, B2=B2)
                    e1.record(); torch.cuda.synchronize()
                    res[v].append(e0.elapsed_time(e1) / n)
        finally:
            lib.avllm_set_gemm_variant(0)
        fl = 2.0 * M * N * (K + K2)
        line = f"{tag:24s} M={M:6d} N={N:5d} K={K:5d}+{K2:2d}"
        for v in variants:
            ms = min(res[v])
            line += f"  | v{v}: {ms * 1000:8.1f} us {fl / ms / 1e12:6.3f} PF/s"
        print(line, flush=True)

if __name__ == "__main__":
    main()
