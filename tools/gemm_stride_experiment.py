#!/usr/bin/env python3
"""Does the row stride of the operands matter to the persistent GEMM (L2 channel aliasing of power-of-two strides)?  Same products with the
operands stored in wider allocations: row stride K + PAD elements (PAD = 0, 64, 128, 192 ...; 64 elements = 128 bytes = one cache line)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops, lib as L

def run(M, N, K, tag, variant):
    lib = L.load()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    line = f"v{variant} {tag:12s} M={M} N={N} K={K}:"
    for pad in [int(p) for p in os.environ.get("PADS", "0,64,128,192,320,1088").split(",")]:
        A = torch.randn(M, K + pad, device="cuda", dtype=torch.bfloat16)[:, :K]
        B = (torch.randn(N, K + pad, device="cuda", dtype=torch.bfloat16) * K ** -0.5)[:, :K]
        lib.avllm_set_gemm_variant(variant)
        best = 1e9
        for rep in range(3):
            ops.gemm(A, B, out=out)
            e0.record()
            for _ in range(6):
                ops.gemm(A, B, out=out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 6)
        lib.avllm_set_gemm_variant(0)
        line += f"  pad {pad}: {best * 1000:.0f} us ({2.0 * M * N * K / best / 1e12:.2f} PF/s)"
        del A, B
    print(line, flush=True)

for v in (8, 9):
    run(4096, 4096, 4096, "q/k/v/o", v)
    run(4096, 22016, 4096, "gate|up", v)
    run(4096, 4096, 11008, "down", v)
    run(4096, 4096, 32000, "d(lm_head)", v)
    run(100000, 2304, 768, "clip qkv/4", v)
