#!/usr/bin/env python3
"""Where does the two-workgroup GEMM (variant 9) lose its time?  Same launches with the kernel's experiment bits (knob GEMM_DBG):
0x40 (variant 8 in an AVLLM_EXPERIMENT_KNOBS build: 0x10000) = every DMA lane fetches row 0 (always a cache hit: the memory path out of the picture), 0x80 = one workgroup per CU, 0x10 = no start stagger."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops, lib as L

def run(M, N, K, tag):
    lib = L.load()
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    line = f"{tag:12s} M={M} N={N} K={K}:"
    for name, variant, dbg in (("v8", 8, 0), ("v8 row0", 8, 0x10000), ("v8 tile0", 8, 0x40000), ("v9", 9, 0), ("v9 row0", 9, 0x40), ("v9 tile0", 9, 0x800)):
        lib.avllm_set_gemm_variant(variant)
        with L.knob("GEMM_DBG", dbg):
            best = 1e9
            for rep in range(3):
                ops.gemm(A, B, out=out)
                e0.record()
                for _ in range(6):
                    ops.gemm(A, B, out=out)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 6)
        line += f"  {name} {best * 1000:.0f} us ({2.0 * M * N * K / best / 1e12:.2f} PF/s)"
    lib.avllm_set_gemm_variant(0)
    print(line, flush=True)

run(394000, 2304, 768, "clip qkv")
run(4096, 4096, 32000, "d(lm_head)")
run(4096, 22016, 4096, "gate|up")
run(4096, 22016, 1024, "K=1024")
run(4096, 4096, 4096, "q/k/v/o")
