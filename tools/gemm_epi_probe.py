#!/usr/bin/env python3
"""Decomposition of a K = 768 launch of the persistent GEMM (variant 8) in an AVLLM_EXPERIMENT_KNOBS build: knob GEMM_DBG bit 0 = the epilogue computes but does not
store, 0x10000 = operands always cache hits (row 0), both = neither memory path; 0x40000 = operands L2-resident (tile 0)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops, lib as L

lib = L.load()
SHAPES = [(394000, 2304, 768, "clip qkv (bias)", 0, 0), (394000, 3072, 768, "clip fc1 + quick-GELU", L.ACT_QUICK_GELU, 0), ]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for M, N, K, tag, act, resid in SHAPES:
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16) if "llama" not in tag else None
    line = f"{tag:24s}"
    lib.avllm_set_gemm_variant(8)
    for name, dbg in [("as shipped", 0), ("no epilogue", 1), ("epilogue arithmetic, one store", 7 << 20), ("as shipped again", 0), ("row0", 0x10000), ("row0, arithmetic, one store", 0x10000 | (7 << 20)), ("row0 no epilogue", 0x10001)]:
        with L.knob("GEMM_DBG", dbg):
            best = 1e9
            for rep in range(3):
                ops.gemm(A, B, out=out, bias=bias, act=act, R=out if resid else None)
                e0.record()
                for _ in range(5):
                    ops.gemm(A, B, out=out, bias=bias, act=act, R=out if resid else None)
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / 5)
        line += f"  {name} {best * 1000:.0f} us ({2.0 * M * N * K / best / 1e12:.2f})"
    lib.avllm_set_gemm_variant(0)
    print(line, flush=True)
