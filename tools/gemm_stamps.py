#!/usr/bin/env python3
"""Where does a tile of the persistent GEMM spend its cycles?  Needs the diagnostic build of the library (tools/gemm_stamps.sh builds it with
-DAVLLM_GEMM_STAMPS): s_memtime stamps at the K-step boundaries, summed per wave.  Prints the mean per tile of each segment in cycles and in us
(clock from the s_memtime / s_memrealtime ratio; s_memrealtime runs at 100 MHz).  Read the SHARES: the stamps fence overlaps the real kernel has."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import numpy as np, torch
from avllm import ops, lib as L

SHAPES = [(394000, 2304, 768, "clip qkv", 0, 0), (394000, 3072, 768, "clip fc1 + quick-GELU", 2, 0), (394000, 768, 768, "clip out + residual", 0, 1),
          (394000, 768, 3072, "clip fc2 + residual", 0, 1), (4096, 22016, 4096, "llama gate+up", 0, 0), (4096, 4096, 4096, "llama q/k/v/o", 0, 0)]
lib = L.load()
lib.avllm_debug_read_gemm_stamps.argtypes = [C.c_void_p, C.c_int32]
names = ["K-step 0", "K-step 1", "K-steps 2..", "epilogue", "frag re-read"]
for M, N, K, tag, act, resid in SHAPES:
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16) * K ** -0.5
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    bias = torch.randn(N, device="cuda", dtype=torch.bfloat16) if act or resid else None
    for _ in range(3):
        ops.gemm(A, B, out=out, bias=bias, act=act, R=out if resid else None)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.gemm(A, B, out=out, bias=bias, act=act, R=out if resid else None); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * 4 * 8, dtype=np.uint64)
    L.check(lib.avllm_debug_read_gemm_stamps(buf.ctypes.data, buf.size))
    s = buf.reshape(256, 4, 8).astype(np.float64)
    tiles = s[..., 5].sum()
    ghz = (s[..., 6] / np.maximum(s[..., 7], 1)).mean() * 0.1          # cycles per 10 ns tick -> GHz
    per = [s[..., i].sum() / tiles for i in range(5)]
    nt = K // 64
    tot = sum(per)
    print(f"{tag:24s} M={M} N={N} K={K}: {e0.elapsed_time(e1) * 1000:8.1f} us (stamped build), clock {ghz:.2f} GHz, {tot:8.0f} cycles = {tot / ghz / 1000:6.2f} us per tile")
    for nm, v in zip(names, per):
        extra = f" ({v / (nt - 2):.0f} per K-step)" if nm == "K-steps 2.." and nt > 2 else ""
        print(f"    {nm:14s} {v:9.0f} cycles {v / ghz / 1000:6.2f} us {100 * v / tot:5.1f} %{extra}")
