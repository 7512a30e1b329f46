#!/usr/bin/env python3
"""Weight-streaming rate of the decode-step projections (Llama-2-7B shapes, bf16, B rows): avllm_dec_proj (fused forms) next to the
round-1 small-M avllm_gemm, each over a rotation of weight copies larger than the 256 MB of L2 + MALL so every launch streams from HBM."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm import ops

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--iters", type=int, default=40)
a = ap.parse_args()
dev, BF, M = "cuda:0", torch.bfloat16, a.batch
d, f, V, hd = 4096, 11008, 32000, 128
g = torch.Generator(device=dev).manual_seed(0)
rope = torch.rand(hd // 2, 2, device=dev, generator=g)
norm = torch.ones(d, device=dev, dtype=BF)


def timeit(fn, n):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for name, rows, K, mode in (("q|k|v + norm + rope + append", 3 * d, d, 2), ("o + residual", d, d, 0), ("gate|up + norm + SwiGLU", 2 * f, d, 1),
                            ("down + residual", d, f, 0), ("lm_head + norm (f32 out)", V, d, 0)):
    ncopy = max(2, int(1.2e9 // (rows * K * 2)))
    Ws = [(torch.randn(rows, K, device=dev, generator=g) * K ** -0.5).to(BF) for _ in range(ncopy)]
    A = torch.randn(M, K, device=dev, generator=g).to(BF)
    kc = torch.zeros(M, 320, d, device=dev, dtype=BF); vc = torch.zeros_like(kc)
    x = torch.zeros(M, d, device=dev, dtype=BF)
    if mode == 2:
        new = lambda i: ops.dec_proj(A, Ws[i % ncopy], mode=2, norm_w=norm, rope=rope, kc=kc, vc=vc, pos=300, dq=d, dkv=d, hd=hd)
        out_old = torch.empty(M, rows, device=dev, dtype=BF)
        old = lambda i: ops.gemm(A, Ws[i % ncopy], out=out_old)
    elif mode == 1:
        new = lambda i: ops.dec_proj(A, Ws[i % ncopy], mode=1, norm_w=norm)
        out_old = torch.empty(M, rows, device=dev, dtype=BF)
        old = lambda i: ops.gemm(A, Ws[i % ncopy], out=out_old)
    elif rows == V:
        o32 = torch.empty(M, V, device=dev, dtype=torch.float32)
        new = lambda i: ops.dec_proj(A, Ws[i % ncopy], norm_w=norm, out=o32)
        old = lambda i: ops.gemm(A, Ws[i % ncopy], out=o32, out_f32=True)
    else:
        new = lambda i: ops.dec_proj(A, Ws[i % ncopy], R=x, out=x)
        old = lambda i: ops.gemm(A, Ws[i % ncopy], out=x, R=x)
    tn, to = timeit(new, a.iters), timeit(old, a.iters)
    gb = rows * K * 2 / 1e9
    print(f"{name:32s} [{rows:6d} x {K:5d}] {gb * 1e3:7.1f} MB  dec_proj {tn * 1e6:7.1f} us = {gb / tn / 1e3:5.2f} TB/s   small-M gemm {to * 1e6:7.1f} us = {gb / to / 1e3:5.2f} TB/s", flush=True)
    del Ws
