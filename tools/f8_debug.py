import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-llm_amd")]
import torch
from avllm import ops
torch.manual_seed(0)
M, N, K = 64, 128, 128
ar_m = torch.arange(M).float(); ar_n = torch.arange(N).float(); ar_k = (torch.arange(K) // 32).float()
blk = lambda b: (ar_k == b).float()[None, :]
cases = {}
for b in range(4):
    cases[f"A blocks 2^g x 3^row%2, W only block {b}"] = (torch.ones(M, K) * 2.0 ** ar_k[None, :] * (1 + 0.5 * (ar_m % 2))[:, None], torch.ones(N, K) * blk(b))
cases["A rows x blocks pow2 (m%3 + g)"] = (torch.ones(M, K) * 2.0 ** (ar_m % 3)[:, None] * 2.0 ** ar_k[None, :], torch.ones(N, K))
cases["A rows x blocks pow2 (m%3 * g)"] = (torch.ones(M, K) * 2.0 ** ((ar_m % 3)[:, None] * ar_k[None, :]), torch.ones(N, K))
for name, (A, W) in cases.items():
    A = A.bfloat16().cuda(); W = W.bfloat16().cuda()
    Aq, As = ops.mx_quantize(A, 0); Wq, Ws = ops.mx_quantize(W, 1)
    out = ops.gemm_f8(Aq, As, Wq, Ws).float().cpu()
    ref = A.float().cpu() @ W.float().cpu().T
    bad = (out != ref)
    print(f"{name:20s} wrong entries {int(bad.sum())}/{out.numel()}")
    if bad.any():
        r = (out / ref)
        rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
        print("   wrong rows:", rows[:40]); print("   wrong cols:", cols[:40])
        i, j = bad.nonzero()[0].tolist()
        print("   first wrong (m,n)=", (i, j), "out", float(out[i, j]), "ref", float(ref[i, j]), " ratios row", r[i, :8].tolist())
