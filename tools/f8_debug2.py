import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-llm_amd")]
import torch
from avllm import ops
M, N, K = 64, 128, 128
# W row n = one-hot at k = n (N = K = 128): out[m][n] = A[m][pair(n)]; A[m][k] = k + 1 (exact small ints need care: use 2^(k%4)*(1+ (k//4 %2)*0.5)) -> instead probe with A one-hot sweeps
W = torch.eye(128).bfloat16().cuda()
Wq, Ws = ops.mx_quantize(W, 1)
pairs = {}
for ka in range(128):
    A = torch.zeros(M, K); A[:, ka] = 1.0
    Aq, As = ops.mx_quantize(A.bfloat16().cuda(), 0)
    out = ops.gemm_f8(Aq, As, Wq, Ws).float().cpu()
    nz = out[0].nonzero().flatten().tolist()
    pairs[ka] = (nz, [float(out[0, n]) for n in nz])
bad = {k: v for k, v in pairs.items() if v[0] != [k] or v[1] != [1.0]}
print("mismatched k:", len(bad))
for k in sorted(bad)[:40]:
    print(k, bad[k])
