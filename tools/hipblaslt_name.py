#!/usr/bin/env python3
"""Print nothing; just run torch.matmul on the hot-path shapes so a rocprofv3 --kernel-trace shows which hipBLASLt kernels
(macro tile, MFMA shape, wave layout are encoded in the Tensile kernel name) the library picks.  Yardstick only."""
import torch
for M, N, K in ((4096, 4096, 4096), (4096, 22016, 4096), (4096, 4096, 11008), (394000, 2304, 768)):
    A = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    B = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        torch.matmul(A, B.t())
    torch.cuda.synchronize()
