"""More op-level parity (GPU): special-cased GEMM paths and full-size shapes checked through size-independent
properties (linearity) or against torch on the same device."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from avllm import ops  # noqa: E402
from avllm.lib import ACT_GELU as L_ACT_GELU  # noqa: E402
from test_ops_gpu import close, rnd  # noqa: E402


@pytest.mark.parametrize("M,K", [(2048, 4096), (300, 512), (257, 1024)])
def test_gemm_skinny_n64(dev, M, K):
    """N == 64 (LoRA rank side) takes the split-K-in-block kernel."""
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=1), rnd(64, K, dtype=torch.bfloat16, seed=2)
    out = ops.gemm(A, B, alpha=2.0)
    close(out, 2.0 * A.float() @ B.float().t(), 0.05 * (K ** 0.5) / 8, 2e-2, "skinny gemm")
    big = rnd(M, 3 * 64, dtype=torch.bfloat16, seed=3)
    dst = torch.zeros(M, 192, device=dev, dtype=torch.bfloat16)
    ops.gemm(A, B, out=dst[:, 64:128])                       # strided destination (column slice)
    close(dst[:, 64:128], A.float() @ B.float().t(), 0.05 * (K ** 0.5) / 8, 2e-2, "skinny gemm slice")
    assert dst[:, :64].abs().max().item() == 0 and dst[:, 128:].abs().max().item() == 0


@pytest.mark.parametrize("M,N,K,how", [(4096, 22016, 256, "cols"), (9000, 768, 128, "rows"), (24000, 2304, 64, "rows"), (4100, 22016, 64, "cols")])
def test_gemm_large_grids_with_fused_epilogue(dev, M, N, K, how):
    """Large ragged tile grids (tile counts just above a multiple of the CU count, edge tiles in M) with bias + GELU + residual and
    with the LoRA second K segment, against torch."""
    A = rnd(M, K, dtype=torch.bfloat16, seed=51)
    B = rnd(N, K, dtype=torch.bfloat16, seed=52, scale=K ** -0.5)
    bias = rnd(N, dtype=torch.bfloat16, seed=53)
    R = rnd(M, N, dtype=torch.bfloat16, seed=54)
    out = ops.gemm(A, B, bias=bias, R=R, act=L_ACT_GELU)
    ref = torch.nn.functional.gelu(A.float() @ B.float().t() + bias.float()) + R.float()
    close(out, ref, 4e-2, 2e-2, f"large-grid gemm ({how})")
    # LoRA second K segment through both parts
    A2, B2 = rnd(M, 64, dtype=torch.bfloat16, seed=55), rnd(N, 64, dtype=torch.bfloat16, seed=56, scale=0.1)
    out2 = ops.gemm(A, B, A2=A2, B2=B2)
    close(out2, A.float() @ B.float().t() + A2.float() @ B2.float().t(), 4e-2, 2e-2, "large-grid gemm with K2")


@pytest.mark.parametrize("r", [16, 32, 8])
def test_gemm_skinny_n_valid(dev, r):
    """Rank padded to 64: with n_valid = r only the real rank rows of the weight image are streamed; the padding columns come out as
    exact zeros and the real ones are identical to the full computation."""
    M, K = 4096, 4096
    x = rnd(M, K, dtype=torch.bfloat16, seed=41)
    A = torch.zeros(64, K, device=dev, dtype=torch.bfloat16)
    A[:r] = rnd(r, K, dtype=torch.bfloat16, seed=42)
    full = ops.gemm(x, A, alpha=2.0)
    part = ops.gemm(x, A, alpha=2.0, n_valid=r)
    nv = (r + 15) // 16 * 16
    assert torch.equal(part[:, :nv], full[:, :nv]) and float(part[:, nv:].abs().max() if nv < 64 else 0.0) == 0.0
    pd = ops.gemm(x, A, alpha=2.0, n_valid=r, a_drop=(9, 0.05))
    fd = ops.gemm(ops.dropout(x, 9, 0.05), A, alpha=2.0)
    assert torch.equal(pd[:, :nv], fd[:, :nv])


@pytest.mark.parametrize("M,NB", [(2048, 4096), (1000, 256), (70, 128)])
def test_gemm_tn_mfma(dev, M, NB):
    big = rnd(M, NB, dtype=torch.bfloat16, seed=4)
    small = rnd(M, 64, dtype=torch.bfloat16, seed=5)
    out = torch.zeros(NB, 16, device=dev)
    ops.gemm_tn(big, small, out, J=16, alpha=0.5)
    ref = 0.5 * big.float().t() @ small.float()[:, :16]
    close(out, ref, 2e-3 * (M ** 0.5), 1e-3, "gemm_tn mfma [NB,16]")
    out_t = torch.zeros(16, NB, device=dev)
    ops.gemm_tn(small, big, out_t, I=16)
    close(out_t, 2 * ref.t(), 4e-3 * (M ** 0.5), 1e-3, "gemm_tn mfma [16,NB]")
    ops.gemm_tn(small, big, out_t, I=16)                     # accumulates
    close(out_t, 4 * ref.t(), 8e-3 * (M ** 0.5), 1e-3, "gemm_tn accumulate")


@pytest.mark.parametrize("M,K", [(4096, 4096), (300, 512), (17, 256)])
def test_fused_dropout_equals_materialised(dev, M, K):
    """The rank-side GEMM and the dA reduction regenerate the dropout mask in-kernel: same result as running them on
    avllm_dropout's output (peft lora.Linear: lora_A(dropout(x)), clip_whisper_model.py:961-1005)."""
    seed, p = 1234, 0.05
    x = rnd(M, K, dtype=torch.bfloat16, seed=21)
    A = rnd(64, K, dtype=torch.bfloat16, seed=22)
    xd = ops.dropout(x, seed, p)
    frac = (xd == 0).float().mean().item()
    assert abs(frac - p) < 0.01 + 3 * (p / (M * K)) ** 0.5, frac
    t_ref = ops.gemm(xd, A, alpha=2.0)
    t = ops.gemm(x, A, alpha=2.0, a_drop=(seed, p))
    assert torch.equal(t, t_ref)                               # identical kernel and operand bits
    dt = rnd(M, 64, dtype=torch.bfloat16, seed=23)
    if K % 128 == 0:
        g_ref = torch.zeros(16, K, device=dev)
        ops.gemm_tn(dt, xd, g_ref, I=16)
        g = torch.zeros(16, K, device=dev)
        ops.gemm_tn(dt, x, g, I=16, drop=(seed, p))
        close(g, g_ref, 1e-3 * (M ** 0.5), 1e-4, "gemm_tn fused dropout")     # fp32 atomics: order only
    with pytest.raises(RuntimeError):
        ops.gemm(x, rnd(128, K, dtype=torch.bfloat16, seed=24), a_drop=(seed, p))      # only the N == 64 kernel implements it


def test_gemm_full_size_llama_shapes(dev):
    """Llama-2-7B projection shapes at the bench batch (M = 16*256 = 4096 rows: the persistent 4-wave kernel's grids), with and without the
    LoRA second K segment, against an fp32 torch product on a sample of rows (first / last rows of tiles included)."""
    M = 4096
    rows = torch.cat([torch.randperm(M, device=dev)[:384], torch.tensor([0, 127, 128, 255, 256, M - 257, M - 1], device=dev)])
    for N, K in ((4096, 4096), (12288, 4096), (22016, 4096), (4096, 11008), (32000, 4096)):
        A, B = rnd(M, K, dtype=torch.bfloat16, seed=6), rnd(N, K, dtype=torch.bfloat16, seed=7, scale=K ** -0.5)
        out = ops.gemm(A, B)
        ref = A[rows].float() @ B.float().t()
        close(out[rows], ref, 2e-2, 1e-2, f"gemm {M}x{N}x{K}")
        if N <= 12288:
            A2, B2 = rnd(M, 64, dtype=torch.bfloat16, seed=8), rnd(N, 64, dtype=torch.bfloat16, seed=9, scale=0.1)
            out2 = ops.gemm(A, B, A2=A2, B2=B2)
            close(out2[rows], ref + A2[rows].float() @ B2.float().t(), 2e-2, 1e-2, f"gemm+lora {M}x{N}x{K}+64")


def test_gemm_full_size_clip_shapes(dev):
    """CLIP ViT-B/16 projections at the bench size (16 clips x 125 frames x 197 tokens = 394000 rows, a ragged last row tile, 13860 tiles on
    256 CUs through the persistent kernel): qkv with bias, fc1 with bias + quick-GELU, fc2 with bias + in-place residual, checked on a
    sample of rows (first / last tile rows included) against fp32 torch."""
    from avllm import lib as L
    M = 394000
    rows = torch.cat([torch.randperm(M, device=dev)[:1500], torch.tensor([0, 255, 256, M - 257, M - 1], device=dev)])
    for N, K, act, res in ((2304, 768, L.ACT_NONE, False), (3072, 768, L.ACT_QUICK_GELU, False), (768, 3072, L.ACT_NONE, True)):
        A, B = rnd(M, K, dtype=torch.bfloat16, seed=8), rnd(N, K, dtype=torch.bfloat16, seed=9, scale=K ** -0.5)
        bias = rnd(N, dtype=torch.bfloat16, seed=10)
        x = rnd(M, N, dtype=torch.bfloat16, seed=11) if res else None
        xr = x[rows].float() if res else 0.0
        out = ops.gemm(A, B, out=x, bias=bias, R=x, act=act) if res else ops.gemm(A, B, bias=bias, act=act)
        ref = A[rows].float() @ B.float().t() + bias.float()
        if act == L.ACT_QUICK_GELU:
            ref = ref * torch.sigmoid(1.702 * ref)
        close(out[rows], ref + xr, 4e-2, 2e-2, f"clip gemm {M}x{N}x{K}")
        del A, B, out, x
        torch.cuda.empty_cache()


def test_attention_full_size(dev):
    """Bench-size attention: Llama (B8,H32,T256,hd128 causal) fwd+bwd, Whisper (T1500) and CLIP (197) fwd vs torch SDPA."""
    import torch.nn.functional as F
    for B, T, H, hd, causal, bwd in ((8, 256, 32, 128, True, True), (2, 1500, 12, 64, False, False), (64, 197, 12, 64, False, False)):
        qkv = rnd(B * T, 3 * H * hd, dtype=torch.bfloat16, seed=8)
        o, lse = ops.attention_fwd(qkv, B, T, H, hd, causal)
        x = qkv.float().requires_grad_(bwd)
        q, k, v = (t.view(B, T, H, hd).transpose(1, 2) for t in x.split(H * hd, dim=1))
        ref = F.scaled_dot_product_attention(q, k, v, is_causal=causal).transpose(1, 2).reshape(B * T, H * hd)
        close(o, ref.detach(), 3e-2, 2e-2, f"attention fwd T={T}")
        if bwd:
            dout = rnd(B * T, H * hd, dtype=torch.bfloat16, seed=9)
            ref.backward(dout.float())
            dqkv = ops.attention_bwd(qkv, o, dout, lse, B, T, H, hd, causal)
            err = (dqkv.float() - x.grad).pow(2).sum().sqrt() / x.grad.pow(2).sum().sqrt()
            assert err < 2e-2, err
            close(dqkv, x.grad, 8e-2, 5e-2, "attention bwd full size")


@pytest.mark.parametrize("dtype,impl", [(torch.bfloat16, 0), (torch.float32, 1)])
@pytest.mark.parametrize("B,T,H,Hkv,hd", [(2, 200, 8, 2, 128), (1, 77, 4, 1, 64), (2, 256, 32, 8, 128)])
def test_attention_grouped_query(dev, dtype, impl, B, T, H, Hkv, hd):
    """Grouped-query attention (HF repeat_kv, models/llama/modeling_llama.py:203-212): K/V hold Hkv heads, query head h reads
    head h // (H/Hkv); dK/dV sum over the group.  Checked against torch SDPA on the repeated K/V, forward and backward."""
    import torch.nn.functional as F
    qkv = rnd(B * T, (H + 2 * Hkv) * hd, dtype=dtype, seed=31)
    o, lse = ops.attention_fwd(qkv, B, T, H, hd, True, impl=impl, kv_heads=Hkv)
    x = qkv.float().requires_grad_(True)
    q, k, v = x.split([H * hd, Hkv * hd, Hkv * hd], dim=1)
    q = q.view(B, T, H, hd).transpose(1, 2)
    k = k.view(B, T, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    v = v.view(B, T, Hkv, hd).transpose(1, 2).repeat_interleave(H // Hkv, dim=1)
    ref = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(B * T, H * hd)
    tol = (3e-2, 2e-2) if dtype == torch.bfloat16 else (2e-5, 1e-5)
    close(o, ref.detach(), tol[0], tol[1], "gqa attention fwd")
    dout = rnd(B * T, H * hd, dtype=dtype, seed=32)
    ref.backward(dout.float())
    dqkv = ops.attention_bwd(qkv, o, dout, lse, B, T, H, hd, True, impl=impl, kv_heads=Hkv)
    err = (dqkv.float() - x.grad).pow(2).sum().sqrt() / x.grad.pow(2).sum().sqrt()
    assert err < (2e-2 if dtype == torch.bfloat16 else 1e-5), err
    for name, sl in (("dq", slice(0, H * hd)), ("dk", slice(H * hd, (H + Hkv) * hd)), ("dv", slice((H + Hkv) * hd, None))):
        e = (dqkv[:, sl].float() - x.grad[:, sl]).pow(2).sum().sqrt() / x.grad[:, sl].pow(2).sum().sqrt()
        assert e < (3e-2 if dtype == torch.bfloat16 else 1e-5), (name, e)


@pytest.mark.parametrize("variant", [1, 2, 5, 6, 7, 8, 9])
def test_every_gemm_tiling_agrees(dev, variant):
    """All bf16 tilings compiled into the library (A/B variants included) compute the same epilogue-fused GEMM."""
    from avllm import lib as L
    lib = L.load()
    M, N, K, K2 = 700, 520, 256, 64
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=41), rnd(N, K, dtype=torch.bfloat16, seed=42)
    A2, B2 = rnd(M, K2, dtype=torch.bfloat16, seed=43), rnd(N, K2, dtype=torch.bfloat16, seed=44)
    bias, R = rnd(N, dtype=torch.bfloat16, seed=45), rnd(M, N, dtype=torch.bfloat16, seed=46)
    ref = torch.nn.functional.gelu(A.float() @ B.float().t() + A2.float() @ B2.float().t() + bias.float()) + R.float()
    try:
        lib.avllm_set_gemm_variant(variant)
        out = ops.gemm(A, B, bias=bias, R=R, A2=A2, B2=B2, act=L.ACT_GELU)
    finally:
        lib.avllm_set_gemm_variant(0)
    close(out, ref, 0.3, 2e-2, f"gemm variant {variant}")


@pytest.mark.parametrize("K,K2", [(128, 0), (192, 0), (64, 64), (448, 64), (1024, 128)])
def test_gemm_4wave_kernel_ksteps_edges_epilogues(dev, K, K2):
    """The 4-wave 256x256 kernel (in-place LDS refill, two K-steps of prefetch): even and odd K-step counts, the minimum of two, a LoRA
    segment of one and two K-steps, M and N edges inside a tile, both epilogues, in-place residual, f32 output, row remap."""
    from avllm import lib as L
    lib = L.load()
    try:
        lib.avllm_set_gemm_variant(7)
        for M, N in ((900, 520), (257, 516), (512, 256)):           # ragged edges / narrow epilogue (N % 8 != 0) / exact tiles
            A, B = rnd(M, K, dtype=torch.bfloat16, seed=71), rnd(N, K, dtype=torch.bfloat16, seed=72)
            A2 = rnd(M, K2, dtype=torch.bfloat16, seed=73) if K2 else None
            B2 = rnd(N, K2, dtype=torch.bfloat16, seed=74) if K2 else None
            bias, x = rnd(N, dtype=torch.bfloat16, seed=75), rnd(M, N, dtype=torch.bfloat16, seed=76)
            acc = A.float() @ B.float().t() + (A2.float() @ B2.float().t() if K2 else 0.0)
            out = x.clone()
            ops.gemm(A, B, out=out, bias=bias, R=out, A2=A2, B2=B2, alpha=0.5)
            close(out, 0.5 * acc + bias.float() + x.float(), 0.02 * math.sqrt(K + K2), 2e-2, f"4-wave in-place residual {M}x{N}x{K}+{K2}")
            o32 = ops.gemm(A, B, bias=bias, A2=A2, B2=B2, out_f32=True, act=L.ACT_GELU)
            close(o32, torch.nn.functional.gelu(acc + bias.float()), 0.01 * math.sqrt(K + K2), 1e-2, f"4-wave f32 gelu {M}x{N}x{K}+{K2}")
        A, B = rnd(900, K, dtype=torch.bfloat16, seed=77), rnd(256, K, dtype=torch.bfloat16, seed=78)
        pos = rnd(9, 256, dtype=torch.bfloat16, seed=79)
        out = torch.zeros(1000, 256, device=dev, dtype=torch.bfloat16)
        ops.gemm(A, B, out=out, R=pos, r_mod=9, remap=(9, 10, 1), M=900)
        close(out.view(100, 10, 256)[:, 1:], (A.float() @ B.float().t()).view(100, 9, 256) + pos.float(), 0.02 * math.sqrt(K), 2e-2, "4-wave remap")
        assert out.view(100, 10, 256)[:, 0].abs().max().item() == 0
    finally:
        lib.avllm_set_gemm_variant(0)


def test_gemm_auto_dispatch_long_k_matches_16wave(dev):
    """K >= 4096 with a chip-filling grid goes to the 4-wave kernel by itself; the 16-wave kernel must give the same numbers (same
    MFMA, same K order inside a K-step; accumulation order across the two k-halves is identical too -> bit-equal bf16 outputs)."""
    from avllm import lib as L
    lib = L.load()
    M, N, K, K2 = 4096, 4096, 4096, 64
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=81), rnd(N, K, dtype=torch.bfloat16, seed=82, scale=K ** -0.5)
    A2, B2 = rnd(M, K2, dtype=torch.bfloat16, seed=83), rnd(N, K2, dtype=torch.bfloat16, seed=84, scale=0.1)
    R = rnd(M, N, dtype=torch.bfloat16, seed=85)
    auto = ops.gemm(A, B, R=R, A2=A2, B2=B2)
    try:
        lib.avllm_set_gemm_variant(5)
        ref16 = ops.gemm(A, B, R=R, A2=A2, B2=B2)
    finally:
        lib.avllm_set_gemm_variant(0)
    assert torch.equal(auto, ref16)
    rows = torch.randperm(M, device=dev)[:64]
    ref = A[rows].float() @ B.float().t() + A2[rows].float() @ B2.float().t() + R[rows].float()
    close(auto[rows], ref, 0.05, 2e-2, "auto-dispatched long-K gemm")


@pytest.mark.parametrize("K,K2,act", [(128, 0, "none"), (64, 64, "none"), (192, 64, "quick_gelu"), (320, 0, "gelu"), (4096, 64, "quick_gelu"), (4096, 64, "none")])
def test_gemm_persistent_kernel_many_tiles(dev, K, K2, act):
    """The persistent 4-wave kernel with more tiles than CUs (each workgroup walks several tiles, the K-step pipeline runs across tile
    boundaries, odd and even K-step counts so the buffer parity flips between tiles), ragged M / N edges, bias + in-place residual +
    activation: bit-equal to the 16-wave kernel (same MFMA order, same single rounding) and within bf16 tolerance of fp32 torch."""
    from avllm import lib as L
    lib = L.load()
    M, N = 4300, 4360                                              # 17 x 18 = 306 tiles on 256 CUs
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=91), rnd(N, K, dtype=torch.bfloat16, seed=92, scale=K ** -0.5)
    A2 = rnd(M, K2, dtype=torch.bfloat16, seed=93) if K2 else None
    B2 = rnd(N, K2, dtype=torch.bfloat16, seed=94, scale=0.1) if K2 else None
    bias, x = rnd(N, dtype=torch.bfloat16, seed=95), rnd(M, N, dtype=torch.bfloat16, seed=96)
    code = {"none": L.ACT_NONE, "gelu": L.ACT_GELU, "quick_gelu": L.ACT_QUICK_GELU}[act]
    outs = {}
    try:
        for variant in (8, 5):
            lib.avllm_set_gemm_variant(variant)
            o = x.clone()
            ops.gemm(A, B, out=o, bias=bias, R=o, A2=A2, B2=B2, act=code)
            outs[variant] = o
    finally:
        lib.avllm_set_gemm_variant(0)
    assert torch.equal(outs[8], outs[5])
    rows = torch.cat([torch.randperm(M, device=dev)[:48], torch.tensor([0, 255, 256, M - 1], device=dev)])
    acc = A[rows].float() @ B.float().t() + (A2[rows].float() @ B2.float().t() if K2 else 0.0) + bias.float()
    fn = {"none": lambda t: t, "gelu": torch.nn.functional.gelu, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t)}[act]
    close(outs[8][rows], fn(acc) + x[rows].float(), 0.06, 2e-2, f"persistent gemm K={K}+{K2} {act}")


@pytest.mark.parametrize("M,N", [(4300, 4360), (9000, 2304), (300, 136), (256, 128)])
@pytest.mark.parametrize("K,K2,act,resid", [(128, 0, "none", False), (64, 64, "none", True), (192, 64, "quick_gelu", False), (320, 0, "gelu", False),
                                            (768, 0, "none", True), (4096, 64, "none", True), (1024, 128, "none", False)])
def test_gemm_two_workgroup_kernel_many_tiles(dev, M, N, K, K2, act, resid):
    """The 256x128 persistent kernel that runs two workgroups per CU (csrc/gemm_dp.hip, variant 9): more tiles than workgroups (the three-stage
    K-step ring runs across tile boundaries, with and without the 16 epilogue stores in flight), one and many macro steps, a LoRA segment of one
    and two macro steps, ragged M / N edges and tiles smaller than a workgroup's, every lean epilogue form (plain, bias, bias + activation,
    in-place residual with and without bias): bit-equal to the 16-wave kernel (same MFMA, same k order per accumulator, one rounding)."""
    from avllm import lib as L
    lib = L.load()
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=191), rnd(N, K, dtype=torch.bfloat16, seed=192, scale=K ** -0.5)
    A2 = rnd(M, K2, dtype=torch.bfloat16, seed=193) if K2 else None
    B2 = rnd(N, K2, dtype=torch.bfloat16, seed=194, scale=0.1) if K2 else None
    x = rnd(M, N, dtype=torch.bfloat16, seed=196)
    code = {"none": L.ACT_NONE, "gelu": L.ACT_GELU, "quick_gelu": L.ACT_QUICK_GELU}[act]
    for bias in ((rnd(N, dtype=torch.bfloat16, seed=195), None) if act == "none" else (rnd(N, dtype=torch.bfloat16, seed=195),)):
        outs = {}
        try:
            for variant in (9, 5):
                lib.avllm_set_gemm_variant(variant)
                o = x.clone()
                ops.gemm(A, B, out=o, bias=bias, R=o if resid else None, A2=A2, B2=B2, act=code)
                outs[variant] = o
        finally:
            lib.avllm_set_gemm_variant(0)
        nd = int((outs[9] != outs[5]).sum())
        assert nd == 0, f"{nd} of {M * N} values differ (bias={bias is not None}); first at {(outs[9] != outs[5]).nonzero()[:4].tolist()}"
    rows = torch.cat([torch.randperm(M, device=dev)[:48], torch.tensor([0, min(255, M - 1), min(256, M - 1), M - 1], device=dev)])
    acc = A[rows].float() @ B.float().t() + (A2[rows].float() @ B2.float().t() if K2 else 0.0) + (bias.float() if bias is not None else 0.0)      # the last bias tried
    fn = {"none": lambda t: t, "gelu": torch.nn.functional.gelu, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t)}[act]
    close(outs[9][rows], fn(acc) + (x[rows].float() if resid else 0.0), 0.06, 2e-2, f"two-workgroup gemm K={K}+{K2} {act}")


@pytest.mark.parametrize("M", [1, 2, 8, 16])
def test_gemm_small_m_decode_path(dev, M):
    """M <= 16 (greedy decode): weight-streaming kernel, incl. LoRA second segment, residual in place and f32 logits."""
    N, K, K2 = 528, 512, 64
    A, B = rnd(M, K, dtype=torch.bfloat16, seed=51), rnd(N, K, dtype=torch.bfloat16, seed=52)
    A2, B2 = rnd(M, K2, dtype=torch.bfloat16, seed=53), rnd(N, K2, dtype=torch.bfloat16, seed=54)
    R = rnd(M, N, dtype=torch.bfloat16, seed=55)
    ref = A.float() @ B.float().t() + A2.float() @ B2.float().t() + R.float()
    out = R.clone()
    ops.gemm(A, B, out=out, R=out, A2=A2, B2=B2)
    close(out, ref, 0.25, 2e-2, "small-M gemm (in-place residual)")
    o32 = ops.gemm(A, B, out_f32=True)
    assert o32.dtype == torch.float32
    close(o32, A.float() @ B.float().t(), 0.05, 1e-2, "small-M gemm f32 out")
    rows = rnd(4 * M, K, dtype=torch.bfloat16, seed=56)          # strided rows (last position of each sequence)
    o = ops.gemm(rows[3::4], B, out_f32=True)
    close(o, rows[3::4].float() @ B.float().t(), 0.05, 1e-2, "small-M gemm strided A")


def test_gemm_16wave_narrow_and_wide_epilogues(dev):
    """The 256x256 kernel's two epilogues: N % 8 != 0 or unaligned pointers take the per-lane path, everything else the
    LDS-staged 16-byte path; both with row remap / broadcast residual / in-place residual / f32 output."""
    from avllm import lib as L
    lib = L.load()
    try:
        lib.avllm_set_gemm_variant(5)
        M, K = 900, 128
        for N in (516, 520):                                   # narrow, wide
            A, B = rnd(M, K, dtype=torch.bfloat16, seed=61), rnd(N, K, dtype=torch.bfloat16, seed=62)
            bias = rnd(N, dtype=torch.bfloat16, seed=63)
            x = rnd(M, N, dtype=torch.bfloat16, seed=64)
            ref = (A.float() @ B.float().t() + bias.float()) * 1.0 + x.float()
            out = x.clone()
            ops.gemm(A, B, out=out, bias=bias, R=out)          # in-place residual
            close(out, ref, 0.15, 2e-2, f"in-place residual N={N}")
            o32 = ops.gemm(A, B, bias=bias, out_f32=True, act=L.ACT_QUICK_GELU)
            r = A.float() @ B.float().t() + bias.float()
            close(o32, r * torch.sigmoid(1.702 * r), 0.05, 1e-2, f"f32 out quick_gelu N={N}")
        N = 256
        A, B = rnd(M, K, dtype=torch.bfloat16, seed=65), rnd(N, K, dtype=torch.bfloat16, seed=66)
        pos = rnd(9, N, dtype=torch.bfloat16, seed=67)
        out = torch.zeros(100 * 10, N, device=dev, dtype=torch.bfloat16)
        ops.gemm(A, B, out=out, R=pos, r_mod=9, remap=(9, 10, 1), M=900)
        ref = (A.float() @ B.float().t()).view(100, 9, N) + pos.float()
        close(out.view(100, 10, N)[:, 1:], ref, 0.1, 2e-2, "remap + broadcast residual (wide)")
        assert out.view(100, 10, N)[:, 0].abs().max().item() == 0
    finally:
        lib.avllm_set_gemm_variant(0)


@pytest.mark.parametrize("T", [1, 10, 16, 17, 50, 197, 208, 209, 257, 272])
def test_attention_short_noncausal_every_length(dev, T):
    """The whole-sequence-in-registers kernel (hd 64, non-causal, T <= 272: CLIP ViT-B/16 = 197, ViT-L/14 = 257): every block-edge
    length, output and log-sum-exp against fp32 torch; plus one dominant key per query (softmax far from uniform) and grouped heads."""
    from test_ops_gpu import _attn_ref
    B, H, hd = 3, 2, 64
    qkv = rnd(B * T, 3 * H * hd, dtype=torch.bfloat16, seed=31)
    if T > 4:
        qkv[T // 2, H * hd:H * hd + hd] = qkv[T - 1, 0:hd] * 6          # a key aligned with the last query of item 0, head 0
    o, lse = ops.attention_fwd(qkv, B, T, H, hd, False)
    ro, rl = _attn_ref(qkv, B, T, H, hd, False)
    close(o, ro, 3e-2, 2e-2, f"short attention T={T}")
    close(lse, rl, 2e-2, 1e-4, "lse")
    if T in (50, 197):                                                    # 4 query heads on 2 key/value heads
        Hq, Hkv = 4, 2
        x = rnd(B * T, (Hq + 2 * Hkv) * hd, dtype=torch.bfloat16, seed=32)
        og, _ = ops.attention_fwd(x, B, T, Hq, hd, False, kv_heads=Hkv)
        q, k, v = x.float().split([Hq * hd, Hkv * hd, Hkv * hd], dim=1)
        k = k.view(B * T, Hkv, 1, hd).expand(-1, -1, Hq // Hkv, -1).reshape(B * T, Hq * hd)
        v = v.view(B * T, Hkv, 1, hd).expand(-1, -1, Hq // Hkv, -1).reshape(B * T, Hq * hd)
        rg, _ = _attn_ref(torch.cat([q, k, v], 1), B, T, Hq, hd, False)
        close(og, rg, 3e-2, 2e-2, f"short attention grouped T={T}")
