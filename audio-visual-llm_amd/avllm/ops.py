"""Tensor-level wrappers over the op-level C ABI (include/avllm.h).  Tensors only carry pointers/strides;
all arithmetic happens in libavllm.so on the current torch stream."""
from __future__ import annotations

import ctypes as C

import torch

from . import lib as L


def _ld(t):
    assert t.stride(-1) == 1, "innermost dimension must be contiguous"
    return t.stride(-2) if t.dim() >= 2 else t.shape[-1]


def gemm(A, B, out=None, bias=None, R=None, A2=None, B2=None, act=L.ACT_NONE, alpha=1.0, out_f32=False,
         r_mod=0, remap=None, M=None, a_drop=None, n_valid=0, drop=None):
    """out[M,N] = act(alpha*(A.B^T + A2.B2^T) + bias) + R.  A [M,K] (row stride free), B [N,K].
    a_drop=(seed, p): A is replaced by dropout(A) on the fly (bf16, N == 64 only).
    drop=(seed, p): the product (before +R) is multiplied by the dropout mask keep(seed, m*N+n, p)/(1-p) -- the adapter's
    input-gradient GEMM of the training backward (csrc/engine.hip)."""
    lib = L.load()
    M = A.shape[0] if M is None else M
    N, K = B.shape[0], B.shape[1]
    dt = L.dt_of(A)
    if out is None:
        rows = M if remap is None else (M // remap[0]) * remap[1]
        out = torch.empty(rows, N, device=A.device, dtype=torch.float32 if (out_f32 or dt == L.F32) else A.dtype)
    d = L.GemmDesc()
    d.A, d.B, d.C = L.ptr(A), L.ptr(B), L.ptr(out)
    d.lda, d.ldb, d.ldc = _ld(A), _ld(B), _ld(out)
    d.M, d.N, d.K = M, N, K
    if A2 is not None:
        d.A2, d.B2, d.lda2, d.ldb2, d.K2 = L.ptr(A2), L.ptr(B2), _ld(A2), _ld(B2), B2.shape[1]
    d.bias = L.ptr(bias)
    if R is not None:
        d.R, d.ldr = L.ptr(R), _ld(R)
    d.dtype, d.out_f32, d.act, d.alpha, d.r_mod = dt, int(out_f32), act, alpha, r_mod
    if remap is not None:
        d.g_in, d.g_out, d.g_off = remap
    if a_drop is not None:
        d.a_drop_seed, d.a_drop_p = a_drop[0] & 0xFFFFFFFF, a_drop[1]
    if drop is not None:
        d.drop_seed, d.drop_p = drop[0] & 0xFFFFFFFF, drop[1]
    d.n_valid = n_valid
    L.check(lib.avllm_gemm(C.byref(d), L.stream_ptr()))
    return out


def gemm_tn(P, Q, out, I=None, J=None, alpha=1.0, drop=None):
    """out[I,J] += alpha * P[:, :I]^T . Q[:, :J]; drop=(seed, p) applies dropout to the wide operand on the fly."""
    lib = L.load()
    I = P.shape[1] if I is None else I
    J = Q.shape[1] if J is None else J
    if drop is not None:
        L.check(lib.avllm_gemm_tn_drop(L.ptr(P), _ld(P), I, L.ptr(Q), _ld(Q), J, P.shape[0], L.ptr(out), _ld(out), alpha,
                                       drop[0] & 0xFFFFFFFF, drop[1], L.dt_of(P), L.stream_ptr()))
        return out
    L.check(lib.avllm_gemm_tn(L.ptr(P), _ld(P), I, L.ptr(Q), _ld(Q), J, P.shape[0], L.ptr(out), _ld(out), alpha,
                              L.dt_of(P), L.stream_ptr()))
    return out


def layernorm(x, w, b, eps=1e-5):
    y = torch.empty_like(x)
    L.check(L.load().avllm_layernorm(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), x.numel() // x.shape[-1], x.shape[-1], eps,
                                     L.dt_of(x), L.stream_ptr()))
    return y


def rmsnorm_fwd(x, w, eps):
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    L.check(L.load().avllm_rmsnorm_fwd(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(rstd), rows, x.shape[-1], eps, L.dt_of(x), L.stream_ptr()))
    return y, rstd


def rmsnorm_bwd(dy, x, w, rstd, dres=None):
    dx = torch.empty_like(x)
    L.check(L.load().avllm_rmsnorm_bwd(L.ptr(dy), L.ptr(x), L.ptr(w), L.ptr(rstd), L.ptr(dres), L.ptr(dx),
                                       x.numel() // x.shape[-1], x.shape[-1], L.dt_of(x), L.stream_ptr()))
    return dx


def rope_(x2d, T, heads, hd, pos0=0, theta=10000.0, inverse=False):
    """In place on a [rows, heads*hd] view (row stride free)."""
    L.check(L.load().avllm_rope(L.ptr(x2d), _ld(x2d), x2d.shape[0], T, heads, hd, pos0, theta, int(inverse), L.dt_of(x2d), L.stream_ptr()))
    return x2d


def swiglu_fwd(gu):
    M, F2 = gu.shape
    h = torch.empty(M, F2 // 2, device=gu.device, dtype=gu.dtype)
    L.check(L.load().avllm_swiglu_fwd(L.ptr(gu), L.ptr(h), M, F2 // 2, L.dt_of(gu), L.stream_ptr()))
    return h


def swiglu_bwd(dh, gu):
    dgu = torch.empty_like(gu)
    L.check(L.load().avllm_swiglu_bwd(L.ptr(dh), L.ptr(gu), L.ptr(dgu), gu.shape[0], gu.shape[1] // 2, L.dt_of(gu), L.stream_ptr()))
    return dgu


def attention_fwd(qkv, B, T, H, hd, causal, scale=None, impl=0, want_lse=True, kv_heads=None):
    """qkv [B*T, (H + 2*kv_heads)*hd] fused rows [q | k | v] -> (o [B*T, H*hd], lse [B,H,T])."""
    d = H * hd
    dkv = (kv_heads or H) * hd
    o = torch.empty(B * T, d, device=qkv.device, dtype=qkv.dtype)
    lse = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32) if want_lse else None
    es = qkv.element_size()
    scale = hd ** -0.5 if scale is None else scale
    p = L.ptr(qkv)
    L.check(L.load().avllm_attention_fwd(p, p + d * es, p + (d + dkv) * es, L.ptr(o), L.ptr(lse), B, T, T, H, hd, _ld(qkv), _ld(qkv),
                                         _ld(qkv), d, scale, int(causal), L.dt_of(qkv), impl, kv_heads or 0, L.stream_ptr()))
    return o, lse


def attention_fwd_mxq(qkv, B, T, H, hd, scale=None):
    """Short non-causal self-attention (T <= 272, hd 64, bf16) with the output block-scaled to e4m3 in the kernel's epilogue:
    qkv [B*T, 3*H*hd] -> (codes uint8 [B*T, H*hd], layout-0 scale image) == mx_quantize(attention_fwd(...)[0], 0) bit for bit."""
    lib = L.load()
    d = H * hd
    q = torch.empty(B * T, d, device=qkv.device, dtype=torch.uint8)
    s = torch.zeros(lib.avllm_mx_scale_bytes(B * T, d), device=qkv.device, dtype=torch.uint8)
    es = qkv.element_size()
    p = L.ptr(qkv)
    L.check(lib.avllm_attention_fwd_mxq(p, p + d * es, p + 2 * d * es, L.ptr(q), d, L.ptr(s), B, T, H, hd, _ld(qkv), _ld(qkv), _ld(qkv),
                                        float(hd ** -0.5 if scale is None else scale), L.stream_ptr()))
    return q, s


def attention_bwd(qkv, o, dout, lse, B, T, H, hd, causal, scale=None, impl=0, kv_heads=None):
    d = H * hd
    dkv = (kv_heads or H) * hd
    dqkv = torch.empty_like(qkv)
    delta = torch.empty(B, H, T, device=qkv.device, dtype=torch.float32)
    es = qkv.element_size()
    scale = hd ** -0.5 if scale is None else scale
    p, g = L.ptr(qkv), L.ptr(dqkv)
    L.check(L.load().avllm_attention_bwd(p, p + d * es, p + (d + dkv) * es, L.ptr(o), L.ptr(dout), L.ptr(lse), g, g + d * es, g + (d + dkv) * es,
                                         L.ptr(delta), B, T, H, hd, _ld(qkv), _ld(qkv), _ld(qkv), d, _ld(dqkv), _ld(dqkv), _ld(dqkv),
                                         scale, int(causal), L.dt_of(qkv), impl, kv_heads or 0, L.stream_ptr()))
    return dqkv


def ce_fwd(logits, labels):
    """logits [B,T,V], labels int64 [B,T] (-100 = ignore) -> (row_lse [B*T], loss_sum [1], count [1])."""
    B, T, V = logits.shape
    row_lse = torch.empty(B * T, device=logits.device, dtype=torch.float32)
    acc = torch.zeros(2, device=logits.device, dtype=torch.float32)
    L.check(L.load().avllm_ce_fwd(L.ptr(logits), logits.stride(1), L.ptr(labels), B, T, V, L.ptr(row_lse), L.ptr(acc), L.ptr(acc) + 4,
                                  L.dt_of(logits), L.stream_ptr()))
    return row_lse, acc


def ce_bwd(logits, labels, row_lse, acc, grad_scale=1.0):
    B, T, V = logits.shape
    dl = torch.empty_like(logits)
    L.check(L.load().avllm_ce_bwd(L.ptr(logits), logits.stride(1), L.ptr(labels), L.ptr(row_lse), L.ptr(acc) + 4, grad_scale, L.ptr(dl),
                                  B, T, V, L.dt_of(logits), L.stream_ptr()))
    return dl


def argmax_rows(logits2d):
    out = torch.empty(logits2d.shape[0], device=logits2d.device, dtype=torch.int64)
    L.check(L.load().avllm_argmax_rows(L.ptr(logits2d), _ld(logits2d), logits2d.shape[0], logits2d.shape[1], L.ptr(out),
                                       L.dt_of(logits2d), L.stream_ptr()))
    return out


def embedding(table, ids):
    ids = ids.contiguous()
    out = torch.empty(*ids.shape, table.shape[1], device=table.device, dtype=table.dtype)
    L.check(L.load().avllm_embedding(L.ptr(table), L.ptr(ids), L.ptr(out), ids.numel(), table.shape[1], L.dt_of(table), L.stream_ptr()))
    return out


def cast(x, dtype):
    if x.dtype == dtype:
        return x
    x = x.contiguous()
    out = torch.empty_like(x, dtype=dtype)
    if x.numel():
        L.check(L.load().avllm_cast(L.ptr(x), L.dt_of(x), L.ptr(out), L.dt_of(out), x.numel(), L.stream_ptr()))
    return out


def act_residual(x, act, r=None):
    """act(x) + r elementwise (avllm_act_residual)."""
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.load().avllm_act_residual(L.ptr(x), L.ptr(r.contiguous()) if r is not None else None, L.ptr(y), x.numel(), act, L.dt_of(x), L.stream_ptr()))
    return y


def conv1d_k3(x, w2d, bias, stride=1):
    """nn.Conv1d(kernel_size=3, padding=1, stride) on token-major x [B,T,C]: avllm_im2col_k3 + avllm_gemm.  w2d = weight [out, C, 3] reshaped to
    [out, kw*C + c] (permute(0, 2, 1)), zero-padded along K to a multiple of 64 by the caller when 3C is not one."""
    B, T, Cc = x.shape
    To = (T - 1) // stride + 1
    K = w2d.shape[1]
    x = x.contiguous()
    cols = torch.empty(B * To, 3 * Cc, device=x.device, dtype=x.dtype) if K == 3 * Cc else torch.zeros(B * To, K, device=x.device, dtype=x.dtype)
    if K == 3 * Cc:
        L.check(L.load().avllm_im2col_k3(L.ptr(x), L.ptr(cols), B, T, Cc, stride, L.dt_of(x), L.stream_ptr()))
    else:                                                       # padded K: im2col into a dense scratch, then one strided copy into the padded rows
        dense = torch.empty(B * To, 3 * Cc, device=x.device, dtype=x.dtype)
        L.check(L.load().avllm_im2col_k3(L.ptr(x), L.ptr(dense), B, T, Cc, stride, L.dt_of(x), L.stream_ptr()))
        cols[:, : 3 * Cc] = dense
    return gemm(cols, w2d, bias=bias).view(B, To, -1)


def groupnorm_tokens(x, w, b, groups, eps=1e-5, act=L.ACT_NONE):
    """nn.GroupNorm(groups, C) of the [B,C,T] view of x [B,T,C] (+ activation)."""
    B, T, Cc = x.shape
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.load().avllm_groupnorm_tokens(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), B, T, Cc, groups, eps, act, L.dt_of(x), L.stream_ptr()))
    return y


def mha_self(x, in_w, in_b, out_w, out_b, heads):
    """nn.MultiheadAttention(batch_first=True)(x, x, x) in eval mode: packed in-projection (avllm_gemm), softmax(QK^T / sqrt(hd)) V per head
    (avllm_attention_fwd; head dims up to 512 go to the scalar kernel), out-projection.  x [B,T,E]."""
    B, T, E = x.shape
    qkv = gemm(x.reshape(B * T, E), in_w, bias=in_b)
    o, _ = attention_fwd(qkv, B, T, heads, E // heads, causal=False, want_lse=False)
    return gemm(o, out_w, bias=out_b).view(B, T, E)


def fuse_pool(a, v, prompt_emb, L_, S_out, fusion_scale, D, B):
    """See avllm_fuse_pool in include/avllm.h.  a [B,Ta,D] | None, v [B,Tv,D] | None, prompt_emb [B,P,D] | None."""
    ref = a if a is not None else v
    out = torch.empty(B, S_out, D, device=ref.device, dtype=ref.dtype)
    Ta = a.shape[1] if a is not None else 0
    Tv = v.shape[1] if v is not None else 0
    P = prompt_emb.shape[1] if prompt_emb is not None else 0
    L.check(L.load().avllm_fuse_pool(L.ptr(a), Ta, L.ptr(v), Tv, L.ptr(prompt_emb), P, L.ptr(out), B, L_, S_out, D, fusion_scale,
                                     L.dt_of(ref), L.stream_ptr()))
    return out


def grad_sumsq(g, out, partials=None):
    """out += sum g^2 (float atomics), or -- with a `partials` scratch tensor (float32, <= 1024 used) -- out = sum g^2 in a fixed order."""
    if partials is not None:
        L.check(L.load().avllm_grad_sumsq_det(L.ptr(g), g.numel(), L.ptr(partials), partials.numel(), L.ptr(out), L.stream_ptr()))
        return
    L.check(L.load().avllm_grad_sumsq(L.ptr(g), g.numel(), L.ptr(out), L.stream_ptr()))


def adamw_step(p, g, m, v, lr, step, sumsq=None, max_norm=0.0, beta1=0.9, beta2=0.95, eps=1e-8, wd=0.01, prescale=1.0, guard=None,
               skipped=None, state=None):
    """guard: device float (e.g. the step's loss_sum); a non-finite guard or sumsq makes the launch a no-op and bumps `skipped`.
    state: device avllm_step_state (uint8 tensor): lr and Adam's bias corrections are read from it instead of `lr` / `step`."""
    L.check(L.load().avllm_adamw_step(L.ptr(p), L.ptr(g), L.ptr(m), L.ptr(v), p.numel(), lr, beta1, beta2, eps, wd, step,
                                      L.ptr(sumsq), max_norm, prescale, L.ptr(guard), L.ptr(skipped), L.ptr(state), L.stream_ptr()))


def lora_dx_masked(Ts, ATs, seeds, r, p, R=None, out=None, seed_dev=None):
    """out = R + sum_j mask_j o (T_j . A_j)/(1-p): the adapters' input gradient under LoRA dropout in one pass (avllm_lora_dx_masked).
    Ts[j] [M,>=32] (zeros past r), ATs[j] [N,>=32] padded transposed A images; out may be R."""
    n = len(Ts)
    M, N = Ts[0].shape[0], ATs[0].shape[0]
    if out is None:
        out = torch.empty(M, N, device=Ts[0].device, dtype=Ts[0].dtype)
    arr = lambda ty, vals: (ty * n)(*vals)
    L.check(L.load().avllm_lora_dx_masked(arr(L.vp, [L.ptr(t) for t in Ts]), arr(L.i64, [_ld(t) for t in Ts]), arr(L.vp, [L.ptr(t) for t in ATs]),
                                          arr(L.i64, [_ld(t) for t in ATs]), arr(C.c_uint32, [s & 0xFFFFFFFF for s in seeds]), n, r, L.ptr(R),
                                          _ld(R) if R is not None else 0, L.ptr(out), _ld(out), M, N, p, seed_dev, L.dt_of(out), L.stream_ptr()))
    return out


def lora_rank3(As, Bs, outs, r, alpha=1.0, seeds=None, p=0.0, shared=False, seed_dev=None):
    """Batched rank-side products of up to 3 adapters (avllm_lora_rank3): outs[j][M,64] = alpha * A_j . Bs[j][r,K]^T (columns >= 16 zero).
    shared=True: every adapter reads As[0] through its own dropout mask (seeds[j], p)."""
    n = len(Bs)
    arr = lambda ty, vals: (ty * n)(*vals)
    Aa = [As[0]] * n if shared else As
    L.check(L.load().avllm_lora_rank3(arr(L.vp, [L.ptr(t) for t in Aa]), arr(L.i64, [_ld(t) for t in Aa]), arr(L.i32, [t.shape[1] for t in Aa]),
                                      arr(L.vp, [L.ptr(t) for t in Bs]), arr(L.i64, [_ld(t) for t in Bs]), arr(L.vp, [L.ptr(t) for t in outs]),
                                      arr(L.i64, [_ld(t) for t in outs]), arr(C.c_uint32, [(s & 0xFFFFFFFF) for s in (seeds or [0] * n)]), n,
                                      Aa[0].shape[0], r, alpha, p, seed_dev, int(shared), L.dt_of(Aa[0]), L.stream_ptr()))
    return outs


def gemm_tn_multi(big, smalls, outs, r, alpha=1.0, seeds=None, p=0.0, shared=False, cols=None, seed_dev=None):
    """Batched LoRA-gradient reductions over tokens (avllm_gemm_tn_multi), ACCUMULATING into the fp32 outs.
    shared=True: outs[j][r, NB] += alpha * smalls[j]^T . dropout_j(big); shared=False: outs[j][ncol_j, r] += alpha * big[:, cols[j]]^T . smalls[j]
    with cols = [(col0, ncol), ...] tiling big's columns."""
    n = len(smalls)
    arr = lambda ty, vals: (ty * n)(*vals)
    c0 = arr(L.i32, [c[0] for c in cols]) if cols else None
    nc = arr(L.i32, [c[1] for c in cols]) if cols else None
    L.check(L.load().avllm_gemm_tn_multi(L.ptr(big), _ld(big), big.shape[1], arr(L.vp, [L.ptr(t) for t in smalls]), arr(L.i64, [_ld(t) for t in smalls]),
                                         arr(L.vp, [L.ptr(t) for t in outs]), arr(L.i64, [_ld(t) for t in outs]), c0, nc,
                                         arr(C.c_uint32, [(s & 0xFFFFFFFF) for s in (seeds or [0] * n)]), n, r, big.shape[0], alpha, p, seed_dev,
                                         int(shared), L.dt_of(big), L.stream_ptr()))
    return outs


def mx_quantize(x, layout=0):
    """x [R,K] bf16/f32 -> (q uint8 [R,K] e4m3, scale image uint8 tensor): OCP-MX block scaling, 32 elements per E8M0 scale.
    layout 0 = activation side, 1 = weight side of avllm_gemm_f8."""
    lib = L.load()
    R, K = x.shape
    q = torch.empty(R, K, device=x.device, dtype=torch.uint8)
    s = torch.zeros(lib.avllm_mx_scale_bytes(R, K), device=x.device, dtype=torch.uint8)
    L.check(lib.avllm_mx_quantize(L.ptr(x), _ld(x), R, K, L.ptr(q), K, L.ptr(s), layout, L.dt_of(x), L.stream_ptr()))
    return q, s


def gemm_f8(Aq, As, Bq, Bs, out=None, bias=None, R=None, act=L.ACT_NONE, quantised_out=False):
    """out[M,N] (bf16) = act(A.B^T + bias) + R on the block-scaled fp8 matrix pipe; (Aq, As) / (Bq, Bs) from mx_quantize(layout 0 / 1).
    quantised_out=True: returns (codes uint8 [M,N], scale image) = the e4m3 block-scaled result, quantised in the epilogue (no bf16 copy)."""
    M, K = Aq.shape
    N = Bq.shape[0]
    d = L.GemmF8Desc()
    d.A, d.SA, d.B, d.SB, d.bias = L.ptr(Aq), L.ptr(As), L.ptr(Bq), L.ptr(Bs), L.ptr(bias)
    d.lda, d.ldb = _ld(Aq), _ld(Bq)
    d.M, d.N, d.K, d.act = M, N, K, act
    if quantised_out:
        q = torch.empty(M, N, device=Aq.device, dtype=torch.uint8)
        sc = torch.zeros(L.load().avllm_mx_scale_bytes(M, N), device=Aq.device, dtype=torch.uint8)
        d.Cq, d.SCq, d.ldcq = L.ptr(q), L.ptr(sc), N
        L.check(L.load().avllm_gemm_f8(C.byref(d), L.stream_ptr()))
        return q, sc
    if out is None:
        out = torch.empty(M, N, device=Aq.device, dtype=torch.bfloat16)
    d.C, d.ldc = L.ptr(out), _ld(out)
    if R is not None:
        d.R, d.ldr = L.ptr(R), _ld(R)
    L.check(L.load().avllm_gemm_f8(C.byref(d), L.stream_ptr()))
    return out


def norm_mxq(x, w, b=None, eps=1e-5, want_y=False, want_rstd=False):
    """LayerNorm (b given) / RMSNorm (b None) of bf16 rows, block-scaled to e4m3 in the same pass (avllm_norm_mxq):
    -> (codes uint8 [rows, d], scale image, y bf16 | None, rstd | None)."""
    rows, d = x.shape
    q = torch.empty(rows, d, device=x.device, dtype=torch.uint8)
    sc = torch.zeros(L.load().avllm_mx_scale_bytes(rows, d), device=x.device, dtype=torch.uint8)
    y = torch.empty_like(x) if want_y else None
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32) if want_rstd else None
    L.check(L.load().avllm_norm_mxq(L.ptr(x), L.ptr(w), L.ptr(b), L.ptr(y), L.ptr(rstd), L.ptr(q), d, L.ptr(sc), rows, d, eps, L.stream_ptr()))
    return q, sc, y, rstd


def dec_proj(A, W, mode=0, norm_w=None, eps=1e-5, R=None, out=None, out_f32=False, rope=None, kc=None, vc=None, pos=0, pos_dev=None, dq=0, dkv=0, hd=0,
             lora_t=None, lora_b=None, lora_r=0, lora_scale=0.0):
    """One projection of a decode token step (avllm_dec_proj): A [M<=16, K] bf16, W [rows, K] bf16.
    mode 0: out[M, rows] = rmsnorm?(A) . W^T (+ R);  mode 1: W = [gate; up], out[M, rows/2] = silu(gate) * up;
    mode 2: W = [q; k; v]: RoPE on q, k with `rope` [hd/2, 2]; q -> out[M, dq]; k, v -> kc / vc [M, Tmax, dkv] at row pos (+ *pos_dev)."""
    M, K = A.shape
    d = L.DecProjDesc()
    d.A, d.lda, d.W, d.ldw, d.M, d.K, d.mode = L.ptr(A), _ld(A), L.ptr(W), _ld(W), M, K, mode
    if norm_w is not None:
        d.norm_w, d.eps = L.ptr(norm_w), eps
    N = W.shape[0] // 2 if mode == 1 else W.shape[0]
    d.N = N
    if out is None:
        out = torch.empty(M, dq if mode == 2 else N, device=A.device, dtype=torch.float32 if out_f32 else torch.bfloat16)
    d.C, d.ldc, d.out_f32 = L.ptr(out), _ld(out), int(out.dtype == torch.float32)
    if R is not None:
        d.R, d.ldr = L.ptr(R), _ld(R)
    if mode == 2:
        d.dq, d.dkv, d.hd, d.rope, d.kc, d.vc, d.Tmax, d.pos = dq, dkv, hd, L.ptr(rope), L.ptr(kc), L.ptr(vc), kc.shape[1], pos
        d.pos_dev = L.ptr(pos_dev)
    if lora_t is not None:          # adapters: lora_t [M, >= 64 per module] f32 rank-side products, lora_b = padded B images [rows, 64] (one, or q / k / v)
        d.lora_t, d.ld_lora_t, d.lora_r, d.lora_scale = L.ptr(lora_t), _ld(lora_t), lora_r, lora_scale
        for j, b in enumerate(lora_b):
            d.lora_b[j] = L.ptr(b)
    L.check(L.load().avllm_dec_proj(C.byref(d), L.stream_ptr()))
    return out


def attention_decode(q, kc, vc, H, Tk, tk_dev=None, scale=None):
    """q [B, H*hd]; kc, vc [B, Tmax, Hkv*hd] -> [B, H*hd]: softmax(q.K^T * scale) V over cache rows [0, Tk (+ *tk_dev))."""
    B, d = q.shape
    hd = d // H
    Hkv = kc.shape[2] // hd
    o = torch.empty_like(q)
    L.check(L.load().avllm_attention_decode(L.ptr(q), _ld(q), L.ptr(kc), L.ptr(vc), L.ptr(o), _ld(o), B, H, hd, Tk, L.ptr(tk_dev), kc.shape[1],
                                            float(scale if scale is not None else hd ** -0.5), H // Hkv, L.dt_of(q), L.stream_ptr()))
    return o


def step_advance(state, base_lr, total_steps, warmup_steps=0, beta1=0.9, beta2=0.95, rank=0):
    """One-thread kernel: state.step += 1 and this step's lr / bias corrections / dropout seed (include/avllm.h avllm_step_state)."""
    sc = L.Schedule(base_lr, beta1, beta2, int(warmup_steps), int(total_steps), int(rank))
    L.check(L.load().avllm_step_advance(L.ptr(state), C.byref(sc), L.stream_ptr()))


def whisper_im2col1(mel, Kpad, dtype):
    B, n_mels, T = mel.shape
    cols = torch.empty(B * T, Kpad, device=mel.device, dtype=dtype)
    L.check(L.load().avllm_whisper_im2col1(L.ptr(mel), L.ptr(cols), B, n_mels, T, Kpad, L.dt_of(cols), L.stream_ptr()))
    return cols


def whisper_im2col2(h, B, T):
    d = h.shape[-1]
    cols = torch.empty(B * (T // 2), 3 * d, device=h.device, dtype=h.dtype)
    L.check(L.load().avllm_whisper_im2col2(L.ptr(h), L.ptr(cols), B, T, d, L.dt_of(h), L.stream_ptr()))
    return cols


def clip_patchify(frames, patch, Kpad, dtype):
    N, _, S, _ = frames.shape
    g = S // patch
    cols = torch.empty(N * g * g, Kpad, device=frames.device, dtype=dtype)
    L.check(L.load().avllm_clip_patchify(L.ptr(frames), L.ptr(cols), N, S, patch, Kpad, L.dt_of(cols), L.stream_ptr()))
    return cols


def dropout(x, seed, p):
    """y = x * keep/(1-p) with the library's counter-based mask (avllm_dropout)."""
    x = x.contiguous()
    y = torch.empty_like(x)
    L.check(L.load().avllm_dropout(L.ptr(x), L.ptr(y), x.numel() // x.shape[-1], x.shape[-1], seed & 0xFFFFFFFF, p, L.dt_of(x), L.stream_ptr()))
    return y
