"""Greedy-decode token step (clip_whisper_model.py:1337-1340 -> LlamaDecoderLayer.forward with q_len == 1): the fused projections
(avllm_dec_proj: RMSNorm in the A operand, RoPE + cache append / SwiGLU / residual in the epilogue), the single-pass cache attention and the
whole step, against plain torch fp32 arithmetic on the same bf16 inputs.  Floating point: bars are bf16's (tests/bars.py)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from avllm import ops  # noqa: E402
from bars import BF16_LOGITS_REL_L2, rel_l2  # noqa: E402
from test_ops_gpu import rnd  # noqa: E402

BF = torch.bfloat16


def rms(x, w, eps):
    x = x.float()
    return x * torch.rsqrt((x * x).mean(-1, keepdim=True) + eps) * w.float()


# K / 128 units dealt to the 8 waves: 1 (one wave works), 3, 9 (two groups on one wave), 23 (odd / even group counts mixed: 3,3,3,3,3,3,3,2),
# 32 (4096: four groups each), 86 (11008: 11 / 10)
@pytest.mark.parametrize("K", [128, 384, 1152, 2944, 4096, 11008])
@pytest.mark.parametrize("M", [1, 3, 4, 5, 8, 9, 16])
def test_dec_proj_plain_norm_residual(dev, M, K):
    N = 528
    A, W = rnd(M, K, dtype=BF, seed=1), rnd(N, K, dtype=BF, seed=2, scale=K ** -0.5)
    R, g = rnd(M, N, dtype=BF, seed=3), (1.0 + 0.1 * rnd(K, dtype=torch.float32, seed=4)).to(BF)
    ref = A.float() @ W.float().t()
    assert rel_l2(ops.dec_proj(A, W, out_f32=True), ref) < 2e-3                      # bf16 products, fp32 accumulate: only summation order differs
    out = R.clone()
    ops.dec_proj(A, W, R=out, out=out)                                               # in-place residual (o_proj / down_proj)
    assert rel_l2(out, ref + R.float()) < 6e-3
    refn = rms(A, g, 1e-5) @ W.float().t()
    assert rel_l2(ops.dec_proj(A, W, norm_w=g, eps=1e-5, out_f32=True), refn) < 6e-3  # the normed operand is rounded to bf16 once, like the unfused path's xn
    rows = rnd(4 * M, K, dtype=BF, seed=5)                                           # strided A rows
    assert rel_l2(ops.dec_proj(rows[3::4], W, out_f32=True), rows[3::4].float() @ W.float().t()) < 2e-3


@pytest.mark.parametrize("M,al", [(1, "2"), (3, "4"), (4, "2"), (8, "4")])
def test_dec_proj_activation_load_forms(dev, M, al):
    """M <= 8 / M <= 4 share one activation load between 2 / 4 K-steps (row rotate inside the MFMA operand); AVLLM_DEC_AL forces the
    wider forms on the same inputs: the sums are bit-identical (same products, same order)."""
    K = 2944
    A, W = rnd(M, K, dtype=BF, seed=1), rnd(528, K, dtype=BF, seed=2, scale=K ** -0.5)
    g = (1.0 + 0.1 * rnd(K, dtype=torch.float32, seed=4)).to(BF)
    want = ops.dec_proj(A, W, out_f32=True), ops.dec_proj(A, W, norm_w=g, eps=1e-5, out_f32=True)
    with ops.L.knob("DEC_AL", int(al)):
        got = ops.dec_proj(A, W, out_f32=True), ops.dec_proj(A, W, norm_w=g, eps=1e-5, out_f32=True)
    assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
    assert rel_l2(want[1], rms(A, g, 1e-5) @ W.float().t()) < 6e-3


@pytest.mark.parametrize("M,K,F", [(1, 256, 64), (8, 4096, 11008), (16, 1152, 520)])
def test_dec_proj_swiglu(dev, M, K, F):
    A, W = rnd(M, K, dtype=BF, seed=11), rnd(2 * F, K, dtype=BF, seed=12, scale=K ** -0.5)
    g = (1.0 + 0.1 * rnd(K, dtype=torch.float32, seed=13)).to(BF)
    xn = rms(A, g, 1e-6)
    gate, up = xn @ W[:F].float().t(), xn @ W[F:].float().t()
    out = ops.dec_proj(A, W, mode=1, norm_w=g, eps=1e-6)
    assert out.shape == (M, F) and rel_l2(out, torch.nn.functional.silu(gate) * up) < 8e-3


@pytest.mark.parametrize("M,heads,kvh,hd,K", [(2, 4, 4, 128, 512), (8, 4, 2, 64, 256), (16, 8, 2, 128, 1024), (3, 32, 32, 128, 4096)])
def test_dec_proj_qkv_rope_cache(dev, M, heads, kvh, hd, K):
    """q|k|v projection with RoPE (HF rotate_half pairing (i, i + hd/2)) and the k/v rows written into the cache at `pos` (+ *pos_dev);
    every other cache row must stay untouched."""
    dq, dkv, Tmax, pos = heads * hd, kvh * hd, 9, 5
    A, W = rnd(M, K, dtype=BF, seed=21), rnd(dq + 2 * dkv, K, dtype=BF, seed=22, scale=K ** -0.5)
    g = (1.0 + 0.1 * rnd(K, dtype=torch.float32, seed=23)).to(BF)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    ang = pos * inv
    rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous()
    y = rms(A, g, 1e-5) @ W.float().t()

    def rot(t, nh):
        t = t.view(M, nh, hd)
        a, b = t[..., : hd // 2], t[..., hd // 2:]
        return torch.cat([a * ang.cos() - b * ang.sin(), b * ang.cos() + a * ang.sin()], -1).reshape(M, nh * hd)

    for use_dev in (False, True):
        kc = torch.full((M, Tmax, dkv), 7.0, device=dev, dtype=BF)
        vc = torch.full((M, Tmax, dkv), -7.0, device=dev, dtype=BF)
        pd = torch.tensor([3], device=dev, dtype=torch.int32) if use_dev else None
        q = ops.dec_proj(A, W, mode=2, norm_w=g, eps=1e-5, rope=rope, kc=kc, vc=vc, pos=pos - (3 if use_dev else 0), pos_dev=pd, dq=dq, dkv=dkv, hd=hd)
        assert rel_l2(q, rot(y[:, :dq], heads)) < 8e-3
        assert rel_l2(kc[:, pos], rot(y[:, dq:dq + dkv], kvh)) < 8e-3
        assert rel_l2(vc[:, pos], y[:, dq + dkv:]) < 8e-3
        keep = [t for t in range(Tmax) if t != pos]
        assert (kc[:, keep] == 7.0).all() and (vc[:, keep] == -7.0).all()


@pytest.mark.parametrize("M,r", [(1, 16), (5, 8), (8, 16), (16, 4)])
def test_dec_proj_adapter_side_term_plain_and_qkv(dev, M, r):
    """peft lora.Linear on the fused token step: y = W x + scale * B (A x).  The rank-side products A x come from a mode-0 launch over the
    padded A images (f32 out), the B side rides in the projection's epilogue -- BEFORE RoPE in the q|k|v form.  Against fp32 torch on the
    same bf16 operands; a wrong module offset, a missing scale or a term added after the rotation each miss the bar by far."""
    K, scale = 1152, 2.0
    g = (1.0 + 0.1 * rnd(K, dtype=torch.float32, seed=43)).to(BF)
    A = rnd(M, K, dtype=BF, seed=41)
    # ---- plain (o_proj form: no norm, residual)
    N = 528
    W = rnd(N, K, dtype=BF, seed=42, scale=K ** -0.5)
    Ap = torch.zeros(64, K, device=dev, dtype=BF); Ap[:r] = rnd(r, K, dtype=BF, seed=44, scale=K ** -0.5)
    Bp = torch.zeros(N, 64, device=dev, dtype=BF); Bp[:, :r] = rnd(N, r, dtype=BF, seed=45, scale=0.3)
    R = rnd(M, N, dtype=BF, seed=46)
    t = torch.zeros(M, 256, device=dev, dtype=torch.float32)
    ops.dec_proj(A, Ap[:16], out=t[:, 192:208], out_f32=True)
    ref_t = A.float() @ Ap[:16].float().t()
    assert rel_l2(t[:, 192:208], ref_t) < 2e-3
    out = ops.dec_proj(A, W, R=R, lora_t=t[:, 192:], lora_b=[Bp], lora_r=r, lora_scale=scale, out_f32=True)
    ref = A.float() @ W.float().t() + R.float() + scale * (ref_t[:, :r] @ Bp[:, :r].float().t())
    assert rel_l2(out, ref) < 3e-3
    assert rel_l2(ops.dec_proj(A, W, R=R, out_f32=True), ref) > 0.05              # the side term is not small here
    # ---- q|k|v with norm + RoPE + cache (grouped-query widths)
    heads, kvh, hd, Tmax, pos = 4, 2, 64, 6, 3
    dq, dkv = heads * hd, kvh * hd
    Wq = rnd(dq + 2 * dkv, K, dtype=BF, seed=47, scale=K ** -0.5)
    A3 = torch.zeros(192, K, device=dev, dtype=BF)
    Bs = []
    for j, rows in enumerate((dq, dkv, dkv)):
        A3[64 * j:64 * j + r] = rnd(r, K, dtype=BF, seed=50 + j, scale=K ** -0.5)
        b = torch.zeros(rows, 64, device=dev, dtype=BF); b[:, :r] = rnd(rows, r, dtype=BF, seed=60 + j, scale=0.3)
        Bs.append(b)
    ops.dec_proj(A, A3, norm_w=g, eps=1e-5, out=t[:, :192], out_f32=True)
    xn = rms(A, g, 1e-5)
    y = xn @ Wq.float().t()
    off = 0
    for j, rows in enumerate((dq, dkv, dkv)):
        y[:, off:off + rows] += scale * ((xn @ A3[64 * j:64 * j + r].float().t()) @ Bs[j][:, :r].float().t())
        off += rows
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, device=dev).float() / hd))
    ang = pos * inv
    rope = torch.stack([ang.cos(), ang.sin()], -1).contiguous()

    def rot(t_, nh):
        t_ = t_.view(M, nh, hd)
        a, b = t_[..., : hd // 2], t_[..., hd // 2:]
        return torch.cat([a * ang.cos() - b * ang.sin(), b * ang.cos() + a * ang.sin()], -1).reshape(M, nh * hd)

    kc = torch.zeros(M, Tmax, dkv, device=dev, dtype=BF); vc = torch.zeros_like(kc)
    q = ops.dec_proj(A, Wq, mode=2, norm_w=g, eps=1e-5, rope=rope, kc=kc, vc=vc, pos=pos, dq=dq, dkv=dkv, hd=hd,
                     lora_t=t, lora_b=Bs, lora_r=r, lora_scale=scale)
    assert rel_l2(q, rot(y[:, :dq], heads)) < 8e-3
    assert rel_l2(kc[:, pos], rot(y[:, dq:dq + dkv], kvh)) < 8e-3
    assert rel_l2(vc[:, pos], y[:, dq + dkv:]) < 8e-3


def test_dec_proj_refuses_what_it_cannot_do(dev):
    A, W = rnd(17, 256, dtype=BF, seed=1), rnd(64, 256, dtype=BF, seed=2)
    with pytest.raises(ValueError):
        ops.dec_proj(A, W)                                       # 17 rows
    with pytest.raises(ValueError):
        ops.dec_proj(A[:4, :192], W[:, :192])                    # K % 128
    with pytest.raises(ValueError):
        ops.dec_proj(A[:4], W[:60])                              # N % 16


@pytest.mark.parametrize("B,H,Hkv,hd,Tk,dtype", [(2, 4, 4, 128, 1, BF), (8, 32, 32, 128, 300, BF), (3, 8, 2, 64, 700, BF), (16, 128, 16, 64, 130, BF),
                                                   (2, 4, 2, 128, 257, torch.float32)])
def test_attention_decode_single_pass(dev, B, H, Hkv, hd, Tk, dtype):
    """One pass with an online softmax; (8,32,..,300) is the decode bench's shape (two trips, the second ragged), (16,128,..) takes the 4-wave
    instantiation (B*H > 1024), Tk = 1 leaves most row groups empty (-inf merge)."""
    Tmax = Tk + 5
    q = rnd(B, H * hd, dtype=dtype, seed=31)
    kc, vc = rnd(B, Tmax, Hkv * hd, dtype=dtype, seed=32), rnd(B, Tmax, Hkv * hd, dtype=dtype, seed=33)
    kc[:, Tk:] = float("nan")                                    # rows past Tk must never be read into the result
    vc[:, Tk:] = float("nan")
    rep = H // Hkv
    qf = q.float().view(B, H, 1, hd)
    kf = kc[:, :Tk].float().view(B, Tk, Hkv, hd).permute(0, 2, 1, 3).repeat_interleave(rep, 1)
    vf = vc[:, :Tk].float().view(B, Tk, Hkv, hd).permute(0, 2, 1, 3).repeat_interleave(rep, 1)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, -1) @ vf).reshape(B, H * hd)
    tol = 1e-5 if dtype == torch.float32 else 6e-3
    assert rel_l2(ops.attention_decode(q, kc, vc, H, Tk), ref) < tol
    td = torch.tensor([Tk - 1], device=dev, dtype=torch.int32)
    assert rel_l2(ops.attention_decode(q, kc, vc, H, 1, tk_dev=td), ref) < tol


def _engine(dev, hidden, heads, kvh, layers, ffn, vocab, seed=5):
    from avllm.arch import LlamaCfg
    from avllm.engine import LlamaEngine
    cfg = LlamaCfg(hidden, heads, layers, ffn, vocab)
    cfg.kv_heads = kvh
    g = torch.Generator().manual_seed(seed)
    dkv = kvh * (hidden // heads)
    sd = {"model.embed_tokens.weight": torch.randn(vocab, hidden, generator=g) * 0.5, "model.norm.weight": 1 + 0.1 * torch.randn(hidden, generator=g),
          "lm_head.weight": torch.randn(vocab, hidden, generator=g) * hidden ** -0.5}
    for i in range(layers):
        p = f"model.layers.{i}."
        for nm, (o, k) in {"self_attn.q_proj": (hidden, hidden), "self_attn.k_proj": (dkv, hidden), "self_attn.v_proj": (dkv, hidden),
                           "self_attn.o_proj": (hidden, hidden), "mlp.gate_proj": (ffn, hidden), "mlp.up_proj": (ffn, hidden),
                           "mlp.down_proj": (hidden, ffn)}.items():
            sd[p + nm + ".weight"] = torch.randn(o, k, generator=g) * k ** -0.5
        sd[p + "input_layernorm.weight"] = 1 + 0.1 * torch.randn(hidden, generator=g)
        sd[p + "post_attention_layernorm.weight"] = 1 + 0.1 * torch.randn(hidden, generator=g)
    return LlamaEngine(sd, cfg, None, None, dtype=BF, device=dev, training=False), cfg


@pytest.mark.parametrize("hidden,heads,kvh,B", [(256, 2, 2, 3), (512, 8, 2, 8)])
def test_token_step_fused_matches_general_path_and_prefill(dev, hidden, heads, kvh, B):
    """The 5-launch token step against (a) the general 10-launch path on the same cache (AVLLM_DECODE_FUSED=0) and (b) a prefill that
    recomputes every position: logits of position S from `prefill(S) + step` vs `prefill(S + 1)`; then three more steps with the
    position taken from device memory (what a captured step replays)."""
    eng, cfg = _engine(dev, hidden, heads, kvh, 2, 384, 256)
    S, new = 37, 4
    g = torch.Generator(device=dev).manual_seed(3)
    ids = torch.randint(0, cfg.vocab, (B, S + new), generator=g, device=dev)
    x = ops.embedding(eng.embed, ids.reshape(-1).contiguous()).view(B, S + new, hidden)
    ref_all = eng.prefill(x, *eng.alloc_cache(B, S + new), all_logits=True)[1].float()      # [B, S+new, vocab]: teacher-forced logits

    def run(fused, device_pos):
        with ops.L.knob("DECODE_FUSED", 1 if fused else 0):
            kc, vc = eng.alloc_cache(B, S + new + 2)
            eng.prefill(x[:, :S].contiguous(), kc, vc)
            pd = torch.zeros(1, device=dev, dtype=torch.int32) if device_pos else None
            outs = []
            for t in range(new):
                if device_pos:
                    outs.append(eng.decode_step(ids[:, S + t].contiguous(), S, kc, vc, pos_dev=pd).clone())
                    ops.L.check(ops.L.load().avllm_pos_advance(ops.L.ptr(pd), 1, ops.L.stream_ptr()))
                else:
                    outs.append(eng.decode_step(ids[:, S + t].contiguous(), S + t, kc, vc).clone())
            return torch.stack(outs, 1)

    general, fused, fused_dev = run(False, False), run(True, False), run(True, True)
    want = ref_all[:, S:S + new]
    assert rel_l2(general, want) < BF16_LOGITS_REL_L2
    assert rel_l2(fused, want) < BF16_LOGITS_REL_L2
    assert rel_l2(fused, general) < BF16_LOGITS_REL_L2
    assert torch.equal(fused, fused_dev)                          # same kernels, the position merely comes from memory
    with pytest.raises(ValueError), ops.L.knob("DECODE_FUSED", 0):
        eng.decode_step(ids[:, 0].contiguous(), 1, *eng.alloc_cache(B, 4), pos_dev=torch.zeros(1, device=dev, dtype=torch.int32))


@pytest.mark.parametrize("with_lora", [False, True])
def test_generate_bf16_fused_token_step_vs_oracle_greedy(dev, golden_dir, with_lora):
    """generate() in bf16 -- the arithmetic bench.py's decode{} times: prefill + the fused token step (asserted: avllm_llama_decode_is_fused) --
    against the ORACLE's greedy search on the reference-pinned tiny model (clip_whisper_model.py:1337-1340 -> HF greedy).  bf16 cannot promise
    the fp32 token where the top-2 logits are closer than its own logit error, so a row is followed for as long as the two searches agree:
    at every step where the oracle's top-2 margin is at least 2 x the bf16 logit bar the token MUST be the oracle's; at a near-tie either
    choice is accepted, and the row ends there if bf16 took the other one (the two searches then see different prefixes).  With and
    without adapters (decode.py runs without; the trainer's eval and `decode.py --load_lora` with)."""
    import numpy as np
    from bars import BF16_LOGIT_MAX_ABS
    from oracle import avsr_oracle as O
    from oracle import weights as Wt
    from test_model_gpu import make_model
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    oc = Wt.tiny()
    W = Wt.all_weights(oc, int(g["seed"]), lora_b_std=0.05)
    if not with_lora:
        W = {k: v for k, v in W.items() if k != "lora"}
    audio, video, _, _ = Wt.synthetic_batch(oc, 4, int(g["frames"]), seed=int(g["batch_seed"]))
    new = 24
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device="cuda:0", use_lora=with_lora, lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=256, config=cfg,
                         weights=W, precision="bf16").eval()
    assert m.llm_engine.decode_is_fused(4), "the fused token step must take this model (bf16, B <= 16" + (", adapters" if with_lora else "") + ")"
    ids = m.generate(audio=audio.to(dev), video=video.to(dev), max_new_tokens=new).cpu()
    oc.max_seq_len = 256
    ref, margins = O.generate(W, oc, audio, video, None, max_new_tokens=new, eos_token_id=m.eos_token_id, return_margins=True)
    thr = 2.0 * BF16_LOGIT_MAX_ABS
    compared = 0
    for b in range(ref.shape[0]):
        for t in range(min(ref.shape[1], ids.shape[1])):
            same = int(ids[b, t]) == int(ref[b, t])
            if float(margins[b, t]) >= thr:
                assert same, (b, t, ids[b].tolist(), ref[b].tolist(), margins[b].tolist())      # a clear decision of the oracle: bf16 must make it too
                compared += 1
            elif not same:
                break                                            # a near-tie went the other way: from here on the two searches see different prefixes
    assert compared >= 8, (compared, margins.tolist())           # not vacuous: at least 8 tokens were held to the oracle's choice
