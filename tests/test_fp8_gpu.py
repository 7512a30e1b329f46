"""Block-scaled fp8 (BASELINE config 5's "fp8 MFMA"): the quantiser bit for bit against the CPU restatement of the OCP MX rule
(oracle/mxfp8.py), the fp8 GEMM against fp32 products of the fake-quantised operands.  The reference has no fp8 mode
(clip_whisper_model.py:164), so this arithmetic is "parity unpinned" against it by construction; what is pinned is that the HIP path
computes exactly the published MX rule, and (test_model_fp8) how far that moves the model's outputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from bars import rel_l2  # noqa: E402
from oracle import mxfp8 as MX  # noqa: E402
from test_ops_gpu import rnd  # noqa: E402


def _image_exponents(simg, layout, R, K):
    word, byte = MX.scale_image_index(layout, R, K)
    w = simg.cpu().view(torch.int32)[word.reshape(-1)].reshape(word.shape)
    return ((w >> (8 * byte)) & 0xFF) - 127


@pytest.mark.parametrize("layout", [0, 1])
@pytest.mark.parametrize("R,K,dtype", [(256, 256, torch.bfloat16), (300, 768, torch.bfloat16), (64, 128, torch.float32), (1000, 1024, torch.bfloat16),
                                       (17, 4096, torch.bfloat16)])
def test_mx_quantizer_bit_exact(dev, layout, R, K, dtype):
    """Codes and E8M0 exponents of avllm_mx_quantize == the MX rule restated on the CPU, for both scale-image layouts, ragged row counts,
    all-zero blocks, outliers (saturation to 448) and tiny values."""
    from avllm import ops
    x = rnd(R, K, dtype=dtype, seed=5)
    x[0, :32] = 0                                   # an all-zero block
    x[1, 40] = 3.0e4                                # an outlier dominating its block
    x[2, 64:96] *= 1e-6                             # a tiny block
    x[3, 100] = 448.0 * 2.0 ** 3 * 1.07             # lands above 448 after scaling: saturates
    q, s = ops.mx_quantize(x, layout)
    codes, e = MX.quantize(x.cpu())
    assert torch.equal(q.cpu(), codes), int((q.cpu() != codes).sum())
    assert torch.equal(_image_exponents(s, layout, R, K), e), "scale image"
    deq = MX.dequantize(q.cpu(), e)
    err = (deq - x.float().cpu()).abs()
    blk = x.float().cpu().abs().reshape(R, K // 32, 32).amax(-1).repeat_interleave(32, dim=1)
    # e4m3 keeps 3 mantissa bits: half an ulp of the block's top binade is amax * 2^-4; a block maximum in (448, 512) * 2^e saturates to 448
    # (the MX rule clamps: the exponent is floor(log2(amax)) - 8 although e4m3 tops out at 1.75 * 2^8), which costs up to amax / 8
    assert bool((err <= blk * 2.0 ** -3 + 1e-30).all())


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (256, 256, 512), (300, 200, 384), (4096, 4096, 1024), (130, 1000, 256), (4096, 4096, 256), (5000, 2304, 768),
                                   (4100, 4104, 384), (20000, 768, 3072)])
def test_gemm_f8_matches_fake_quant_product(dev, M, N, K):
    from avllm import lib as L
    from avllm import ops
    A = rnd(M, K, dtype=torch.bfloat16, seed=11)
    W = rnd(N, K, dtype=torch.bfloat16, seed=12, scale=K ** -0.5)
    bias = rnd(N, dtype=torch.bfloat16, seed=13)
    Rr = rnd(M, N, dtype=torch.bfloat16, seed=14)
    Aq, As = ops.mx_quantize(A, 0)
    Wq, Ws = ops.mx_quantize(W, 1)
    ref = MX.fake_quant(A.float().cpu()) @ MX.fake_quant(W.float().cpu()).T
    out = ops.gemm_f8(Aq, As, Wq, Ws).float().cpu()
    assert rel_l2(out, ref) < 3e-3, rel_l2(out, ref)                          # bf16 rounding of the output only
    err = (out - ref.to(torch.bfloat16).float()).abs()
    assert int((err > ref.abs() * 2.0 ** -7 + 4e-3).sum()) == 0             # one bf16 ulp + the fp32 summation-order slack on cancelling sums
    full = ops.gemm_f8(Aq, As, Wq, Ws, bias=bias, R=Rr, act=L.ACT_QUICK_GELU).float().cpu()
    z = ref + bias.float().cpu()
    ref2 = z * torch.sigmoid(1.702 * z) + Rr.float().cpu()
    assert rel_l2(full, ref2) < 4e-3
    # how far fp8 is from the unquantised product (information: ~2^-4 / sqrt(3) per operand, averaged over K)
    exact = A.float().cpu() @ W.float().cpu().T
    assert rel_l2(ref, exact) < 6e-2
    if M > 128:        # the same call through the reference-grade kernel (fragments straight from global memory) must agree with the fast one
        import os, subprocess, sys
        from avllm import lib as Lb
        # the dispatcher reads AVLLM_F8_FAST once per process: ask a child for the reference-grade result of the same seeded operands
        code = ("import sys,torch; sys.path[:0]=%r; from avllm import ops; from test_ops_gpu import rnd; "
                "A=rnd(%d,%d,dtype=torch.bfloat16,seed=11); W=rnd(%d,%d,dtype=torch.bfloat16,seed=12,scale=%d**-0.5); "
                "o=ops.gemm_f8(*ops.mx_quantize(A,0),*ops.mx_quantize(W,1)); torch.save(o.cpu(), %r)")
        path = f"/tmp/f8ref_{M}_{N}_{K}.pt"
        r = subprocess.run([sys.executable, "-c", code % ([p for p in sys.path if p], M, K, N, K, K, path)], env=dict(os.environ, AVLLM_F8_FAST="0"),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        slow = torch.load(path).float()
        assert rel_l2(out, slow) < 1e-3 and float((out - slow).abs().max()) <= float(slow.abs().max()) * 2.0 ** -6


def test_gemm_f8_identity_asymmetric(dev):
    """A = I (exactly representable) against an asymmetric integer B: catches a transposed or permuted output map, and a wrong scale byte."""
    from avllm import ops
    n = 256
    A = torch.eye(n, device=dev, dtype=torch.bfloat16) * 4.0
    B = ((torch.arange(n * n, device=dev).reshape(n, n) % 13) - 6).to(torch.bfloat16)          # small integers: exact in e4m3
    B[:, 128:] *= 16                                                                          # different block scales along K
    Aq, As = ops.mx_quantize(A, 0)
    Bq, Bs = ops.mx_quantize(B, 1)
    out = ops.gemm_f8(Aq, As, Bq, Bs).float()
    assert torch.equal(out, 4.0 * B.float().t())


def test_model_fp8_train_step_vs_fake_quant_oracle(dev, golden_dir):
    """precision="fp8" end to end on the golden tiny model: encoders (fp8 projections) and one LoRA train step (fp8 frozen projections in the
    forward, bf16 backward) against the oracle run with the same MX fake-quantisation; bars in tests/bars.py.  Also: the step's distance from
    the unquantised oracle (what fp8 costs), and an fp8 trainer step that actually moves the LoRA parameters."""
    import bars as Bar
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer
    from oracle import avsr_oracle as O
    from oracle import weights as Wt
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    oc = Wt.tiny()
    W = Wt.all_weights(oc, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, _ = Wt.synthetic_batch(oc, 2, int(g["frames"]), seed=int(g["batch_seed"]))
    prompt = torch.from_numpy(g["prompt"])
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=512, config=cfg, weights=W,
                         precision="fp8").train()
    assert m.fp8 and m.llm_engine.desc.fp8 == 1 and m.whisper_engine.desc.fp8 == 1 and m.clip_engine.desc.fp8 == 1
    fr = video.reshape(-1, 3, oc.clip.image, oc.clip.image)
    with torch.no_grad(), O.fp8_mode():
        ref_w = O.whisper_encoder(W["whisper"], oc.whisper, audio)
        ref_c = O.clip_vision_cls(W["clip"], oc.clip, fr)
    assert rel_l2(m.whisper_engine.forward(audio.to(dev)).float().cpu(), ref_w) < Bar.FP8_ENC_REL_L2
    assert rel_l2(m.clip_engine.forward(fr.to(dev)).float().cpu(), ref_c) < Bar.FP8_ENC_REL_L2
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    m.lora_param.grad = None
    out["loss"].backward()
    with O.fp8_mode():
        ol, ologits, og = O.train_step_grads(W, oc, audio, video, prompt, labels)
    logits = out["logits"].float().cpu()
    assert rel_l2(logits, ologits) < Bar.FP8_LOGITS_REL_L2, rel_l2(logits, ologits)
    assert abs(float(out["loss"].detach()) - float(ol)) < Bar.FP8_LOSS_ABS
    gv = {k: v.cpu() for k, v in m.llm_engine.lora_views(m.lora_param.grad).items()}
    keys = sorted(gv)
    whole = rel_l2(torch.cat([gv[k].flatten() for k in keys]), torch.cat([og[k].flatten() for k in keys]))
    assert whole < Bar.FP8_GRAD_REL_L2, whole
    # what the quantisation costs against the unquantised reference arithmetic (the reference-generated golden logits)
    gold = torch.from_numpy(g["train_logits"])
    cost = rel_l2(logits, gold)
    assert 1e-3 < cost < Bar.FP8_VS_UNQUANTISED_REL_L2, cost
    # and the trainer steps in fp8 (graph-replayed like any other precision)
    tr = ClipWhisperTrainer(m, learning_rate=1e-3, total_steps=10, max_epochs=1)
    p0 = m.llm_engine.lora_p.clone()
    losses = [float(tr.train_step(audio.to(dev), video.to(dev), labels.to(dev), prompt.to(dev))) for _ in range(3)]
    assert all(np.isfinite(losses)) and not torch.equal(p0, m.llm_engine.lora_p) and losses[-1] < losses[0] + 0.05


@pytest.mark.parametrize("T,H,B", [(197, 12, 5), (257, 16, 7), (50, 4, 3), (272, 2, 2)])
def test_short_attention_with_mx_quantised_output_is_bit_identical_to_the_separate_pass(dev, T, H, B):
    """The fp8 encoders take the attention output as e4m3 codes + scale image straight from the attention kernel's epilogue
    (avllm_attention_fwd_mxq); it must equal avllm_mx_quantize applied to the bf16 attention output BIT FOR BIT -- codes, and every scale byte
    (rows of different frames share scale words: the byte addressing by global row is what this pins)."""
    from avllm import ops
    d = H * 64
    qkv = rnd(B * T, 3 * d, dtype=torch.bfloat16, seed=70 + T)
    o, _ = ops.attention_fwd(qkv, B, T, H, 64, causal=False, want_lse=False)
    q_ref, s_ref = ops.mx_quantize(o, 0)
    q, s = ops.attention_fwd_mxq(qkv, B, T, H, 64)
    assert torch.equal(q, q_ref), int((q != q_ref).sum())
    assert torch.equal(s, s_ref), int((s != s_ref).sum())
    assert int(s.max()) > 100 and int((q != 0).sum()) > q.numel() // 2          # not vacuous


def test_config5_family_whole_fp8_step_vs_fake_quant_oracle__mx_rule_parity_unpinned(dev):
    """BASELINE configs[4] AS ONE MODEL at test size and realistic width: a 128-mel Whisper (large-v3 layout), a patch-14 CLIP on 224-pixel
    frames (257 tokens per frame: the ViT-L/14 sequence), a grouped-query LLM (Mistral layout: 8 query / 2 key-value heads), every width 1024,
    precision="fp8" (block-scaled e4m3 on the frozen forward projections of all three towers), LoRA on q/k/v/o.  One whole train step --
    encoders -> connectors -> fusion -> LLM forward + loss -> backward -- against the fp32 oracle run with the SAME MX fake-quantisation,
    held to the fp8 bars of tests/bars.py (width does not shrink them: derivation there).  "parity unpinned": no reference implementation of the MX rule exists (the reference has no
    fp8 path: clip_whisper_model.py:164 only knows use_fp16); oracle/mxfp8.py restates the OCP MX definition and is pinned against the
    hardware's scale/operand layout only (tools/ubench/mfma_scale_probe.hip).  Reference for the model family: :1074 (80-bin guard, lifted),
    :966-970 (LoRA target selection)."""
    import bars as Bar
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer
    from oracle import avsr_oracle as O
    from oracle import weights as Wt
    oc = Wt.tiny()
    oc.whisper = Wt.WhisperCfg(d_model=1024, heads=16, layers=2, ffn=2048, n_mels=128)
    oc.clip = Wt.ClipCfg(hidden=1024, heads=16, layers=2, mlp=4096, image=224, patch=14)
    oc.llama = Wt.LlamaCfg(hidden=1024, heads=8, layers=2, ffn=3584, vocab=512, kv_heads=2)
    assert oc.clip.tokens == 257
    W = Wt.all_weights(oc, 31, lora_b_std=0.05)
    audio, video, labels, prompt = Wt.synthetic_batch(oc, 2, 4, seed=13)
    assert audio.shape[1] == 128
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=512, config=cfg, weights=W,
                         precision="fp8").train()
    assert m.fp8 and m.llm_engine.desc.fp8 == 1 and m.whisper_engine.desc.fp8 == 1 and m.clip_engine.desc.fp8 == 1 and m.llm_engine.desc.kv_heads == 2
    fr = video.reshape(-1, 3, oc.clip.image, oc.clip.image)
    with torch.no_grad(), O.fp8_mode():
        ref_w = O.whisper_encoder(W["whisper"], oc.whisper, audio)
        ref_c = O.clip_vision_cls(W["clip"], oc.clip, fr)
    e_w = rel_l2(m.whisper_engine.forward(audio.to(dev)).float().cpu(), ref_w)
    e_c = rel_l2(m.clip_engine.forward(fr.to(dev)).float().cpu(), ref_c)
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    m.lora_param.grad = None
    out["loss"].backward()
    with O.fp8_mode():
        ol, ologits, og = O.train_step_grads(W, oc, audio, video, prompt, labels)
    logits = out["logits"].float().cpu()
    gv = {k: v.cpu() for k, v in m.llm_engine.lora_views(m.lora_param.grad).items()}
    keys = sorted(gv)
    e_l = rel_l2(logits, ologits)
    e_g = rel_l2(torch.cat([gv[k].flatten() for k in keys]), torch.cat([og[k].flatten() for k in keys]))
    print(f"config-5 family, fp8: whisper {e_w:.4f} clip {e_c:.4f} logits {e_l:.4f} grads {e_g:.4f} loss {float(out['loss'].detach()):.5f} vs {float(ol):.5f}")
    assert e_w < Bar.FP8_ENC_REL_L2 and e_c < Bar.FP8_ENC_REL_L2, (e_w, e_c)
    assert e_l < Bar.FP8_LOGITS_REL_L2, e_l
    assert abs(float(out["loss"].detach()) - float(ol)) < Bar.FP8_LOSS_ABS
    assert e_g < Bar.FP8_GRAD_REL_L2, e_g
    # what the quantisation costs against the unquantised oracle (information; loosely bounded)
    ul, ulogits, _ = O.train_step_grads(W, oc, audio, video, prompt, labels)
    cost = rel_l2(logits, ulogits)
    assert 1e-3 < cost < Bar.FP8_VS_UNQUANTISED_REL_L2, cost
    # and the same model trains through the trainer (hipGraph replay) in fp8
    tr = ClipWhisperTrainer(m, learning_rate=1e-3, total_steps=10, max_epochs=1)
    p0 = m.llm_engine.lora_p.clone()
    losses = [float(tr.train_step(audio.to(dev), video.to(dev), labels.to(dev), prompt.to(dev))) for _ in range(3)]
    assert all(np.isfinite(losses)) and not torch.equal(p0, m.llm_engine.lora_p) and losses[-1] < losses[0] + 0.05


@pytest.mark.parametrize("M,N,K,act", [(4096, 4096, 1024, "quick_gelu"), (5000, 3072, 768, "gelu"), (4100, 4096, 384, "none")])
def test_gemm_f8_quantised_output(dev, M, N, K, act):
    """fc1 -> fc2 without a bf16 round trip: the persistent fp8 GEMM block-scales act(A.W^T + b) to e4m3 in its epilogue.  Against the MX rule
    (oracle/mxfp8.py) applied to the fp32 reference result: exponents equal except where the block maximum sits on a power-of-two boundary
    within summation-order noise, codes within one e4m3 step, and the (codes, scale image) pair is accepted by the next GEMM as its A operand."""
    from avllm import lib as L
    from avllm import ops
    A = rnd(M, K, dtype=torch.bfloat16, seed=21)
    W = rnd(N, K, dtype=torch.bfloat16, seed=22, scale=K ** -0.5)
    bias = rnd(N, dtype=torch.bfloat16, seed=23)
    Aq, As = ops.mx_quantize(A, 0)
    Wq, Ws = ops.mx_quantize(W, 1)
    code = {"none": L.ACT_NONE, "gelu": L.ACT_GELU, "quick_gelu": L.ACT_QUICK_GELU}[act]
    q, sc = ops.gemm_f8(Aq, As, Wq, Ws, bias=bias, act=code, quantised_out=True)
    z = MX.fake_quant(A.float().cpu()) @ MX.fake_quant(W.float().cpu()).T + bias.float().cpu()
    ref = {"none": lambda t: t, "gelu": torch.nn.functional.gelu, "quick_gelu": lambda t: t * torch.sigmoid(1.702 * t)}[act](z)
    rc, re = MX.quantize(ref)
    e = _image_exponents(sc, 0, M, N)
    assert float((e != re).float().mean()) < 2e-3                              # a block maximum within ~1e-6 of a power of two may land on either side
    deq = MX.dequantize(q.cpu(), e)
    blk = ref.abs().reshape(M, N // 32, 32).amax(-1).repeat_interleave(32, dim=1)
    same = (e == re).repeat_interleave(32, dim=1)
    err = (deq - ref).abs()
    assert bool((err[same] <= blk[same] * 2.0 ** -3 + 1e-6).all())             # the quantiser's own bound (test_mx_quantizer_bit_exact)
    assert bool((err[~same] <= blk[~same] * 2.0 ** -2).all())                  # boundary blocks: one exponent off = one more bit of step or a clamp
    assert rel_l2(deq, MX.dequantize(rc, re)) < 5e-3                           # and the same numbers as the rule gives, up to boundary cases
    # the pair feeds the next projection directly: equal (bf16 output) to quantising a bf16 copy first, up to the boundary cases
    W2 = rnd(256, N, dtype=torch.bfloat16, seed=24, scale=N ** -0.5)
    W2q, W2s = ops.mx_quantize(W2, 1)
    direct = ops.gemm_f8(q, sc, W2q, W2s).float()
    two_step = ops.gemm_f8(*ops.mx_quantize(ops.gemm_f8(Aq, As, Wq, Ws, bias=bias, act=code), 0), W2q, W2s).float()
    assert rel_l2(direct, two_step) < 2e-2
    with pytest.raises(Exception):
        ops.gemm_f8(Aq[:64], As, Wq, Ws, quantised_out=True)                   # too small for the persistent kernel: refused, not silently slow


@pytest.mark.parametrize("rows,d,rms", [(1000, 1024, False), (300, 1280, False), (257, 768, False), (130, 4096, True), (64, 8192, True), (5, 128, False)])
def test_norm_mxq_one_pass(dev, rows, d, rms):
    """LayerNorm / RMSNorm with the result block-scaled to e4m3 in the same pass (avllm_norm_mxq): against the MX rule applied to the fp32
    normalisation of the same bf16 rows; the optional bf16 copy and 1/rms equal the plain norm kernels'."""
    from avllm import ops
    x = rnd(rows, d, dtype=torch.bfloat16, seed=31) * 3 + 0.5
    w = (1 + 0.1 * rnd(d, dtype=torch.float32, seed=32)).to(torch.bfloat16)
    b = None if rms else rnd(d, dtype=torch.bfloat16, seed=33)
    q, sc, y, rstd = ops.norm_mxq(x, w, b, eps=1e-5, want_y=True, want_rstd=rms)
    xf = x.float().cpu()
    if rms:
        r = torch.rsqrt((xf * xf).mean(-1, keepdim=True) + 1e-5)
        ref = w.float().cpu() * (xf * r)
        assert rel_l2(rstd.cpu(), r[:, 0]) < 1e-6
        assert rel_l2(y, ops.rmsnorm_fwd(x, w, 1e-5)[0]) < 1e-3      # the same formula; the plain kernel splits a long row over 4 waves (other summation order)
    else:
        ref = torch.nn.functional.layer_norm(xf, (d,), w.float().cpu(), b.float().cpu(), 1e-5)
        assert rel_l2(y, ops.layernorm(x, w, b, 1e-5)) < 1e-3
    rc, re = MX.quantize(ref)
    e = _image_exponents(sc, 0, rows, d)
    assert float((e != re).float().mean()) < 2e-3
    deq = MX.dequantize(q.cpu(), e)
    blk = ref.abs().reshape(rows, d // 32, 32).amax(-1).repeat_interleave(32, dim=1)
    same = (e == re).repeat_interleave(32, dim=1)
    err = (deq - ref).abs()
    assert bool((err[same] <= blk[same] * 2.0 ** -3 + 1e-6).all()) and bool((err[~same] <= blk[~same] * 2.0 ** -2 + 1e-6).all())
    assert float((q.cpu() != rc).float().mean()) < 5e-3          # codes equal except where fp32 summation order moves a value across a rounding boundary


def test_fp8_encoder_fused_quantisation_at_vit_l_width(dev, monkeypatch):
    """ViT-L/14 width (d 1024, 16 heads, mlp 4096, 257 tokens), 2 full layers + the CLS-only last one on 48 frames (12336 rows: the persistent
    fp8 kernel takes every projection): LayerNorm -> e4m3 in one pass and fc1's epilogue quantisation against the same engine with the separate
    quantiser passes (AVLLM_F8_UNFUSED_QUANT=1), and both against the fake-quantised oracle."""
    from avllm.arch import ClipCfg
    from avllm.engine import ClipEngine
    from bars import FP8_ENC_REL_L2
    from oracle import avsr_oracle as O
    from oracle import weights as Wt
    c = Wt.ClipCfg(hidden=1024, heads=16, layers=3, mlp=4096, image=224, patch=14)
    sd = Wt.clip_weights(c, seed=6)
    frames = torch.randn(48, 3, 224, 224, generator=torch.Generator().manual_seed(10))
    eng = ClipEngine(sd, ClipCfg(**vars(c)), torch.bfloat16, dev, fp8=True)
    fused = eng.forward(frames.to(dev)).float().cpu()
    from avllm import lib as Lk
    with Lk.knob("F8_UNFUSED_QUANT", 1):
        unfused = eng.forward(frames.to(dev)).float().cpu()
    # they differ by the bf16 rounding the fused path skips, which flips e4m3 rounding decisions (each worth 2^-4 of the element): same class as
    # either path's distance to the oracle
    assert rel_l2(fused, unfused) < FP8_ENC_REL_L2, rel_l2(fused, unfused)
    with torch.no_grad(), O.fp8_mode():
        ref = O.clip_vision_cls(sd, c, frames[:8])
    assert rel_l2(fused[:8], ref) < FP8_ENC_REL_L2, rel_l2(fused[:8], ref)
    assert rel_l2(unfused[:8], ref) < FP8_ENC_REL_L2
