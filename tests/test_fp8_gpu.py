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
