"""Pins the arithmetic bench.py times -- bf16 storage, fused LoRA dropout, real widths -- against the oracle.

The north-star bar is carried by precision="fp32", which runs different kernels from the benchmarked ones; these tests close
that gap:
  * op level: the dropout-masked dX GEMM epilogue (drop_seed/drop_p of avllm_gemm_desc; csrc/engine.hip adapter input gradient) against
    GEMM -> avllm_dropout mask -> +R, and the fused A-operand mask of the rank-side GEMM at every M;
  * model level: a bf16 step with lora_dropout > 0 on a geometry where the fused paths are taken (d % 256 == 0, r <= 16), the oracle
    fed the SAME masks -- plus a negative control (other masks must miss the bar by far);
  * G3: real-width single layers against the oracle run on the box's host cores: one Llama-2-7B-width decoder layer + lm_head +
    shifted CE forward AND backward (d 4096, ffn 11008, V 32000) at B=2 (fp32 north-star bar + bf16 bar) and at the bench's own
    M = 16 x 256 rows in bf16 with lora_dropout = 0.05 (the persistent 4-wave GEMM, the masked 16-wave epilogue, the fused rank-side
    masks and the MFMA attention backward are exactly the launches bench.py makes), one Whisper-small layer at T = 1500, CLIP ViT-B/16
    layers at 197 tokens;
  * the same full-size step through two GEMM tilings must agree (was a manual run in round 1).
Bars: tests/bars.py (stated once, derived there).
Reference semantics: peft lora.Linear dropout (clip_whisper_model.py:961-1005), HF:models/llama/modeling_llama.py:284-324,
HF:models/whisper/modeling_whisper.py:379-413, HF:models/clip/modeling_clip.py:362-384.
"""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

import bars as Bar  # noqa: E402
from bars import rel_l2  # noqa: E402
from oracle import avsr_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402
from test_ops_gpu import rnd  # noqa: E402


# ------------------------------------------------------------------------------------------------ op level
@pytest.mark.parametrize("M,N,K,with_r", [(4096, 4096, 64, True), (4096, 4096, 192, True), (4096, 1024, 64, False), (4000, 4096, 64, True),
                                          (512, 256, 64, True), (300, 264, 128, True)])
def test_masked_dx_gemm_epilogue_bf16(dev, M, N, K, with_r):
    """C = mask * (A.B^T) / (1-p) + R with mask = keep(seed, m*N+n, p): the masked GEMM must equal the plain product with the
    avllm_dropout mask applied, within one bf16 ulp (the fp32 sums differ by summation order only), and be EXACTLY R where the mask
    drops.  (4096,4096,64) is the bench's launch (16-wave kernel, 16-byte epilogue, av_mask8); (512,256,64) takes the 128x128 kernel
    (8-byte epilogue, av_keep); (300,264,128) ragged edges."""
    from avllm import ops
    p, seed = 0.05, 0xC0FFEE + M + N
    A = rnd(M, K, dtype=torch.bfloat16, seed=31)
    Bw = rnd(N, K, dtype=torch.bfloat16, seed=32, scale=K ** -0.5)
    R = rnd(M, N, dtype=torch.bfloat16, seed=33) if with_r else None
    got = ops.gemm(A, Bw, R=R, drop=(seed, p)).float()
    mask = ops.dropout(torch.ones(M, N, device=dev, dtype=torch.float32), seed, p)        # 0 or 1/(1-p'), p' = p quantised to 1/65536
    keep_frac = (mask > 0).float().mean().item()
    assert abs(keep_frac - (1 - p)) < 5e-3, keep_frac
    prod = A.float() @ Bw.float().t()
    exp = prod * mask + (R.float() if with_r else 0.0)
    exp16 = exp.to(torch.bfloat16).float()
    err = (got - exp16).abs()
    tol = exp16.abs() * 2.0 ** -7 + 1e-6                                                    # one bf16 ulp of the expected value
    assert int((err > tol).sum()) == 0, (err.max().item(), int((err > tol).sum()))
    dropped = mask == 0
    assert torch.equal(got[dropped], (R.float() if with_r else torch.zeros_like(got))[dropped])
    # sensitivity: a different seed gives a different mask (the assertion above would fail on ~2p of the elements)
    other = ops.gemm(A, Bw, R=R, drop=(seed + 1, p)).float()
    assert ((other - exp16).abs() > tol).float().mean().item() > p


@pytest.mark.parametrize("M,N,nj,ld", [(4096, 4096, 3, 192), (4096, 4096, 1, 64), (1000, 256, 2, 192), (40, 128, 3, 64)])
def test_fused_adapter_input_gradient_under_dropout(dev, M, N, nj, ld):
    """avllm_lora_dx_masked: out = R + sum_j mask_j o (T_j . A_j)/(1-p) in one pass over dX (q/k/v: three adapters, one launch) against
    fp32 torch with the avllm_dropout masks, within one bf16 ulp; in place (out aliases R) as the backward pass uses it; and equal to
    the per-adapter masked GEMMs it replaces up to their three intermediate roundings."""
    from avllm import ops
    p, r = 0.05, 16
    seeds = [1000 + 4 * 7 + j for j in range(nj)]
    Tbig = torch.zeros(M, ld, device=dev, dtype=torch.bfloat16)
    ATbig = torch.zeros(N, ld, device=dev, dtype=torch.bfloat16)
    Ts, ATs = [], []
    for j in range(nj):
        c0 = j * 64 if ld == 192 else 0
        if ld == 64 and j > 0:                                        # separate images per adapter
            Tbig2, ATbig2 = torch.zeros_like(Tbig), torch.zeros_like(ATbig)
        else:
            Tbig2, ATbig2 = Tbig, ATbig
        Tbig2[:, c0:c0 + r] = rnd(M, r, dtype=torch.bfloat16, seed=60 + j)
        ATbig2[:, c0:c0 + r] = rnd(N, r, dtype=torch.bfloat16, seed=70 + j, scale=0.1)
        Ts.append(Tbig2[:, c0:c0 + 64]); ATs.append(ATbig2[:, c0:c0 + 64])
    R = rnd(M, N, dtype=torch.bfloat16, seed=80)
    exp = R.float()
    for j in range(nj):
        mask = ops.dropout(torch.ones(M, N, device=dev, dtype=torch.float32), seeds[j], p)
        exp = exp + mask * (Ts[j][:, :r].float() @ ATs[j][:, :r].float().t())
    exp16 = exp.to(torch.bfloat16).float()
    got = ops.lora_dx_masked(Ts, ATs, seeds, r, p, R=R).float()
    err = (got - exp16).abs()
    tol = exp16.abs() * 2.0 ** -7 + 1e-6
    assert int((err > tol).sum()) == 0, (err.max().item(), int((err > tol).sum()))
    inplace = R.clone()
    ops.lora_dx_masked(Ts, ATs, seeds, r, p, R=inplace, out=inplace)
    assert torch.equal(inplace.float(), got)
    seq = R.clone()                                                  # what round 1 did: one masked K=64 GEMM per adapter, each rounding dX
    for j in range(nj):
        seq = ops.gemm(Ts[j], ATs[j], R=seq, drop=(seeds[j], p))
    assert rel_l2(seq.float(), got) < 4e-3


@pytest.mark.parametrize("M", [8, 16, 200, 4096])
def test_rank_side_gemm_fused_a_dropout_every_m(dev, M):
    """t = alpha * dropout(x) . A_pad^T with the mask generated on the A fragments (a_drop) == the same GEMM on a materialised
    dropout(x), at every M -- including M <= 16, which used to fall into the small-M kernel WITHOUT the mask."""
    from avllm import ops
    K, p, seed = 4096, 0.25, 777
    x = rnd(M, K, dtype=torch.bfloat16, seed=41)
    Ap = torch.zeros(64, K, device=dev, dtype=torch.bfloat16)
    Ap[:16] = rnd(16, K, dtype=torch.bfloat16, seed=42, scale=K ** -0.5)
    xd = ops.dropout(x, seed, p)
    ref = ops.gemm(xd, Ap, alpha=2.0, n_valid=16).float()
    got = ops.gemm(x, Ap, alpha=2.0, n_valid=16, a_drop=(seed, p)).float()
    assert rel_l2(got[:, :16], ref[:, :16]) < 1e-2
    assert got[:, 16:].abs().max().item() == 0
    plain = ops.gemm(x, Ap, alpha=2.0, n_valid=16).float()
    assert rel_l2(plain[:, :16], ref[:, :16]) > 0.2                       # the mask matters: an unmasked product is far off


# ------------------------------------------------------------------------------------------------ model level, tiny geometry
def _extract_masks(ops, eng, B, S, d, layers, p, dev):
    seed = eng.desc.dropout_seed
    ones = torch.ones(B * S, d, device=dev, dtype=torch.float32)
    masks = {}
    for l in range(layers):
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj", "o_proj")):
            masks[f"layers.{l}.{nm}"] = ops.dropout(ones, seed + 4 * l + j, p).cpu().view(B, S, d)
    return masks


def test_lora_dropout_bf16_fused_paths_match_oracle_with_same_masks(dev, golden_dir):
    """bf16 + lora_dropout > 0 on d = 256, r = 16: `fuse_drop` is true (csrc/engine.hip), i.e. no dropout(x) tensor exists -- the
    rank-side GEMM masks its A fragments, the dA reduction masks the wide operand while staging it, the adapter's input gradient is
    masked in a GEMM epilogue.  The oracle gets the SAME masks (avllm_dropout on ones) and must agree on loss, logits and every LoRA
    gradient within the bf16 bars; with OTHER masks it must miss them by far (negative control: the masks are really applied, and
    consistently, in all three places)."""
    import numpy as np
    from avllm import ops
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    oc = Wt.tiny()
    W = Wt.all_weights(oc, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, _ = Wt.synthetic_batch(oc, 2, int(g["frames"]), seed=int(g["batch_seed"]))
    prompt = torch.from_numpy(g["prompt"])
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    p = 0.25
    assert oc.llama.hidden % 256 == 0 and oc.lora.r <= 16            # the fused-dropout condition of engine.hip
    m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=p, max_seq_len=512, config=cfg, weights=W,
                         precision="bf16").train()
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    m.lora_param.grad = None
    out["loss"].backward()
    masks = _extract_masks(ops, m.llm_engine, 2, 256, oc.llama.hidden, oc.llama.layers, p, dev)
    ol, ologits, og = O.train_step_grads(W, oc, audio, video, prompt, labels, masks=masks)
    assert abs(float(out["loss"].detach()) - float(ol)) < Bar.BF16_LOSS_ABS
    assert rel_l2(out["logits"].float().cpu(), ologits) < Bar.BF16_LOGITS_REL_L2
    gv = {k: v.cpu() for k, v in m.llm_engine.lora_views(m.lora_param.grad).items()}
    allg = torch.cat([gv[k].flatten() for k in sorted(gv)])
    allo = torch.cat([og[k].flatten() for k in sorted(gv)])
    assert rel_l2(allg, allo) < Bar.BF16_GRAD_REL_L2, rel_l2(allg, allo)
    for k in gv:
        assert rel_l2(gv[k], og[k]) < Bar.BF16_GRAD_TENSOR_REL_L2, (k, rel_l2(gv[k], og[k]))
    # negative control: masks of another seed
    ones = torch.ones(2 * 256, oc.llama.hidden, device=dev)
    wrong = {k: ops.dropout(ones, 987654 + i, p).cpu().view(2, 256, -1) for i, k in enumerate(sorted(masks))}
    _, _, og2 = O.train_step_grads(W, oc, audio, video, prompt, labels, masks=wrong)
    allw = torch.cat([og2[k].flatten() for k in sorted(gv)])
    assert rel_l2(allg, allw) > 4 * Bar.BF16_GRAD_REL_L2, rel_l2(allg, allw)


# ------------------------------------------------------------------------------------------------ G3: real widths
def _bench_labels(B, V, seed):
    g = torch.Generator().manual_seed(seed)
    labels = torch.full((B, 256), -100, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(8, 41, (1,), generator=g))
        labels[b, 0] = 1
        labels[b, 1:1 + n] = torch.randint(3, V, (n,), generator=g)
    return labels


def _oracle_llm_step(sd, lora_w, c, lc, x, labels, masks=None):
    lora = {k: v.clone().requires_grad_(True) for k, v in lora_w.items()}
    h = O.llama_hidden(sd, lora, c, lc, x, masks=masks)
    logits = h @ sd["lm_head.weight"].T
    loss = O.causal_lm_loss(logits, labels)
    loss.backward()
    return loss.detach(), logits.detach(), {k: v.grad for k, v in lora.items()}


@pytest.fixture(scope="module")
def llama7b_layer():
    """One decoder layer + final norm + lm_head at Llama-2-7B width, deterministic weights (oracle/weights.py)."""
    c = Wt.LlamaCfg(hidden=4096, heads=32, layers=1, ffn=11008, vocab=32000)
    lc = Wt.LoraCfg(r=16, alpha=32.0)
    sd = Wt.llama_weights(c, seed=3)
    lora = Wt.lora_weights(c, lc, seed=3, b_std=0.02)
    return c, lc, sd, lora


def _engine(c, lc, sd, lora, dtype, dev):
    from avllm.arch import LlamaCfg, LoraCfg
    from avllm.engine import LlamaEngine
    return LlamaEngine(sd, LlamaCfg(**vars(c)), LoraCfg(lc.r, lc.alpha), lora, dtype, dev, training=True)


def _engine_step(eng, x, labels, dropout=0.0, seed=0):
    logits = eng.fwd_loss(x, labels, want_logits=True, dropout=dropout, seed=seed)
    loss = float(eng.acc[0] / eng.acc[1])
    eng.lora_g.zero_()
    eng.bwd()
    grads = {k: v.cpu().clone() for k, v in eng.lora_views(eng.lora_g).items()}
    return loss, logits.float().cpu(), grads


def test_g3_llama7b_width_layer_fp32_and_bf16(dev, llama7b_layer):
    """S = 256, B = 2: forward logits, shifted-CE loss and every LoRA gradient against the oracle.  fp32: the north-star bar; bf16: the
    derived bar."""
    c, lc, sd, lora = llama7b_layer
    B, S = 2, 256
    x = (torch.randn(B, S, c.hidden, generator=torch.Generator().manual_seed(5)) * 0.5)
    labels = _bench_labels(B, c.vocab, 6)
    ol, ologits, og = _oracle_llm_step(sd, lora, c, lc, x, labels)
    eng = _engine(c, lc, sd, lora, torch.float32, dev)
    loss, logits, grads = _engine_step(eng, x.to(dev), labels.to(dev))
    assert (logits.view_as(ologits) - ologits).abs().max().item() < Bar.F32_LOGITS_ABS
    assert abs(loss - float(ol)) < Bar.F32_LOSS_ABS
    assert torch.equal(logits.view_as(ologits).argmax(-1), ologits.argmax(-1))
    for k, gr in grads.items():
        assert (gr - og[k]).abs().max() <= Bar.F32_GRAD_REL_MAX * max(1e-6, float(og[k].abs().max())) + 1e-9, k
    del eng
    torch.cuda.empty_cache()
    eng = _engine(c, lc, sd, lora, torch.bfloat16, dev)
    loss, logits, grads = _engine_step(eng, x.to(dev).bfloat16(), labels.to(dev))
    assert rel_l2(logits.view_as(ologits), ologits) < Bar.BF16_LOGITS_REL_L2, rel_l2(logits.view_as(ologits), ologits)
    assert abs(loss - float(ol)) < Bar.BF16_LOSS_ABS
    keys = sorted(grads)
    assert rel_l2(torch.cat([grads[k].flatten() for k in keys]), torch.cat([og[k].flatten() for k in keys])) < Bar.BF16_GRAD_REL_L2
    for k in keys:
        assert rel_l2(grads[k], og[k]) < Bar.BF16_GRAD_TENSOR_REL_L2, (k, rel_l2(grads[k], og[k]))


def test_g3_bench_shape_bf16_with_lora_dropout(dev, llama7b_layer):
    """The bench's own launches: M = 16 x 256 rows, bf16, lora_dropout = 0.05 -> persistent 4-wave GEMM (+LoRA K segment), masked-dX
    16-wave epilogue, rank-side GEMMs with fused masks, masked dA reduction, MFMA attention forward/backward.  Oracle with the same
    masks on the host cores; then the same step forced through the 16-wave GEMM tiling must agree with the automatic choice."""
    from avllm import lib as L
    from avllm import ops
    c, lc, sd, lora = llama7b_layer
    B, S, p = 16, 256, 0.05
    x = (torch.randn(B, S, c.hidden, generator=torch.Generator().manual_seed(7)) * 0.5).bfloat16()
    labels = _bench_labels(B, c.vocab, 8)
    eng = _engine(c, lc, sd, lora, torch.bfloat16, dev)
    loss, logits, grads = _engine_step(eng, x.to(dev), labels.to(dev), dropout=p, seed=4242)
    masks = _extract_masks(ops, eng, B, S, c.hidden, 1, p, dev)
    ol, ologits, og = _oracle_llm_step(sd, lora, c, lc, x.float(), labels, masks=masks)
    assert rel_l2(logits.view_as(ologits), ologits) < Bar.BF16_LOGITS_REL_L2, rel_l2(logits.view_as(ologits), ologits)
    assert abs(loss - float(ol)) < Bar.BF16_LOSS_ABS, (loss, float(ol))
    keys = sorted(grads)
    whole = rel_l2(torch.cat([grads[k].flatten() for k in keys]), torch.cat([og[k].flatten() for k in keys]))
    assert whole < Bar.BF16_GRAD_REL_L2, whole
    for k in keys:
        assert rel_l2(grads[k], og[k]) < Bar.BF16_GRAD_TENSOR_REL_L2, (k, rel_l2(grads[k], og[k]))
    # negative control on the masks at full width: the o_proj adapter with the q_proj mask
    wrong = dict(masks)
    wrong["layers.0.o_proj"] = masks["layers.0.q_proj"]
    _, _, og2 = _oracle_llm_step(sd, lora, c, lc, x.float(), labels, masks=wrong)
    # (p = 0.05: the two masks differ on ~2p of the elements -> relative error ~ sqrt(2p/(1-p)) = 0.32, three times the bar)
    assert rel_l2(grads["layers.0.o_proj.lora_A"], og2["layers.0.o_proj.lora_A"]) > 2.5 * Bar.BF16_GRAD_TENSOR_REL_L2
    # the same step through the 16-wave tiling (every projection) against the automatic choice (persistent 4-wave)
    lib = L.load()
    try:
        lib.avllm_set_gemm_variant(5)
        loss5, logits5, grads5 = _engine_step(eng, x.to(dev), labels.to(dev), dropout=p, seed=4242)
    finally:
        lib.avllm_set_gemm_variant(0)
    assert abs(loss5 - loss) < 2e-3, (loss5, loss)
    assert rel_l2(logits5, logits) < 5e-3, rel_l2(logits5, logits)
    assert rel_l2(torch.cat([grads5[k].flatten() for k in keys]), torch.cat([grads[k].flatten() for k in keys])) < 1e-2


def test_g3_whisper_small_layer(dev):
    """One Whisper-small encoder layer (d 768, 12 heads, ffn 3072) behind the conv stem at T = 1500, one 30 s window."""
    from avllm.arch import WhisperCfg
    from avllm.engine import WhisperEngine
    c = Wt.WhisperCfg(d_model=768, heads=12, layers=1, ffn=3072)
    sd = Wt.whisper_weights(c, seed=4)
    mel = torch.randn(1, 80, 3000, generator=torch.Generator().manual_seed(9))
    with torch.no_grad():
        ref = O.whisper_encoder(sd, c, mel)
    out32 = WhisperEngine(sd, WhisperCfg(**vars(c)), torch.float32, dev).forward(mel.to(dev)).float().cpu()
    assert (out32 - ref).abs().max().item() < Bar.F32_LOGITS_ABS
    out16 = WhisperEngine(sd, WhisperCfg(**vars(c)), torch.bfloat16, dev).forward(mel.to(dev)).float().cpu()
    assert rel_l2(out16, ref) < Bar.BF16_ENC_REL_L2, rel_l2(out16, ref)


def test_g3_clip_b16_layers(dev):
    """CLIP ViT-B/16 (d 768, 12 heads, mlp 3072, 197 tokens per 224-pixel frame): one full layer + the CLS-only last layer, 6 frames."""
    from avllm.arch import ClipCfg
    from avllm.engine import ClipEngine
    c = Wt.ClipCfg(hidden=768, heads=12, layers=2, mlp=3072, image=224, patch=16)
    sd = Wt.clip_weights(c, seed=5)
    frames = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(10))
    with torch.no_grad():
        ref = O.clip_vision_cls(sd, c, frames)
    out32 = ClipEngine(sd, ClipCfg(**vars(c)), torch.float32, dev).forward(frames.to(dev)).float().cpu()
    assert (out32 - ref).abs().max().item() < Bar.F32_LOGITS_ABS * max(1.0, float(ref.abs().max()))
    out16 = ClipEngine(sd, ClipCfg(**vars(c)), torch.bfloat16, dev).forward(frames.to(dev)).float().cpu()
    assert rel_l2(out16, ref) < Bar.BF16_ENC_REL_L2, rel_l2(out16, ref)


# ------------------------------------------------------------------------------------------------ the benchmarked arithmetic at the depth it is timed
def test_full_depth_bf16_train_step_vs_oracle(dev):
    """BASELINE configs[1] at FULL depth and width -- 12 Whisper-small layers, 12 ViT-B/16 layers on 8 frames, 32 Llama-2-7B-width
    decoder layers, r16 adapters on q/k/v/o with lora_dropout 0.05 -- one bf16 training forward + backward against the fp32 oracle on the
    box's host cores (the same masks handed over; ~25 s of host time).  The decoder layers share one set of frozen tensors on the host
    (oracle/cpu_baseline.shared_depth_weights: 0.8 GB instead of 27 GB) but are 32 separate weight sets to the HIP engine, and the
    adapters are independent draws per layer.  Bars: tests/bars.py bf16_depth_* (derived there for L = 32, not fitted to this run).
    Reference: clip_whisper_model.py:489-619 -> HF:models/llama/modeling_llama.py:284-324, 435-488."""
    from avllm import ops
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    from oracle import cpu_baseline
    oc = Wt.config2()
    L = oc.llama.layers
    assert (oc.whisper.layers, oc.clip.layers, L, oc.llama.hidden, oc.llama.ffn, oc.llama.vocab) == (12, 12, 32, 4096, 11008, 32000)
    W = cpu_baseline.shared_depth_weights(oc, seed=11, lora_b_std=0.02, distinct_lora=True)
    B, frames, p = 1, 8, 0.05
    audio, video, labels, prompt = Wt.synthetic_batch(oc, B, frames, seed=77)
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=p, max_seq_len=512, config=cfg, weights=W,
                         precision="bf16").train()
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    m.lora_param.grad = None
    out["loss"].backward()
    loss = float(out["loss"].detach())
    logits = out["logits"].float().cpu()
    gv = {k: v.cpu().clone() for k, v in m.llm_engine.lora_views(m.lora_param.grad).items()}
    masks = _extract_masks(ops, m.llm_engine, B, 256, oc.llama.hidden, L, p, dev)
    del m, out
    torch.cuda.empty_cache()
    ol, ologits, og = O.train_step_grads(W, oc, audio, video, prompt, labels, masks=masks)
    e_logits = rel_l2(logits.view_as(ologits), ologits)
    keys = sorted(gv)
    e_grad = rel_l2(torch.cat([gv[k].flatten() for k in keys]), torch.cat([og[k].flatten() for k in keys]))
    per = {k: rel_l2(gv[k], og[k]) for k in keys}
    worst = max(per, key=per.get)
    print(f"full depth: loss {loss:.5f} vs {float(ol):.5f}; logits rel-L2 {e_logits:.4f} (bar {Bar.bf16_depth_rel_l2(L):.4f}); "
          f"LoRA grad rel-L2 {e_grad:.4f} (bar {Bar.bf16_depth_grad_rel_l2(L):.4f}); worst tensor {worst} {per[worst]:.4f}")
    assert abs(loss - float(ol)) < Bar.bf16_depth_loss_abs(L), (loss, float(ol))
    assert e_logits < Bar.bf16_depth_rel_l2(L), e_logits
    assert e_grad < Bar.bf16_depth_grad_rel_l2(L), e_grad
    for k in keys:
        assert per[k] < 2.0 * Bar.bf16_depth_grad_rel_l2(L), (k, per[k])                 # single tensors are noisier than the whole
    # argmax identical wherever the oracle's top-2 margin exceeds the logit error this depth allows
    top2 = ologits.topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 4.0 * Bar.bf16_depth_rel_l2(L) * float(ologits.std())
    assert int(clear.sum()) >= 8, int(clear.sum())               # 32000 random logits: the top two are close at most positions; a handful are not
    assert torch.equal(logits.view_as(ologits).argmax(-1)[clear], ologits.argmax(-1)[clear])
    # (the negative control on the masks lives in test_g3_bench_shape_bf16_with_lora_dropout: a second oracle pass here would double the host time)


# ------------------------------------------------------------------------------------------------ non-finite guard
def test_nan_batch_leaves_lora_state_untouched(dev, golden_dir):
    """trainer/clip_whisper_trainer.py:444-452 skips backward + optimizer on a NaN/Inf loss.  Here the guard is on the device (no host
    sync in the step): a NaN batch must leave lora_p, m and v bit-identical and count one skipped step; the next good batch trains."""
    import numpy as np
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    oc = Wt.tiny()
    W = Wt.all_weights(oc, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, prompt = Wt.synthetic_batch(oc, 2, 3, seed=3)
    # long transcripts: the NaN enters through the audio frames, i.e. behind the ~15 pooled prompt rows; with causal attention only label
    # positions past them can see it (a short transcript scores none of them and trains normally -- as the reference would)
    labels[:, 1:200] = torch.randint(3, oc.llama.vocab, (2, 199), generator=torch.Generator().manual_seed(5))
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    for precision in ("fp32", "bf16"):
        m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.05, max_seq_len=512, config=cfg,
                             weights=W, precision=precision).train()
        tr = ClipWhisperTrainer(m, learning_rate=1e-3, total_steps=10, max_epochs=1)
        tr.train_step(audio.to(dev), video.to(dev), labels.to(dev), prompt.to(dev))
        p0, m0, v0 = m.llm_engine.lora_p.clone(), tr.m.clone(), tr.v.clone()
        bad = audio.clone()
        bad[0, 3, 100] = float("nan")
        loss = tr.train_step(bad.to(dev), video.to(dev), labels.to(dev), prompt.to(dev))
        assert not torch.isfinite(loss).item()
        assert torch.equal(m.llm_engine.lora_p, p0) and torch.equal(tr.m, m0) and torch.equal(tr.v, v0)
        assert tr.skipped_steps == 1
        loss = tr.train_step(audio.to(dev), video.to(dev), labels.to(dev), prompt.to(dev))
        assert torch.isfinite(loss).item() and not torch.equal(m.llm_engine.lora_p, p0) and tr.skipped_steps == 1
        assert torch.isfinite(m.llm_engine.lora_p).all().item()


def test_backward_refuses_overwritten_activations(dev, golden_dir):
    """out = model(...); <another training forward>; out['loss'].backward() must raise instead of silently differentiating the wrong
    activations; an eval forward / generate() in between is harmless (separate inference workspace)."""
    import numpy as np
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    oc = Wt.tiny()
    W = Wt.all_weights(oc, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, prompt = Wt.synthetic_batch(oc, 2, 3, seed=3)
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=512, config=cfg, weights=W,
                         precision="fp32").train()
    kw = dict(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    out = m(**kw)
    out["loss"].backward()
    g_ref = m.lora_param.grad.clone()
    m.lora_param.grad = None
    out = m(**kw)
    m.eval()
    m.generate(audio=audio.to(dev), video=video.to(dev), max_new_tokens=3)       # inference between forward and backward
    m(**kw)                                                                       # eval forward (prefill path)
    m.train()
    out["loss"].backward()
    assert rel_l2(m.lora_param.grad, g_ref) < 1e-5                               # fp32 atomics: summation order only
    m.lora_param.grad = None
    out = m(**kw)
    m(**kw)                                                                       # a second TRAINING forward overwrites the activations
    with pytest.raises(RuntimeError, match="another training forward"):
        out["loss"].backward()


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_graph_replayed_step_matches_eager_step(dev, golden_dir, precision):
    """The captured step (hipGraph replay, per-step scalars in the device-side avllm_step_state) against the same step launched eagerly:
    same losses, same parameters after 5 steps with LoRA dropout, gradient clipping, warm-up + cosine schedule -- i.e. the device-side
    learning rate, Adam bias corrections and dropout seeds follow the host-side rule (trainer/clip_whisper_trainer.py:457-464)."""
    import numpy as np
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    oc = Wt.tiny()
    W = Wt.all_weights(oc, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, prompt = Wt.synthetic_batch(oc, 2, 3, seed=3)
    audio2, video2, labels2, prompt2 = Wt.synthetic_batch(oc, 2, 3, seed=4)
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    runs = {}
    for graph in (False, True):
        m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.1, max_seq_len=512, config=cfg,
                             weights=W, precision=precision).train()
        tr = ClipWhisperTrainer(m, learning_rate=1e-3, total_steps=8, max_epochs=1, warmup_steps=2, grad_clip=0.5, use_graph=graph)
        losses = []
        for s in range(5):
            b = (audio, video, labels, prompt) if s % 2 == 0 else (audio2, video2, labels2, prompt2)      # same signature, new data
            losses.append(float(tr.train_step(*[t.to(dev) for t in b])))
        st = tr.state.cpu().numpy().view(np.uint32)
        assert int(st[0]) == 5 and tr.global_step == 5
        assert abs(float(tr.state.cpu().numpy().view(np.float32)[2]) - tr.lr_at(4)) < 1e-8       # device-side schedule == host-side rule
        runs[graph] = (losses, m.llm_engine.lora_p.cpu().clone())
        if graph:
            assert any(isinstance(v, dict) for v in tr._graphs.values())                     # a graph really was captured and replayed
    (l0, p0), (l1, p1) = runs[False], runs[True]
    tol = 1e-5 if precision == "fp32" else 2e-3
    assert max(abs(a - b) for a, b in zip(l0, l1)) < tol, (l0, l1)
    init = torch.cat([W["lora"][k].flatten() for k in sorted(W["lora"])])
    assert rel_l2(p1, p0) < (1e-4 if precision == "fp32" else 2e-2)
    assert float((p0 - p1).abs().max()) < float((p0.abs().max())) * 0.05 and not torch.equal(p0, torch.zeros_like(p0)) and init.numel() == p0.numel()
