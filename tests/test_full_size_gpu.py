"""BASELINE.json configs[1] at its FULL size -- 16 clips x 125 frames, whisper-small + ViT-B/16 -> Llama-2-7B with LoRA r16, bf16, seeded random
weights of the true shapes, the very workload bench.py times.  The CPU oracle needs ~20 s for ONE clip at this depth (test_pin_bf16_gpu.py runs
that), so at the full batch parity goes through properties that do not depend on size:

  * a sample's logits do not depend on its neighbours: clips 0..7 give the same logits, bit for bit, whatever clips 8..15 of the batch are.  Every
    kernel of the path is row-independent -- a GEMM row's k order does not depend on the other rows of its tile, one (clip, head) per attention
    workgroup, per-row normalisations, per-sample fusion and pooling; only the loss mean mixes samples.  (The batch SIZE stays 16: a smaller batch
    takes other tilings for some launches, which sum k in another order; cuBLAS behaves the same way under the reference.)
  * the loss is the token-weighted mean over samples and the LoRA gradient is linear in the per-token loss weights: with the labels of one half of
    the batch masked out (-100) and then the other, the two losses / gradients combine to the full batch's (loss = sum of token losses / number of
    label tokens, HF LlamaForCausalLM shift-and-mean; clip_whisper_model.py:586-619);
  * a second identical call returns the same logits bit for bit (no state left in the workspaces between steps) and the same gradient up to the order
    of the fp32 atomic adds that sum dA / dB over row slabs.
LoRA dropout is off here (the counter-based mask of a step is a function of the step counter)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from bars import bf16_depth_rel_l2, rel_l2  # noqa: E402


@pytest.fixture(scope="module")
def full_model(dev):
    from avllm.model import ClipWhisperModel
    torch.cuda.empty_cache()
    m = ClipWhisperModel("meta-llama/Llama-2-7b-hf", "openai/whisper-small", "openai/clip-vit-base-patch16", device=dev, max_seq_len=512,
                         precision="bf16", seed=0, synthetic_weights=True, lora_dropout=0.0).train()
    eng = m.llm_engine
    eng.lora_p.normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(9))      # B = 0 at initialisation would hide the adapters
    eng.pack_lora()
    yield m
    del m
    torch.cuda.empty_cache()


def _batch(cfg, B, frames, dev, seed=1234):
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import synthetic_batch
    return synthetic_batch(cfg, B, frames, seed, dev)


def _step(m, audio, video, labels, prompt):
    eng = m.llm_engine
    eng.lora_g.zero_()
    out = m(audio=audio, video=video, prompt=prompt, labels=labels)
    out["loss"].backward()
    return float(out["loss"].detach()), out["logits"].clone(), eng.lora_g.clone()


def test_full_size_batch_against_its_halves(dev, full_model):
    m = full_model
    B, frames = 16, 125
    audio, video, labels, prompt = _batch(m.cfg, B, frames, dev)
    loss, logits, grad = _step(m, audio, video, labels, prompt)
    assert torch.isfinite(logits.float()).all() and loss == loss and float(grad.abs().max()) > 0
    loss2, logits2, grad2 = _step(m, audio, video, labels, prompt)                          # idempotence: same call, same bits in the forward;
    assert torch.equal(logits, logits2)                                                      # dA / dB are summed over row slabs with fp32 atomics: order only
    assert rel_l2(grad2, grad) < 1e-6
    # neighbours: other clips in slots 8..15
    a2, v2, l2, p2 = _batch(m.cfg, B, frames, dev, seed=999)
    a2[:8], v2[:8], l2[:8], p2[:8] = audio[:8], video[:8], labels[:8], prompt[:8]
    _, logits_y, _ = _step(m, a2, v2, l2, p2)
    nd = int((logits_y[:8] != logits[:8]).sum())
    assert nd == 0, f"{nd} of {logits[:8].numel()} logits of clips 0..7 changed with clips 8..15"
    assert not torch.equal(logits_y[8:], logits[8:])
    del logits_y, a2, v2
    # linearity in the label mask
    pad = m.tokenizer.pad_token_id
    ntok = (m._prep_labels(labels)[:, 1:] != -100).sum(1).float()                            # label tokens per sample after the shift
    parts = []
    for lo in (0, 8):
        lab = labels.clone()
        lab[(8 - lo):(16 - lo)] = pad                                                        # the OTHER half carries no loss
        l_h, lg_h, g_h = _step(m, audio, video, lab, prompt)
        assert torch.equal(lg_h, logits)                                                     # labels do not reach the logits
        parts.append((float(ntok[lo:lo + 8].sum()), l_h, g_h))
    n = sum(p[0] for p in parts)
    assert n == float(ntok.sum()) and n > 100
    assert abs(loss - sum(p[0] * p[1] for p in parts) / n) < 2e-5 * abs(loss)               # fp32 atomics of ~500 token losses: order only
    gsum = sum(p[2] * (p[0] / n) for p in parts)
    # The forward is the same in all three runs; in the backward every gradient row of a half-run is n / n_half times the full run's before each
    # bf16 rounding (the loss normaliser differs), so the roundings of 32 layers differ: the backward's own share of tests/bars.py's depth estimate
    # (1.13e-3 * sqrt(8 L) = 1.8e-2 at L = 32; measured 1.6e-2), bar = 2x
    err = rel_l2(grad, gsum)
    assert err < bf16_depth_rel_l2(32), f"LoRA gradient of the batch vs the token-weighted sum of its halves' gradients: rel-L2 {err:.2e}"


def test_full_size_greedy_decode_does_not_depend_on_neighbours(dev, full_model):
    """generate() at the bench's decode size (8 clips, 7B shapes, adapters attached -> the fused token step with the LoRA side term): the new
    tokens of clips 0..3 are the same whatever clips 4..7 are, and the same on a second call.  With seeded random weights the logits are nearly flat,
    so a single changed low-order bit anywhere in the step would change an argmax within a few tokens (clip_whisper_model.py:1240-1348)."""
    m = full_model.eval()
    try:
        B, frames, new = 8, 125, 12
        audio, video, _, prompt = _batch(m.cfg, B, frames, dev)
        with torch.no_grad():
            t1 = m.generate(audio=audio, video=video, prompt=prompt, max_new_tokens=new)
            t2 = m.generate(audio=audio, video=video, prompt=prompt, max_new_tokens=new)
            a2, v2, _, p2 = _batch(m.cfg, B, frames, dev, seed=4242)
            a2[:4], v2[:4], p2[:4] = audio[:4], video[:4], prompt[:4]
            t3 = m.generate(audio=a2, video=v2, prompt=p2, max_new_tokens=new)
        assert t1.shape[0] == B and t1.shape[1] >= 1
        assert torch.equal(t1, t2)
        n = min(t1.shape[1], t3.shape[1])
        assert torch.equal(t1[:4, :n], t3[:4, :n])
        assert not torch.equal(t1[4:, :n], t3[4:, :n])
        assert m.llm_engine.decode_is_fused(B)                     # the path under test is the 5-launch token step
    finally:
        full_model.train()


def test_full_size_config5_fp8_step_properties(dev, full_model):
    """BASELINE configs[4] at the size bench.py's `variants.config5_fp8` leg runs it (whisper-large-v3 + ViT-L/14 -> Mistral-7B geometry, 4 clips x 750
    frames, block-scaled fp8 on the frozen forward projections; parity unpinned for the fp8 rule itself -- the reference has no fp8): the same
    size-independent properties.  MX scales are per 32-element block of ONE row (layout 0 only groups rows for addressing), so a clip's logits must not
    depend on its neighbours here either; and the fp8 and the bf16 arithmetic of the same model must agree to the depth form of the fp8 bar (tests/bars.py fp8_depth_rel_l2)."""
    from avllm.model import ClipWhisperModel
    from bars import FP8_LOSS_ABS, fp8_depth_rel_l2
    del full_model                                             # (fixture order: the 7B bf16 model of this module stays alive; this one is built beside it)
    m = ClipWhisperModel("mistralai/Mistral-7B-v0.1", "openai/whisper-large-v3", "openai/clip-vit-large-patch14", device=dev, max_seq_len=512,
                         precision="fp8", seed=0, synthetic_weights=True, lora_dropout=0.0).train()
    eng = m.llm_engine
    eng.lora_p.normal_(0, 0.02, generator=torch.Generator(device=dev).manual_seed(9))
    eng.pack_lora()
    B, frames = 4, 750
    audio, video, labels, prompt = _batch(m.cfg, B, frames, dev, seed=4321)
    loss, logits, grad = _step(m, audio, video, labels, prompt)
    assert torch.isfinite(logits.float()).all() and loss == loss and float(grad.abs().max()) > 0
    _, logits2, _ = _step(m, audio, video, labels, prompt)
    assert torch.equal(logits, logits2)
    a2, v2, l2, p2 = _batch(m.cfg, B, frames, dev, seed=77)
    a2[:2], v2[:2], l2[:2], p2[:2] = audio[:2], video[:2], labels[:2], prompt[:2]
    _, logits_y, _ = _step(m, a2, v2, l2, p2)
    nd = int((logits_y[:2] != logits[:2]).sum())
    assert nd == 0, f"{nd} of {logits[:2].numel()} logits of clips 0..1 changed with clips 2..3 (fp8)"
    assert not torch.equal(logits_y[2:], logits[2:])
    del logits_y, a2, v2
    for e in (m.whisper_engine, m.clip_engine, m.llm_engine):
        e.desc.fp8 = 0
    loss_b, logits_b, _ = _step(m, audio, video, labels, prompt)
    err = rel_l2(logits, logits_b)
    assert err < fp8_depth_rel_l2(m.cfg.llama.layers) and abs(loss - loss_b) < FP8_LOSS_ABS, (err, loss, loss_b)      # tests/bars.py: 0.215 at 32 layers
    del m
    torch.cuda.empty_cache()
