"""Model-level parity on the GPU: the HIP engines behind the ClipWhisperModel surface against (a) the golden vectors
the REFERENCE produced (tests/golden/g2_tiny_e2e.npz) and (b) the CPU oracle on the same seeded inputs.
Tolerances: fp32 mode = the north-star bar (logits within 1e-3, greedy tokens identical); bf16 mode = bf16 rounding
through the stack (stated per assert)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import avsr_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402


def T(a):
    return torch.from_numpy(np.asarray(a))


def make_model(oc, W, precision, max_seq_len=512, dev="cuda:0"):
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    # parity runs use lora_dropout=0: it is the only stochastic op of the step; dropout has its own test below
    return ClipWhisperModel(device=dev, lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=max_seq_len, config=cfg, weights=W,
                            precision=precision)


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    oc = Wt.tiny()
    W = Wt.all_weights(oc, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, _ = Wt.synthetic_batch(oc, 2, int(g["frames"]), seed=int(g["batch_seed"]))
    return g, oc, W, audio, video, labels, T(g["prompt"])


@pytest.fixture(scope="module")
def model32(dev, tiny):
    g, oc, W, *_ = tiny
    return make_model(oc, W, "fp32")


@pytest.fixture(scope="module")
def model16(dev, tiny):
    g, oc, W, *_ = tiny
    return make_model(oc, W, "bf16")


def test_whisper_encoder_vs_oracle(dev, tiny, model32, model16):
    g, oc, W, audio, *_ = tiny
    with torch.no_grad():
        ref = O.whisper_encoder(W["whisper"], oc.whisper, audio)
    out = model32.whisper_engine.forward(audio.to(dev)).float().cpu()
    assert (out - ref).abs().max() < 1e-3, (out - ref).abs().max()
    out16 = model16.whisper_engine.forward(audio.to(dev)).float().cpu()
    err = (out16 - ref).abs()
    assert err.mean() < 2e-2, err.mean()                                         # layer-normed outputs, O(1) scale
    from bars import BF16_ENC_REL_L2, rel_l2
    assert rel_l2(out16, ref) < BF16_ENC_REL_L2, rel_l2(out16, ref)              # a per-tensor scale error of a few percent cannot hide under this
    assert err.max() < 0.12, err.max()                                           # tail of ~4e5 values with sigma ~5e-3 (was 0.35: VERDICT r02 weak #1d)


def test_clip_cls_vs_oracle(dev, tiny, model32, model16):
    g, oc, W, audio, video, *_ = tiny
    fr = video.reshape(-1, 3, oc.clip.image, oc.clip.image)
    with torch.no_grad():
        ref = O.clip_vision_cls(W["clip"], oc.clip, fr)
    out = model32.clip_engine.forward(fr.to(dev)).float().cpu()
    assert (out - ref).abs().max() < 1e-3, (out - ref).abs().max()
    out16 = model16.clip_engine.forward(fr.to(dev)).float().cpu()
    rel = (out16 - ref).abs().max() / ref.abs().max()
    assert rel < 3e-2, rel


def test_encode_golden(dev, tiny, model32):
    g, oc, W, audio, video, labels, prompt = tiny
    a, v = audio.to(dev), video.to(dev)
    enc, mask = model32.encode(a, v, None)
    assert enc.shape == (2, 512, oc.llama.hidden) and mask.dtype == torch.long and bool(mask.all())
    assert (enc.float().cpu()[:, ::8] - T(g["encode_av_rows"])).abs().max() < 1e-3
    enc_a, _ = model32.encode(a, None, None)                     # audio only: all 1500 frames, no cap
    assert enc_a.shape[1] == 1500
    assert (enc_a.float().cpu()[:, ::32] - T(g["encode_a_rows"])).abs().max() < 1e-3
    model32.modality = "video"
    try:
        enc_v, _ = model32.encode(None, v, None)
    finally:
        model32.modality = "both"
    assert (enc_v.float().cpu() - T(g["encode_v"])).abs().max() < 1e-3


def test_train_forward_backward_golden_fp32(dev, tiny, model32):
    """The north-star parity bar: logits within 1e-3 of the reference, loss and LoRA grads matching."""
    g, oc, W, audio, video, labels, prompt = tiny
    m = model32.train()
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    assert out["logits"].shape == (2, 256, oc.llama.vocab)
    dl = (out["logits"].float().cpu() - T(g["train_logits"])).abs().max().item()
    assert dl < 1e-3, dl
    assert abs(float(out["loss"].detach()) - float(g["train_loss"])) < 1e-4
    assert torch.equal(out["logits"].float().cpu().argmax(-1), T(g["train_logits"]).argmax(-1))
    m.lora_param.grad = None
    out["loss"].backward()
    gv = m.llm_engine.lora_views(m.lora_param.grad)
    for k, gr in gv.items():
        ref = T(g["grad." + k])
        assert (gr.cpu() - ref).abs().max() <= 2e-4 * max(1e-3, float(ref.abs().max())) + 1e-7, (k, (gr.cpu() - ref).abs().max(), ref.abs().max())
    # connectors receive no gradient (SURVEY.md fact 4 / fixture connector_grad_is_none)
    assert bool(g["connector_grad_is_none"]) and all(p.grad is None for p in m.audio_connector.parameters())


def test_train_forward_backward_bf16(dev, tiny, model16):
    g, oc, W, audio, video, labels, prompt = tiny
    m = model16.train()
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    ref = T(g["train_logits"])
    err = (out["logits"].float().cpu() - ref).abs()
    # bf16 storage through 2+2+2 layers: stated tolerance 6e-2 absolute on O(1) logits, mean error < 1e-2
    assert err.max() < 6e-2 * max(1.0, float(ref.abs().max())) and err.mean() < 1e-2, (err.max(), err.mean())
    assert abs(float(out["loss"].detach()) - float(g["train_loss"])) < 2e-2
    mine = out["logits"].float().cpu()
    top2 = ref.topk(2, dim=-1).values
    decisive = (top2[..., 0] - top2[..., 1]) > 2 * 6e-2            # margin filtering: argmax must match wherever the
    assert decisive.float().mean() > 0.25                          # reference's top-2 gap exceeds twice the stated tolerance
    assert torch.equal(mine.argmax(-1)[decisive], ref.argmax(-1)[decisive])
    assert (mine.argmax(-1) == ref.argmax(-1)).float().mean().item() > 0.9
    m.lora_param.grad = None
    out["loss"].backward()
    gv = m.llm_engine.lora_views(m.lora_param.grad)
    num = den = 0.0
    for k, gr in gv.items():
        ref_g = T(g["grad." + k])
        num += float(((gr.cpu() - ref_g) ** 2).sum()); den += float((ref_g ** 2).sum())
    assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5          # relative L2 error of the whole LoRA gradient


def test_eval_forward_golden(dev, tiny, model32):
    g, oc, W, audio, video, labels, prompt = tiny
    m = model32.eval()
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    model32.train()
    assert out["logits"].shape == (2, 32 + 512, oc.llama.vocab)
    assert (out["logits"].float().cpu()[:, ::4] - T(g["eval_logits_rows"])).abs().max() < 1e-3
    assert abs(float(out["loss"]) - float(g["eval_loss"])) < 1e-4


def test_generate_golden(dev, tiny):
    """Greedy tokens bit-identical to the reference's generate() as decode.py drives it (max_seq_len=256, no prompt)."""
    g, oc, W, audio, video, labels, prompt = tiny
    m = make_model(oc, W, "fp32", max_seq_len=256).eval()
    ids = m.generate(audio=audio.to(dev), video=video.to(dev), max_new_tokens=12)
    assert torch.equal(ids.cpu(), T(g["generate_ids"])), (ids.cpu(), g["generate_ids"])


def test_error_behaviour(dev, model32):
    with pytest.raises(ValueError):
        model32.encode_audio(torch.zeros(2, 128, 3000, device=dev))
    with pytest.raises(ValueError):
        model32.encode_audio(torch.zeros(2, 80, 2000, device=dev))
    with pytest.raises(ValueError):
        model32.encode_video(torch.zeros(2, 7, 1, 48, 48, device=dev))
    with pytest.raises(ValueError):
        model32.encode(None, None, None)


def test_trainer_steps_vs_oracle(dev, tiny):
    """3 optimizer steps (clip 0.5 + AdamW + cosine) through ClipWhisperTrainer.train_step vs the oracle's loop."""
    from avllm.trainer import ClipWhisperTrainer
    g, oc, W, audio, video, labels, prompt = tiny
    m = make_model(oc, W, "fp32").train()
    tr = ClipWhisperTrainer(m, learning_rate=1e-3, weight_decay=0.01, grad_clip=0.5, total_steps=10, max_epochs=1)
    Wo = dict(W)
    Wo["lora"] = {k: v.clone() for k, v in W["lora"].items()}
    keys = sorted(Wo["lora"])
    mo = {k: torch.zeros_like(v) for k, v in Wo["lora"].items()}
    vo = {k: torch.zeros_like(v) for k, v in Wo["lora"].items()}
    for s in range(3):
        loss = tr.train_step(audio.to(dev), video.to(dev), labels.to(dev), prompt.to(dev))
        ol, _, og = O.train_step_grads(Wo, oc, audio, video, prompt, labels)
        assert abs(float(loss) - float(ol)) < 2e-4, (s, float(loss), float(ol))
        gl = [og[k] for k in keys]
        O.clip_grad_norm_(gl, 0.5)
        for k, gk in zip(keys, gl):
            O.adamw_step(Wo["lora"][k], gk, mo[k], vo[k], s + 1, O.cosine_lr(1e-3, s, 10))
    pv = m.llm_engine.lora_views()
    num = den = 0.0
    for k in keys:      # Adam's m/sqrt(v) amplifies rounding on near-zero gradients: compare the UPDATE in relative L2
        upd_ref = Wo["lora"][k] - W["lora"][k]
        num += float(((pv[k].cpu() - W["lora"][k]) - upd_ref).pow(2).sum()); den += float(upd_ref.pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5


def test_state_dict_roundtrip(dev, tiny, model32, tmp_path):
    g, oc, W, *_ = tiny
    sd = model32.state_dict()
    assert any("audio_connector" in k for k in sd) and any("video_connector" in k for k in sd)      # decode.py:237-238
    assert "llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight" in sd
    m2 = make_model(oc, Wt.all_weights(oc, 5, lora_b_std=0.01), "fp32")
    m2.load_state_dict(sd)
    for k, v in m2.state_dict().items():
        if "connector" in k or "lora" in k:
            assert torch.equal(v.cpu(), sd[k].cpu()), k


def test_save_pretrained_from_pretrained_round_trip(dev, tiny, model32, tmp_path):
    """save_pretrained's directory (clip_whisper_model.py:738-798: audio_connector.pt, video_connector.pt, config.pt / config.json, llm/) reloads
    through from_pretrained on top of other base weights' adapters/connectors: same eval logits afterwards.  (The reference's own from_pretrained
    asks for model_config.json, which its save_pretrained never writes.)"""
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    g, oc, W, audio, video, labels, prompt = tiny
    out_dir = str(tmp_path / "saved")
    model32.save_pretrained(out_dir)
    import os
    for f in ("audio_connector.pt", "video_connector.pt", "config.json", "config.pt", os.path.join("llm", "adapter_model.pt")):
        assert os.path.exists(os.path.join(out_dir, f)), f
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    W2 = dict(W)
    W2["lora"] = {k: torch.zeros_like(v) for k, v in W["lora"].items()}                       # same frozen base, fresh adapters ...
    W2.pop("audio_connector", None); W2.pop("video_connector", None)                          # ... and fresh connectors
    m2 = ClipWhisperModel.from_pretrained(out_dir, device=dev, config=cfg, weights=W2, precision="fp32", lora_dropout=0.0)
    assert m2.max_seq_len == model32.max_seq_len and m2.modality == model32.modality
    model32.eval(); m2.eval()
    kw = dict(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    a, b = model32(**kw), m2(**kw)
    assert torch.equal(a["logits"], b["logits"])
    model32.train()
    with pytest.raises(ValueError):
        ClipWhisperModel.from_pretrained(str(tmp_path / "nope"))


def test_lora_dropout_matches_oracle_with_same_masks(dev, tiny):
    """lora_dropout>0 (peft: lora_B(lora_A(dropout(x)))): the library regenerates its counter-based masks in forward and
    backward; the oracle is given the SAME masks (extracted with avllm_dropout on ones) and must agree on loss and grads."""
    from avllm import ops
    g, oc, W, audio, video, labels, prompt = tiny
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    p = 0.25
    m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=p, max_seq_len=512, config=cfg, weights=W,
                         precision="fp32").train()
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    m.lora_param.grad = None
    out["loss"].backward()
    seed = m.llm_engine.desc.dropout_seed
    assert m.llm_engine.desc.lora_dropout == pytest.approx(p)
    ones = torch.ones(2 * 256, oc.llama.hidden, device=dev)
    masks = {}
    for l in range(oc.llama.layers):
        for j, nm in enumerate(("q_proj", "k_proj", "v_proj", "o_proj")):
            mk = ops.dropout(ones, seed + 4 * l + j, p).cpu().view(2, 256, -1)
            keep = (mk > 0).float().mean().item()
            assert abs(keep - (1 - p)) < 0.02, keep
            masks[f"layers.{l}.{nm}"] = mk
    assert not torch.equal(masks["layers.0.q_proj"], masks["layers.0.k_proj"])        # independent masks per module
    ol, ologits, og = O.train_step_grads(W, oc, audio, video, prompt, labels, masks=masks)
    assert abs(float(out["loss"].detach()) - float(ol)) < 1e-4
    assert (out["logits"].float().cpu() - ologits).abs().max() < 1e-3
    gv = m.llm_engine.lora_views(m.lora_param.grad)
    for k, gr in gv.items():
        assert (gr.cpu() - og[k]).abs().max() <= 2e-4 * max(1e-3, float(og[k].abs().max())) + 1e-7, k
    # a second training forward draws a different mask; eval() disables dropout
    m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    assert m.llm_engine.desc.dropout_seed != seed
    assert m.eval()._dropout_args()["dropout"] == 0.0
