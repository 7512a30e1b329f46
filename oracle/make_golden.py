#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build container only).

TEST INFRASTRUCTURE.  This script imports the reference's
`src/clip_whisper/models/{modality_connector,clip_whisper_model}.py` by file path from
/root/reference (SURVEY.md Appendix A recipe: stub `peft`, bypass `ClipWhisperModel.__init__`
because it fetches processors/checkpoints by name) and drives its own `encode`, `forward`
(train and eval branches) and `generate` on config-instantiated `transformers` 5.15.0 modules
loaded with the deterministic weights of oracle/weights.py.  The outputs are committed as
small fixtures; /root/reference never travels to the GPU box and nothing in tests/ reads it.

It also checks the CPU restatement (oracle/avsr_oracle.py) against every vector it writes and
fails if they disagree, so a committed fixture always means "oracle == reference here".

LoRA: `peft` is not installed, so the adapter is a build-owned `nn.Module` that follows the
peft `lora.Linear` definition (out = base(x) + B(A(x)) * alpha/r) inserted into the HF Llama
attention projections -- parity unpinned against real peft.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import avsr_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402

REF = "/root/reference/src/clip_whisper/models"
OUT = os.path.join(ROOT, "tests", "golden")


def load_reference():
    peft = types.ModuleType("peft")
    peft.LoraConfig = object
    peft.get_peft_model = lambda m, c: m
    sys.modules["peft"] = peft
    pkg = types.ModuleType("refmodels")
    pkg.__path__ = [REF]
    sys.modules["refmodels"] = pkg

    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        mod.__package__ = "refmodels"
        spec.loader.exec_module(mod)
        return mod

    load("refmodels.modality_connector", os.path.join(REF, "modality_connector.py"))
    return load("refmodels.clip_whisper_model", os.path.join(REF, "clip_whisper_model.py"))


class LoraLinear(nn.Module):
    """peft lora.Linear restated (dropout p=0)."""

    def __init__(self, base: nn.Linear, A, B, scale):
        super().__init__()
        self.base = base
        self.lora_A = nn.Parameter(A.clone())
        self.lora_B = nn.Parameter(B.clone())
        self.scale = scale
        base.weight.requires_grad_(False)

    def forward(self, x):
        return self.base(x) + (x @ self.lora_A.T) @ self.lora_B.T * self.scale


def build_reference_model(cw, cfg: Wt.ModelCfg, W, freeze_encoders=True):
    from transformers import (CLIPVisionConfig, CLIPVisionModel, LlamaConfig, LlamaForCausalLM,
                              WhisperConfig, WhisperModel)
    wc, cc, lc = cfg.whisper, cfg.clip, cfg.llama
    whisper = WhisperModel(WhisperConfig(
        d_model=wc.d_model, encoder_layers=wc.layers, decoder_layers=1, encoder_attention_heads=wc.heads,
        decoder_attention_heads=wc.heads, encoder_ffn_dim=wc.ffn, decoder_ffn_dim=wc.ffn, num_mel_bins=wc.n_mels,
        max_source_positions=wc.n_ctx, vocab_size=100, pad_token_id=0, bos_token_id=1, eos_token_id=2,
        decoder_start_token_id=1))
    clip = CLIPVisionModel(CLIPVisionConfig(
        hidden_size=cc.hidden, intermediate_size=cc.mlp, num_hidden_layers=cc.layers,
        num_attention_heads=cc.heads, image_size=cc.image, patch_size=cc.patch, layer_norm_eps=cc.eps))
    llm = LlamaForCausalLM(LlamaConfig(
        hidden_size=lc.hidden, intermediate_size=lc.ffn, num_hidden_layers=lc.layers,
        num_attention_heads=lc.heads, num_key_value_heads=(lc.kv_heads or lc.heads), vocab_size=lc.vocab,
        rms_norm_eps=lc.eps, max_position_embeddings=4096, rope_theta=lc.theta,
        bos_token_id=1, eos_token_id=2, pad_token_id=None, tie_word_embeddings=False))
    res = whisper.load_state_dict(W["whisper"], strict=False)
    assert not res.unexpected_keys and all(k.startswith("decoder.") for k in res.missing_keys), res
    res = clip.load_state_dict(W["clip"], strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "position_ids" not in k], res
    res = llm.load_state_dict(W["llama"], strict=False)
    assert not res.unexpected_keys and not [k for k in res.missing_keys if "rotary" not in k and "inv_freq" not in k], res
    for p in llm.parameters():
        p.requires_grad_(False)
    for i, layer in enumerate(llm.model.layers):
        for nm in cfg.lora.targets:
            base = getattr(layer.self_attn, nm)
            setattr(layer.self_attn, nm, LoraLinear(base, W["lora"][f"layers.{i}.{nm}.lora_A"],
                                                    W["lora"][f"layers.{i}.{nm}.lora_B"], cfg.lora.scale))
    M = cw.ClipWhisperModel
    m = M.__new__(M)
    nn.Module.__init__(m)
    m.device = "cpu"; m.use_fp16 = False; m.use_4bit = False
    m.freeze_encoders = freeze_encoders; m.freeze_llm = False
    m.modality = "both"; m.max_seq_len = cfg.max_seq_len; m.fusion_scale = cfg.fusion_scale
    m.connector_type = "simple"; m.dtype = torch.float32
    m.whisper, m.clip, m.llm = whisper, clip, llm
    m.tokenizer = types.SimpleNamespace(pad_token_id=cfg.pad_token_id)
    m.audio_dim, m.video_dim, m.llm_dim = wc.d_model, cc.hidden, m._get_llm_dim()
    m._setup_projections()
    m.audio_connector.load_state_dict(W["audio_connector"])
    m.video_connector.load_state_dict(W["video_connector"])
    return m


def maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


def check(name, ref, ora, tol):
    d = maxdiff(ref, ora)
    print(f"  oracle vs reference  {name:<28s} max|diff| = {d:.3e}  (tol {tol:g})")
    assert d <= tol, f"oracle disagrees with the reference on {name}: {d}"


def golden_glue(cw):
    """G1: _pad_or_truncate / _adaptive_projection / _adapt_mask in isolation."""
    M = cw.ClipWhisperModel
    m = M.__new__(M)
    nn.Module.__init__(m)
    g = torch.Generator().manual_seed(7)
    out = {}
    x = torch.randn(2, 37, 8, generator=g)
    out["pt_in"] = x
    for T in (20, 37, 50):
        r = m._pad_or_truncate(x, T)
        out[f"pt_{T}"] = r
        check(f"pad_or_truncate->{T}", r, O.pad_or_truncate(x, T), 0)
    for L, T in ((544, 256), (1532, 256), (300, 256), (257, 256)):
        x = torch.randn(2, L, 8, generator=g)
        m.train()
        r = m._adaptive_projection(x, T)
        out[f"pool_{L}_{T}_in"] = x; out[f"pool_{L}_{T}"] = r
        check(f"adaptive_pool {L}->{T}", r, O.adaptive_projection(x, T, True), 1e-6)
    for L, T in ((100, 256), (33, 256)):
        x = torch.randn(2, L, 8, generator=g)
        m.train(); r = m._adaptive_projection(x, T)
        out[f"interp_{L}_{T}_in"] = x; out[f"interp_train_{L}_{T}"] = r
        check(f"interp(train) {L}->{T}", r, O.adaptive_projection(x, T, True), 2e-6)
        m.eval(); r = m._adaptive_projection(x, T)
        out[f"interp_eval_{L}_{T}"] = r
        check(f"interp(eval) {L}->{T}", r, O.adaptive_projection(x, T, False), 2e-6)
    mask = torch.ones(2, 40, dtype=torch.long)
    for T in (30, 40, 64):
        r = m._adapt_mask(mask, T)
        out[f"mask_{T}"] = r
        assert torch.equal(r, O.adapt_mask(mask, T))
    np.savez_compressed(os.path.join(OUT, "g1_glue.npz"), **{k: v.numpy() for k, v in out.items()})


def golden_e2e(cw):
    """G2/G4: tiny end-to-end model through the reference's encode/forward/backward/generate."""
    cfg = Wt.tiny()
    seed = 0
    W = Wt.all_weights(cfg, seed, lora_b_std=0.05)
    B, Fr = 2, 7
    audio, video, labels, _ = Wt.synthetic_batch(cfg, B, Fr, seed=1234)
    g = torch.Generator().manual_seed(99)
    prompt = torch.randint(3, cfg.llama.vocab, (B, 50), generator=g)   # >32 so the 32-token cap is exercised
    out = {"seed": np.int64(seed), "batch_seed": np.int64(1234), "frames": np.int64(Fr), "prompt": prompt.numpy(),
           "labels": labels.numpy()}

    m = build_reference_model(cw, cfg, W)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # ---- encode (decode path: no prompt) at max_seq_len 512 and 256
        m.eval()
        with torch.no_grad():
            enc, mask = m.encode(audio, video, None)
            check("encode(av)", enc, O.encode(W, cfg, audio, video, None)[0], 2e-4)
            out["encode_av_rows"] = enc[:, ::8].numpy()          # every 8th row, [B,64,D]
            enc_a, _ = m.encode(audio, None, None)                  # modality "both" but video None -> audio only
            check("encode(a)", enc_a, O.encode(W, cfg, audio, None, None)[0], 2e-4)
            out["encode_a_rows"] = enc_a[:, ::32].numpy()
            m.modality = "video"
            enc_v, _ = m.encode(None, video, None)
            m.modality = "both"
            check("encode(v)", enc_v, O.encode(W, cfg, None, video, None)[0], 2e-4)
            out["encode_v"] = enc_v.numpy()
        # ---- train forward + backward
        m.train()
        res = m(audio=audio, video=video, prompt=prompt, labels=labels)
        res["loss"].backward()
        o_loss, o_logits, o_grads = O.train_step_grads(W, cfg, audio, video, prompt, labels)
        check("train logits", res["logits"].detach(), o_logits, 5e-4)
        check("train loss", res["loss"].detach(), o_loss, 1e-5)
        out["train_loss"] = res["loss"].detach().numpy()
        out["train_logits"] = res["logits"].detach().numpy()
        out["connector_grad_is_none"] = np.bool_(m.audio_connector.linear.weight.grad is None
                                                 and m.video_connector.linear.weight.grad is None)
        for i, layer in enumerate(m.llm.model.layers):
            for nm in cfg.lora.targets:
                mod = getattr(layer.self_attn, nm)
                for ab, p in (("A", mod.lora_A), ("B", mod.lora_B)):
                    key = f"layers.{i}.{nm}.lora_{ab}"
                    check("grad " + key, p.grad, o_grads[key], 5e-5 * max(1.0, float(p.grad.abs().max())))
                    out["grad." + key] = p.grad.numpy().copy()
        # ---- eval forward
        m.eval()
        with torch.no_grad():
            res = m(audio=audio, video=video, prompt=prompt, labels=labels)
        o = O.forward(W, cfg, audio, video, prompt, labels, training=False)
        check("eval logits", res["logits"], o["logits"], 5e-4)
        check("eval loss", res["loss"], o["loss"], 1e-5)
        out["eval_loss"] = res["loss"].numpy()
        out["eval_logits_rows"] = res["logits"][:, ::4].numpy()
        # ---- greedy generate the way scripts/clip_whisper/decode.py:266-274,544-549 drives it
        m.max_seq_len = 256
        cfg256 = Wt.tiny(); cfg256.max_seq_len = 256
        with torch.no_grad():
            ids = m.generate(audio=audio, video=video, max_new_tokens=12)
        o_ids = O.generate(W, cfg256, audio, video, None, max_new_tokens=12, eos_token_id=2)
        print("  reference greedy ids:", ids.tolist())
        assert ids.shape == o_ids.shape and torch.equal(ids, o_ids), (ids, o_ids)
        out["generate_ids"] = ids.numpy()
        # top-2 margin of the reference's own logits at each step is not available from generate();
        # record the oracle's first-step margin so GPU tests can skip margin-fragile steps
    np.savez_compressed(os.path.join(OUT, "g2_tiny_e2e.npz"), **out)
    # G4: freeze_encoders=False -> connector grads present
    m2 = build_reference_model(cw, cfg, W, freeze_encoders=False)
    m2.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r2 = m2(audio=audio, video=video, prompt=prompt, labels=labels)
        r2["loss"].backward()
    assert m2.audio_connector.linear.weight.grad is not None
    print("  G4: freeze_encoders=True -> connector grads None; False -> present  OK")


def golden_optimizer():
    """G5: 3 steps of clip_grad_norm_ + AdamW + CosineAnnealingLR as wired in
    trainer/clip_whisper_trainer.py:171-232,457-464."""
    g = torch.Generator().manual_seed(5)
    p = nn.Parameter(torch.randn(1000, generator=g))
    opt = torch.optim.AdamW([{"params": [p], "weight_decay": 0.01}], lr=5e-5, betas=(0.9, 0.95), eps=1e-8)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10)
    out = {"p0": p.detach().numpy().copy()}
    po = p.detach().clone(); mo = torch.zeros(1000); vo = torch.zeros(1000)
    for s in range(3):
        grad = torch.randn(1000, generator=g) * (3.0 if s == 0 else 0.01)
        out[f"g{s}"] = grad.numpy().copy()
        p.grad = grad.clone()
        norm = torch.nn.utils.clip_grad_norm_([p], 0.5)
        lr_used = opt.param_groups[0]["lr"]
        opt.step(); opt.zero_grad(); sched.step()
        out[f"p{s + 1}"] = p.detach().numpy().copy(); out[f"norm{s}"] = norm.numpy(); out[f"lr{s}"] = np.float64(lr_used)
        go = grad.clone()
        n2 = O.clip_grad_norm_([go], 0.5)
        O.adamw_step(po, go, mo, vo, s + 1, O.cosine_lr(5e-5, s, 10))
        check(f"adamw step {s}", p.detach(), po, 1e-7)
        assert abs(float(n2) - float(norm)) < 1e-4 * float(norm)
    np.savez_compressed(os.path.join(OUT, "g5_optimizer.npz"), **out)


def gqa_cfg():
    """tiny model with a grouped-query LLM: 4 query heads, 2 key/value heads, head_dim 64 (Llama-3 / Mistral / TinyLlama layout)."""
    cfg = Wt.tiny()
    cfg.llama = Wt.LlamaCfg(hidden=256, heads=4, layers=2, ffn=512, vocab=256, kv_heads=2)
    return cfg


def golden_gqa(cw):
    """G7: the reference's forward/backward/generate with num_key_value_heads < num_attention_heads."""
    cfg = gqa_cfg()
    seed = 3
    W = Wt.all_weights(cfg, seed, lora_b_std=0.05)
    B, Fr = 2, 5
    audio, video, labels, _ = Wt.synthetic_batch(cfg, B, Fr, seed=77)
    g = torch.Generator().manual_seed(5)
    prompt = torch.randint(3, cfg.llama.vocab, (B, 20), generator=g)
    out = {"seed": np.int64(seed), "batch_seed": np.int64(77), "frames": np.int64(Fr), "prompt": prompt.numpy()}
    m = build_reference_model(cw, cfg, W)
    assert m.llm.model.layers[0].self_attn.k_proj.base.weight.shape == (128, 256)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.train()
        res = m(audio=audio, video=video, prompt=prompt, labels=labels)
        res["loss"].backward()
        o_loss, o_logits, o_grads = O.train_step_grads(W, cfg, audio, video, prompt, labels)
        check("gqa train logits", res["logits"].detach(), o_logits, 5e-4)
        check("gqa train loss", res["loss"].detach(), o_loss, 1e-5)
        out["train_loss"] = res["loss"].detach().numpy()
        out["train_logits"] = res["logits"].detach().numpy()
        for i, layer in enumerate(m.llm.model.layers):
            for nm in cfg.lora.targets:
                mod = getattr(layer.self_attn, nm)
                for ab, p in (("A", mod.lora_A), ("B", mod.lora_B)):
                    key = f"layers.{i}.{nm}.lora_{ab}"
                    check("gqa grad " + key, p.grad, o_grads[key], 5e-5 * max(1.0, float(p.grad.abs().max())))
                    out["grad." + key] = p.grad.numpy().copy()
        m.eval()
        m.max_seq_len = 256
        cfg.max_seq_len = 256
        with torch.no_grad():
            ids = m.generate(audio=audio, video=video, max_new_tokens=10)
        o_ids = O.generate(W, cfg, audio, video, None, max_new_tokens=10, eos_token_id=2)
        print("  reference greedy ids (gqa):", ids.tolist())
        assert ids.shape == o_ids.shape and torch.equal(ids, o_ids), (ids, o_ids)
        out["generate_ids"] = ids.numpy()
    np.savez_compressed(os.path.join(OUT, "g7_tiny_gqa.npz"), **out)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    os.makedirs(OUT, exist_ok=True)
    cw = load_reference()
    print("G1 glue"); golden_glue(cw)
    print("G5 optimizer"); golden_optimizer()
    print("G2 tiny end-to-end"); golden_e2e(cw)
    print("G7 tiny grouped-query"); golden_gqa(cw)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()
