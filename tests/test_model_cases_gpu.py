"""Edge cases of the hot path on the GPU against the CPU oracle on the same seeded inputs (fp32 mode, so the bar is the
north-star one: logits within 1e-3, loss within 1e-4, LoRA gradients within 2e-4 relative, greedy tokens identical).
Covers what ClipWhisperModel.forward/encode/generate branch on (clip_whisper_model.py:407-462, :577-598, :621-736,
:1280-1294): single-modality inputs, sequences shorter and longer than the label length (linear interpolation with
align_corners / adaptive average pooling), no prompt, over-long prompt, B=1 and one frame, rows with no valid label."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import avsr_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402
from test_model_gpu import make_model  # noqa: E402


@pytest.fixture(scope="module")
def setup():
    oc = Wt.tiny()
    W = Wt.all_weights(oc, 11, lora_b_std=0.05)
    return oc, W


def batch(oc, B, frames, seed, prompt_len=50):
    audio, video, labels, _ = Wt.synthetic_batch(oc, B, frames, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    prompt = torch.randint(3, oc.llama.vocab, (B, prompt_len), generator=g)
    return audio, video, labels, prompt


def check_train(dev, oc, W, m, audio, video, prompt, labels, max_seq_len):
    cfg = copy.copy(oc)
    cfg.max_seq_len = max_seq_len
    ref_loss, ref_logits, ref_grads = O.train_step_grads(W, cfg, audio, video, prompt, labels)
    m.train()
    to = lambda t: None if t is None else t.to(dev)
    out = m(audio=to(audio), video=to(video), prompt=to(prompt), labels=to(labels))
    assert out["logits"].shape == ref_logits.shape
    dl = (out["logits"].float().cpu() - ref_logits).abs().max().item()
    assert dl < 1e-3, dl
    assert abs(float(out["loss"].detach()) - float(ref_loss)) < 1e-4
    m.lora_param.grad = None
    out["loss"].backward()
    for k, gr in m.llm_engine.lora_views(m.lora_param.grad).items():
        ref = ref_grads[k]
        assert (gr.cpu() - ref).abs().max() <= 2e-4 * max(1e-3, float(ref.abs().max())) + 1e-7, (k, (gr.cpu() - ref).abs().max())


@pytest.mark.parametrize("max_seq_len,frames", [(64, 5), (1536, 7), (224, 300)])
def test_both_modalities_short_and_long_sequences(dev, setup, max_seq_len, frames):
    """P+L = 96 (interpolated up to the 256 labels), 1532 (pooled down), and the no-op case P+L = 256 with more video
    frames than the cap (300 > 224: video truncated, audio truncated)."""
    oc, W = setup
    audio, video, labels, prompt = batch(oc, 2, frames, seed=5)
    m = make_model(oc, W, "fp32", max_seq_len=max_seq_len)
    check_train(dev, oc, W, m, audio, video, prompt, labels, max_seq_len)


def test_audio_only_and_video_only_training(dev, setup):
    oc, W = setup
    audio, video, labels, prompt = batch(oc, 2, 9, seed=6)
    m = make_model(oc, W, "fp32")
    m.modality = "audio"                                  # 1500 frames + prompt -> pooled to 256 (no max_seq_len cap, :428-437)
    check_train(dev, oc, W, m, audio, None, prompt, labels, 512)
    m.modality = "video"                                  # 9 frames + 32 prompt tokens = 41 -> interpolated to 256
    check_train(dev, oc, W, m, None, video, prompt, labels, 512)
    m.modality = "both"                                   # modality "both" with one input missing falls back to the other (:438-444)
    check_train(dev, oc, W, m, audio, None, prompt, labels, 512)


def test_no_prompt_single_clip_single_frame(dev, setup):
    oc, W = setup
    audio, video, labels, prompt = batch(oc, 1, 1, seed=7)
    m = make_model(oc, W, "fp32")
    check_train(dev, oc, W, m, audio, video, None, labels, 512)
    check_train(dev, oc, W, m, audio, video, prompt[:, :3], labels, 512)          # 3-token prompt: P+L = 515 -> pooled


def test_rows_without_valid_labels_and_full_rows(dev, setup):
    """Row 0 all pad (contributes nothing), row 1 no pad at all, row 2 ordinary: the mean runs over valid targets only."""
    oc, W = setup
    audio, video, labels, prompt = batch(oc, 3, 4, seed=8)
    labels[0, :] = oc.pad_token_id
    g = torch.Generator().manual_seed(3)
    labels[1, :] = torch.randint(3, oc.llama.vocab, (labels.shape[1],), generator=g)
    m = make_model(oc, W, "fp32")
    check_train(dev, oc, W, m, audio, video, prompt, labels, 512)


def test_eval_forward_and_generate_single_modality(dev, setup):
    oc, W = setup
    audio, video, labels, prompt = batch(oc, 2, 6, seed=9)
    m = make_model(oc, W, "fp32").eval()
    cfg = copy.copy(oc)
    # eval: labels padded with -100 up to P+L (video only: 6 + 32 = 38 < 256 -> labels truncated instead, :586-598)
    ref = O.forward(W, cfg, None, video, prompt, labels, training=False)
    m.modality = "video"
    out = m(audio=None, video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    assert out["logits"].shape == ref["logits"].shape == (2, 38, oc.llama.vocab)
    assert (out["logits"].float().cpu() - ref["logits"]).abs().max() < 1e-3
    assert abs(float(out["loss"]) - float(ref["loss"])) < 1e-4
    toks = m.generate(video=video.to(dev), prompt=prompt.to(dev), max_new_tokens=6)
    ref_t = O.generate(W, cfg, None, video, prompt, max_new_tokens=6)
    assert torch.equal(toks.cpu(), ref_t)
    m.modality = "both"
    toks = m.generate(audio=audio.to(dev), prompt=None, max_new_tokens=4)          # generate() switches modality by its inputs (:1280-1294)
    ref_t = O.generate(W, cfg, audio, None, None, max_new_tokens=4)
    assert torch.equal(toks.cpu(), ref_t)
    assert m.modality == "both"


def test_grouped_query_llm_golden_and_oracle(dev, golden_dir):
    """Grouped-query LLM (4 query heads, 2 key/value heads, head_dim 64): the REFERENCE's own outputs (g7) in fp32 mode,
    bf16 sanity, greedy tokens identical, KV cache sized by the key/value heads."""
    import numpy as np
    from oracle.make_golden import gqa_cfg
    g = np.load(f"{golden_dir}/g7_tiny_gqa.npz")
    cfg = gqa_cfg()
    W = Wt.all_weights(cfg, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, _ = Wt.synthetic_batch(cfg, 2, int(g["frames"]), seed=int(g["batch_seed"]))
    prompt = torch.from_numpy(g["prompt"])
    m = make_model(cfg, W, "fp32").train()
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    assert (out["logits"].float().cpu() - torch.from_numpy(g["train_logits"])).abs().max() < 1e-3
    assert abs(float(out["loss"].detach()) - float(g["train_loss"])) < 1e-4
    m.lora_param.grad = None
    out["loss"].backward()
    views = m.llm_engine.lora_views(m.lora_param.grad)
    assert views["layers.0.k_proj.lora_B"].shape == (128, cfg.lora.r)
    for k, gr in views.items():
        ref = torch.from_numpy(g["grad." + k])
        assert (gr.cpu() - ref).abs().max() <= 2e-4 * max(1e-3, float(ref.abs().max())) + 1e-7, (k, (gr.cpu() - ref).abs().max())
    kc, _ = m.llm_engine.alloc_cache(2, 8)
    assert kc.shape[-1] == 128
    m256 = make_model(cfg, W, "fp32", max_seq_len=256).eval()
    m256.eos_token_id = 2
    ids = m256.generate(audio=audio.to(dev), video=video.to(dev), max_new_tokens=10)
    assert torch.equal(ids.cpu(), torch.from_numpy(g["generate_ids"]))
    # bf16 engine: same step within bf16 tolerances (lora_dropout = 0 here; the fused-dropout paths have their own test against the oracle
    # with the same masks: tests/test_pin_bf16_gpu.py)
    m16 = make_model(cfg, W, "bf16").train()
    o16 = m16(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    ref = torch.from_numpy(g["train_logits"])
    err = (o16["logits"].float().cpu() - ref).abs()
    assert err.max() < 6e-2 * max(1.0, float(ref.abs().max())) and err.mean() < 1e-2, (err.max(), err.mean())
    m16.lora_param.grad = None
    o16["loss"].backward()
    num = den = 0.0
    for k, gr in m16.llm_engine.lora_views(m16.lora_param.grad).items():
        r_ = torch.from_numpy(g["grad." + k])
        num += float(((gr.cpu() - r_) ** 2).sum()); den += float((r_ ** 2).sum())
    assert (num / den) ** 0.5 < 5e-2, (num / den) ** 0.5


def test_config5_family_geometry_vs_oracle(dev):
    """BASELINE config 5's encoder family at test size: a 128-mel Whisper (large-v3; the reference's 80-bin guard :1074 is lifted), a
    patch-14 CLIP whose 224-pixel frames give 257 tokens per frame (the ViT-L/14 sequence length: general attention path, not the
    197-token one-shot kernel), and a grouped-query LLM (Mistral layout).  Whole train step against the oracle, fp32 bars."""
    oc = Wt.tiny()
    oc.whisper = Wt.WhisperCfg(d_model=128, heads=2, layers=2, ffn=256, n_mels=128)
    oc.clip = Wt.ClipCfg(hidden=128, heads=2, layers=2, mlp=256, image=224, patch=14)
    oc.llama = Wt.LlamaCfg(hidden=256, heads=4, layers=2, ffn=512, vocab=256, kv_heads=2)
    assert oc.clip.tokens == 257
    W = Wt.all_weights(oc, 21, lora_b_std=0.05)
    audio, video, labels, prompt = batch(oc, 2, 3, seed=9)
    assert audio.shape[1] == 128
    m = make_model(oc, W, "fp32")
    check_train(dev, oc, W, m, audio, video, prompt, labels, 512)
    with pytest.raises(ValueError, match="128"):
        m.encode_audio(audio[:, :80].to(dev))                    # an 80-bin tensor is now the wrong shape for this encoder
    m16 = make_model(oc, W, "bf16").train()
    to = lambda t: t.to(dev)
    o16 = m16(audio=to(audio), video=to(video), prompt=to(prompt), labels=to(labels))
    cfg = copy.copy(oc); cfg.max_seq_len = 512
    ref_loss, ref_logits, _ = O.train_step_grads(W, cfg, audio, video, prompt, labels)
    err = (o16["logits"].float().cpu() - ref_logits).abs()
    assert err.max() < 6e-2 and err.mean() < 1e-2 and abs(float(o16["loss"].detach()) - float(ref_loss)) < 2e-2


def test_llama3_rope_scaling_train_and_generate_vs_oracle(dev):
    """Llama-3.1 / 3.2 checkpoints carry `rope_scaling: {rope_type: llama3, ...}` (the reference's decode.py defaults to Llama-3.2-1B): the
    frequency rule lives in the cos/sin table every RoPE consumer of the engine reads (training forward, the attention backward's fused
    inverse rotation, prefill, the fused token step).  A grouped-query model with a small original context (so that most frequencies are
    stretched or interpolated inside 256 positions): whole train step vs the oracle at the fp32 bars, greedy tokens identical, and a
    negative control -- the same weights WITHOUT the rule differ by far more than the bar.  The oracle's rule is pinned against
    transformers in tests/test_oracle.py."""
    oc = Wt.tiny()
    rs = (8.0, 1.0, 4.0, 64)
    oc.llama = Wt.LlamaCfg(hidden=256, heads=4, layers=2, ffn=512, vocab=256, kv_heads=2, theta=500000.0, rope_scaling=rs)
    W = Wt.all_weights(oc, 23, lora_b_std=0.05)
    audio, video, labels, prompt = batch(oc, 2, 3, seed=12)
    m = make_model(oc, W, "fp32")
    assert m.llm_engine.desc.rope_orig_ctx == 64 and abs(m.llm_engine.desc.rope_factor - 8.0) < 1e-6
    check_train(dev, oc, W, m, audio, video, prompt, labels, 512)
    cfg = copy.copy(oc); cfg.max_seq_len = 256
    m.max_seq_len = 256
    ids = m.eval().generate(audio=audio.to(dev), video=video.to(dev), max_new_tokens=10).cpu()
    ref = O.generate(W, cfg, audio, video, None, max_new_tokens=10, eos_token_id=m.eos_token_id)
    assert torch.equal(ids, ref), (ids, ref)
    # negative control: plain RoPE on the same weights
    plain = copy.copy(oc); plain.llama = Wt.LlamaCfg(**{**vars(oc.llama), "rope_scaling": ()}); plain.max_seq_len = 512
    _, logits_plain, _ = O.train_step_grads(W, plain, audio, video, prompt, labels)
    cfg512 = copy.copy(oc); cfg512.max_seq_len = 512
    _, logits_scaled, _ = O.train_step_grads(W, cfg512, audio, video, prompt, labels)
    assert (logits_plain - logits_scaled).abs().max() > 2e-2


def test_same_seed_same_model(dev):
    """Offline (synthetic) construction is a pure function of `seed`, connectors included (the reference draws its connector init from the
    global RNG): two builds agree bit for bit whatever the global RNG state, a different seed gives a different model."""
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    cfg = lambda: ModelCfg(WhisperCfg(128, 2, 2, 256), ClipCfg(128, 2, 2, 256, 48, 16), LlamaCfg(256, 2, 2, 512, 256), LoraCfg(16, 32.0))
    oc = Wt.tiny()
    audio, video, labels, prompt = batch(oc, 2, 3, seed=4)
    outs = []
    for seed, gseed in ((3, 10), (3, 11), (4, 10)):
        torch.manual_seed(gseed)
        m = ClipWhisperModel(device=dev, max_seq_len=512, config=cfg(), precision="fp32", seed=seed, synthetic_weights=True).eval()
        with torch.no_grad():
            outs.append(m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))["logits"].clone())
    assert torch.equal(outs[0], outs[1])
    assert not torch.equal(outs[0], outs[2])


def test_deep_connector_matches_reference_fixture(dev, golden_dir):
    """`connector_type="deep"` (and any unknown name: the reference's factory falls back to it, modality_connector.py:394-396) against the
    output of the REFERENCE's DeepModalityConnector on the same parameters (tests/golden/g9_deep_connector.npz); conv/attention/adaptive
    are refused by name."""
    import numpy as np
    from avllm.connector import DeepModalityConnector, create_modality_connector
    g = np.load(f"{golden_dir}/g9_deep_connector.npz")
    for tag, name in (("a", "deep"), ("b", "perceiver")):
        layers = int(g[tag + ".layers"])
        for dtype, tol in ((torch.float32, 1e-4), (torch.bfloat16, 6e-2)):
            c = create_modality_connector(name, 64, 128, device=dev, dtype=dtype, num_layers=layers)
            assert isinstance(c, DeepModalityConnector)
            c.load_state_dict({k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".sd.")})
            y = c(torch.from_numpy(g[tag + ".x"]).to(dev)).float().cpu()
            assert y.shape == (2, 9, 128) and (y - torch.from_numpy(g[tag + ".y"])).abs().max() < tol


def test_conv_attention_adaptive_connectors_match_reference_fixture(dev, golden_dir):
    """`connector_type` conv / attention / adaptive (modality_connector.py:111-380) against outputs of the REFERENCE's own classes in eval mode
    on the same parameters (tests/golden/g10_connectors.npz, written by oracle/make_golden_connector.py): the reference's state dict loads as
    it is (same module tree), the arithmetic is avllm_gemm / im2col / groupnorm / layernorm / attention.  adapt_long has 600 tokens, i.e. the
    strided-convolution branch (T > 512 -> 150 rows)."""
    import numpy as np
    from avllm.connector import create_modality_connector
    g = np.load(f"{golden_dir}/g10_connectors.npz")
    for tag, name in (("conv", "conv"), ("attn", "attention"), ("adapt_short", "adaptive"), ("adapt_long", "adaptive")):
        x, y = torch.from_numpy(g[tag + ".x"]), torch.from_numpy(g[tag + ".y"])
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".sd.")}
        for dtype, tol in ((torch.float32, 2e-4), (torch.bfloat16, 1e-1)):
            kw = {"max_seq_len": sd["pos_encoder.pe"].shape[0]} if name == "adaptive" else {}
            c = create_modality_connector(name, x.shape[-1], y.shape[-1], device=dev, dtype=dtype, **kw)
            missing = c.load_state_dict(sd, strict=True)
            out = c(x.to(dev)).float().cpu()
            assert out.shape == y.shape, (tag, out.shape, y.shape)
            err = (out - y).abs().max().item()
            assert err < tol, (tag, str(dtype), err)
    # the model accepts them by name (the reference's --connector_type choices, scripts/clip_whisper/train.py:76-78)
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    cfg = ModelCfg(WhisperCfg(128, 2, 2, 256), ClipCfg(128, 2, 2, 256, 48, 16), LlamaCfg(256, 2, 2, 512, 256), LoraCfg(16, 32.0))
    oc = Wt.tiny()
    audio, video, labels, prompt = batch(oc, 2, 3, seed=4)
    for name in ("conv", "attention", "adaptive"):
        m = ClipWhisperModel(device=dev, max_seq_len=512, config=cfg, precision="fp32", seed=3, synthetic_weights=True, connector_type=name).train()
        out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
        assert torch.isfinite(out["loss"]).item() and out["logits"].shape[:2] == (2, 256), name
