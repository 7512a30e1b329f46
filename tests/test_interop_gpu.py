"""Checkpoint interop: HuggingFace-format directories (config.json + model.safetensors, as `save_pretrained`
writes them) and live HF modules handed over the reference's `_provided_*` arguments load into the engines and reproduce the
oracle run on the very same tensors.  transformers is only used to WRITE the checkpoints; skipped where it is not importable."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import avsr_oracle as O  # noqa: E402
from oracle import weights as Wt  # noqa: E402

tf = pytest.importorskip("transformers")


def hf_models(cfg):
    wc, cc, lc = cfg.whisper, cfg.clip, cfg.llama
    torch.manual_seed(7)
    whisper = tf.WhisperModel(tf.WhisperConfig(
        d_model=wc.d_model, encoder_layers=wc.layers, decoder_layers=1, encoder_attention_heads=wc.heads, decoder_attention_heads=wc.heads,
        encoder_ffn_dim=wc.ffn, decoder_ffn_dim=wc.ffn, num_mel_bins=wc.n_mels, max_source_positions=wc.n_ctx, vocab_size=100,
        pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1))
    clip = tf.CLIPVisionModel(tf.CLIPVisionConfig(hidden_size=cc.hidden, intermediate_size=cc.mlp, num_hidden_layers=cc.layers,
                                                  num_attention_heads=cc.heads, image_size=cc.image, patch_size=cc.patch))
    llm = tf.LlamaForCausalLM(tf.LlamaConfig(
        hidden_size=lc.hidden, intermediate_size=lc.ffn, num_hidden_layers=lc.layers, num_attention_heads=lc.heads,
        num_key_value_heads=lc.kv_heads or lc.heads, vocab_size=lc.vocab, rms_norm_eps=lc.eps, max_position_embeddings=4096,
        rope_theta=lc.theta, bos_token_id=1, eos_token_id=2, pad_token_id=None, tie_word_embeddings=False))
    return whisper.eval(), clip.eval(), llm.eval()


def oracle_weights(cfg, whisper, clip, llm, conn):
    W = {"whisper": {k[len("model."):] if k.startswith("model.") else k: v.detach() for k, v in whisper.state_dict().items()},
         "clip": {k: v.detach() for k, v in clip.state_dict().items()},
         "llama": {k: v.detach() for k, v in llm.state_dict().items()},
         "lora": Wt.lora_weights(cfg.llama, cfg.lora, 5, b_std=0.05)}
    W["whisper"] = {k: v for k, v in W["whisper"].items()}
    W.update(conn)
    return W


@pytest.mark.parametrize("how", ["directories", "provided_modules"])
def test_hf_checkpoints_load_and_match_oracle(dev, tmp_path, how):
    from avllm.model import ClipWhisperModel
    from oracle.make_golden import gqa_cfg
    cfg = gqa_cfg()                                          # grouped-query LLM, so num_key_value_heads travels through config.json
    whisper, clip, llm = hf_models(cfg)
    from avllm.tokenizer import ByteTokenizer
    kw = dict(device=dev, lora_r=cfg.lora.r, lora_alpha=cfg.lora.alpha, lora_dropout=0.0, max_seq_len=512, precision="fp32",
              _provided_tokenizer=ByteTokenizer(cfg.llama.vocab))       # the written checkpoints carry no tokenizer files
    if how == "directories":
        dirs = {}
        for name, mod in (("whisper", whisper), ("clip", clip), ("llama", llm)):
            d = tmp_path / name
            mod.save_pretrained(d, safe_serialization=True)
            dirs[name] = str(d)
        m = ClipWhisperModel(dirs["llama"], dirs["whisper"], dirs["clip"], **kw)
    else:
        m = ClipWhisperModel("provided-llama", "provided-whisper", "provided-clip", _provided_llm=llm, _provided_whisper=whisper,
                             _provided_clip=clip, **kw)
    assert m.cfg.llama.kv_heads == 2 and m.cfg.llama.heads == 4 and m.cfg.whisper.d_model == cfg.whisper.d_model
    conn = {"audio_connector": {k: v.detach().cpu() for k, v in m.audio_connector.state_dict().items()},
            "video_connector": {k: v.detach().cpu() for k, v in m.video_connector.state_dict().items()}}
    W = oracle_weights(cfg, whisper, clip, llm, conn)
    m.llm_engine.load_lora(W["lora"]); m.llm_engine.pack_lora()
    audio, video, labels, _ = Wt.synthetic_batch(cfg, 2, 4, seed=12)
    ref = O.forward(W, cfg, audio, video, None, labels, training=True)
    out = m.train()(audio=audio.to(dev), video=video.to(dev), prompt=None, labels=labels.to(dev))
    assert (out["logits"].float().cpu() - ref["logits"]).abs().max() < 1e-3
    assert abs(float(out["loss"].detach()) - float(ref["loss"])) < 1e-4


def test_reference_trainer_checkpoint_resumes_exactly(dev, golden_dir, tmp_path):
    """SURVEY.md §8f N1, both directions.
    (1) reference -> build: the checkpoint written by the reference trainer (tests/golden/n1_reference_model_best.pt) restores model,
    Adam moments and schedule position; the next forward equals the reference's, and the next optimizer step lands on the reference's
    parameters (tests/golden/n1_expected.npz holds what the reference computed from that state).
    (2) build -> reference: a checkpoint of this build passes the reference's own consumers' rules: decode.py:236-260 pulls the connectors by
    substring and load_state_dict()s them into fresh connector modules; the optimizer / scheduler dicts have torch's own layout."""
    import numpy as np
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg, weights_from_reference_state_dict
    from avllm.model import ClipWhisperModel
    from avllm.tokenizer import ByteTokenizer
    from avllm.trainer import ClipWhisperTrainer
    from oracle.make_golden_checkpoint import micro_cfg
    exp = np.load(f"{golden_dir}/n1_expected.npz")
    path = f"{golden_dir}/n1_reference_model_best.pt"
    ck = torch.load(path, map_location="cpu", weights_only=True)
    oc = micro_cfg()
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device=dev, lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=512, config=cfg,
                         weights=weights_from_reference_state_dict(ck["model_state_dict"]), precision="fp32",
                         _provided_tokenizer=ByteTokenizer(oc.llama.vocab)).train()
    tr = ClipWhisperTrainer(m, learning_rate=float(exp["lr"]), weight_decay=float(exp["wd"]), grad_clip=float(exp["clip"]),
                            total_steps=int(exp["total_steps"]), max_epochs=1, output_dir=str(tmp_path), use_graph=False)
    assert tr.load_checkpoint(path) == ck["epoch"] and tr.global_step == 2 and tr.train_losses == pytest.approx(list(exp["losses_12"]))
    st = ck["optimizer_state_dict"]["state"]
    ids = sorted(i for i in st if "exp_avg" in st[i])
    mv = m.llm_engine.lora_views(tr.m)
    assert torch.equal(mv["layers.0.q_proj.lora_A"].cpu(), st[ids[0]]["exp_avg"]) and torch.equal(mv["layers.0.o_proj.lora_B"].cpu(), st[ids[-1]]["exp_avg"])
    audio, video, labels, prompt = Wt.synthetic_batch(oc, 2, 3, seed=int(exp["seeds"][2]))
    out = m(audio=audio.to(dev), video=video.to(dev), prompt=prompt.to(dev), labels=labels.to(dev))
    assert abs(float(out["loss"].detach()) - float(exp["train_loss_3"])) < 1e-4
    assert (out["logits"].float().cpu()[:, ::8] - torch.from_numpy(exp["train_logits_3_sub"])).abs().max() < 1e-3
    tr.train_step(audio.to(dev), video.to(dev), labels.to(dev), prompt.to(dev))
    sd = m.state_dict()
    num = den = 0.0
    for k in exp.files:
        if k.startswith("after3."):
            mine, ref, before = sd[k[7:]].cpu(), torch.from_numpy(exp[k]), ck["model_state_dict"][k[7:]]
            num += float(((mine - before) - (ref - before)).pow(2).sum()); den += float((ref - before).pow(2).sum())
    assert den > 0 and (num / den) ** 0.5 < 2e-2, (num / den) ** 0.5            # the third step's UPDATE, relative L2 (Adam amplifies rounding near zero)
    # ---- (2) this build's checkpoint through the reference's consumers' rules
    tr._save_checkpoint(0, "model_final.pt")
    mine = torch.load(tmp_path / "model_final.pt", map_location="cpu", weights_only=True)
    assert set(ck) <= set(mine)
    state_dict = mine["model_state_dict"]
    audio_connector_dict = {k.replace("audio_connector.", ""): v for k, v in state_dict.items() if "audio_connector" in k}      # decode.py:237
    video_connector_dict = {k.replace("video_connector.", ""): v for k, v in state_dict.items() if "video_connector" in k}      # decode.py:238
    for d_, dim in ((audio_connector_dict, oc.whisper.d_model), (video_connector_dict, oc.clip.hidden)):
        fresh = torch.nn.Module()
        fresh.linear = torch.nn.Linear(dim, oc.llama.hidden)                  # the module tree of the reference's ModalityConnector
        fresh.load_state_dict(d_)                                              # strict, as decode.py:250,258
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros_like(v)) for k, v in state_dict.items() if "lora_" in k], lr=1e-3, betas=(0.9, 0.95))
    opt.load_state_dict(mine["optimizer_state_dict"])                          # torch's own loader accepts the layout
    assert float(opt.state_dict()["state"][0]["step"]) == 3.0
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=int(exp["total_steps"]))
    sched.load_state_dict(mine["scheduler_state_dict"])
    assert sched.last_epoch == 3
    tr2 = ClipWhisperTrainer(m, learning_rate=float(exp["lr"]), total_steps=int(exp["total_steps"]), max_epochs=1, use_graph=False)
    tr2.load_checkpoint(tmp_path / "model_final.pt")
    assert tr2.global_step == 3 and torch.equal(tr2.m, tr.m) and torch.equal(tr2.v, tr.v)
