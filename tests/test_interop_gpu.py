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
