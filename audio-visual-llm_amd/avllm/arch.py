"""Architecture table + weight resolution for ClipWhisperModel (product-side; no oracle imports).

Weights, in priority order: explicit `weights=` dict of HF-named state dicts; `_provided_*` torch modules
(their state_dict()); local HuggingFace checkpoint directories (config.json + *.safetensors); otherwise a
seeded synthetic initialisation of the named architecture generated directly on the device (there are no
checkpoints offline: SURVEY.md §8c)."""
from __future__ import annotations

import glob
import json
import logging
import math
import os
from dataclasses import dataclass, field

import torch


@dataclass
class WhisperCfg:
    d_model: int = 768
    heads: int = 12
    layers: int = 12
    ffn: int = 3072
    n_mels: int = 80
    n_ctx: int = 1500


@dataclass
class ClipCfg:
    hidden: int = 768
    heads: int = 12
    layers: int = 12
    mlp: int = 3072
    image: int = 224
    patch: int = 16
    eps: float = 1e-5

    @property
    def tokens(self):
        return (self.image // self.patch) ** 2 + 1


@dataclass
class LlamaCfg:
    hidden: int = 4096
    heads: int = 32
    layers: int = 32
    ffn: int = 11008
    vocab: int = 32000
    eps: float = 1e-5
    theta: float = 10000.0
    kv_heads: int = 0           # grouped-query attention: key/value heads (0 = heads)
    rope_scaling: tuple = ()    # () = plain RoPE; (factor, low_freq_factor, high_freq_factor, original_max_position_embeddings) = "llama3" scaling

    @property
    def head_dim(self):
        return self.hidden // self.heads


@dataclass
class LoraCfg:
    r: int = 16
    alpha: float = 32.0

    @property
    def scale(self):
        return self.alpha / self.r


@dataclass
class ModelCfg:
    whisper: WhisperCfg = field(default_factory=WhisperCfg)
    clip: ClipCfg = field(default_factory=ClipCfg)
    llama: LlamaCfg = field(default_factory=LlamaCfg)
    lora: LoraCfg = field(default_factory=LoraCfg)
    max_seq_len: int = 256
    fusion_scale: float = 0.5
    pad_token_id: int = 2


WHISPER = {
    "tiny": WhisperCfg(384, 6, 4, 1536), "base": WhisperCfg(512, 8, 6, 2048), "small": WhisperCfg(768, 12, 12, 3072),
    "medium": WhisperCfg(1024, 16, 24, 4096),
    "large-v3": WhisperCfg(1280, 20, 32, 5120, n_mels=128),      # 128 mel bins (matched before "large"; BASELINE config 5)
    "large": WhisperCfg(1280, 20, 32, 5120),
}
CLIP = {
    "base-patch16": ClipCfg(768, 12, 12, 3072, 224, 16), "base-patch32": ClipCfg(768, 12, 12, 3072, 224, 32),
    "large-patch14": ClipCfg(1024, 16, 24, 4096, 224, 14),
}
LLAMA = {
    "llama-2-7b": LlamaCfg(4096, 32, 32, 11008, 32000, 1e-5, 10000.0),
    "llama-2-13b": LlamaCfg(5120, 40, 40, 13824, 32000, 1e-5, 10000.0),
    # grouped-query members of the same architecture (kv_heads < heads)
    "llama-2-70b": LlamaCfg(8192, 64, 80, 28672, 32000, 1e-5, 10000.0, 8),
    "llama-3-8b": LlamaCfg(4096, 32, 32, 14336, 128256, 1e-5, 500000.0, 8),
    "tinyllama": LlamaCfg(2048, 32, 22, 5632, 32000, 1e-5, 10000.0, 4),
    "mistral-7b": LlamaCfg(4096, 32, 32, 14336, 32000, 1e-5, 10000.0, 8),     # sliding window 4096 >= every sequence of this path
}


def _from_name(name, table, what):
    low = str(name).lower()
    for key, cfg in table.items():
        if key in low:
            return cfg
    raise ValueError(f"unknown {what} architecture '{name}': known {sorted(table)} (or pass config=/weights=)")


def _load_dir(path):
    from safetensors.torch import load_file
    sd = {}
    for f in sorted(glob.glob(os.path.join(path, "*.safetensors"))):
        sd.update(load_file(f))
    if not sd:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    return sd


def _strip(sd, prefix):
    return {k[len(prefix):] if k.startswith(prefix) else k: v for k, v in sd.items()}


class _Gen:
    def __init__(self, device, dtype, seed):
        self.device, self.dtype = device, dtype
        self.g = torch.Generator(device=device)
        self.g.manual_seed(seed)

    def n(self, shape, std, mean=0.0):
        t = torch.randn(shape, generator=self.g, device=self.device, dtype=torch.float32)
        return (t * std + mean).to(self.dtype)


def synth_whisper(c, device, dtype, seed):
    g = _Gen(device, dtype, seed + 1)
    d, f = c.d_model, c.ffn
    sd = {"encoder.conv1.weight": g.n((d, c.n_mels, 3), 1 / math.sqrt(3 * c.n_mels)), "encoder.conv1.bias": g.n((d,), 0.05),
          "encoder.conv2.weight": g.n((d, d, 3), 1 / math.sqrt(3 * d)), "encoder.conv2.bias": g.n((d,), 0.05),
          "encoder.embed_positions.weight": g.n((c.n_ctx, d), 0.1),
          "encoder.layer_norm.weight": g.n((d,), 0.1, 1.0), "encoder.layer_norm.bias": g.n((d,), 0.05)}
    for i in range(c.layers):
        p = f"encoder.layers.{i}."
        for nm, bias in (("q_proj", True), ("k_proj", False), ("v_proj", True), ("out_proj", True)):
            sd[p + f"self_attn.{nm}.weight"] = g.n((d, d), 1 / math.sqrt(d))
            if bias:
                sd[p + f"self_attn.{nm}.bias"] = g.n((d,), 0.05)
        for nm in ("self_attn_layer_norm", "final_layer_norm"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = g.n((d,), 0.1, 1.0), g.n((d,), 0.05)
        sd[p + "fc1.weight"], sd[p + "fc1.bias"] = g.n((f, d), 1 / math.sqrt(d)), g.n((f,), 0.05)
        sd[p + "fc2.weight"], sd[p + "fc2.bias"] = g.n((d, f), 1 / math.sqrt(f)), g.n((d,), 0.05)
    return sd


def synth_clip(c, device, dtype, seed):
    g = _Gen(device, dtype, seed + 2)
    d, f = c.hidden, c.mlp
    sd = {"embeddings.class_embedding": g.n((d,), 0.5),
          "embeddings.patch_embedding.weight": g.n((d, 3, c.patch, c.patch), 1 / math.sqrt(3 * c.patch * c.patch)),
          "embeddings.position_embedding.weight": g.n((c.tokens, d), 0.1),
          "pre_layrnorm.weight": g.n((d,), 0.1, 1.0), "pre_layrnorm.bias": g.n((d,), 0.05)}
    for i in range(c.layers):
        p = f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{nm}.weight"], sd[p + f"self_attn.{nm}.bias"] = g.n((d, d), 1 / math.sqrt(d)), g.n((d,), 0.05)
        for nm in ("layer_norm1", "layer_norm2"):
            sd[p + nm + ".weight"], sd[p + nm + ".bias"] = g.n((d,), 0.1, 1.0), g.n((d,), 0.05)
        sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = g.n((f, d), 1 / math.sqrt(d)), g.n((f,), 0.05)
        sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = g.n((d, f), 1 / math.sqrt(f)), g.n((d,), 0.05)
    return sd


def synth_llama(c, device, dtype, seed):
    g = _Gen(device, dtype, seed + 3)
    d, f = c.hidden, c.ffn
    sd = {"model.embed_tokens.weight": g.n((c.vocab, d), 0.5), "model.norm.weight": g.n((d,), 0.1, 1.0),
          "lm_head.weight": g.n((c.vocab, d), 1 / math.sqrt(d))}
    for i in range(c.layers):
        p = f"model.layers.{i}."
        dkv = (c.kv_heads or c.heads) * c.head_dim           # grouped-query attention: k/v project to kv_heads*head_dim
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[p + f"self_attn.{nm}.weight"] = g.n((dkv if nm in ("k_proj", "v_proj") else d, d), 1 / math.sqrt(d))
        sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"] = g.n((f, d), 1 / math.sqrt(d)), g.n((f, d), 1 / math.sqrt(d))
        sd[p + "mlp.down_proj.weight"] = g.n((d, f), 1 / math.sqrt(f))
        sd[p + "input_layernorm.weight"], sd[p + "post_attention_layernorm.weight"] = g.n((d,), 0.1, 1.0), g.n((d,), 0.1, 1.0)
    return sd


def synth_lora(c, l, device, seed):
    """peft init_lora_weights="gaussian": A ~ N(0, 1/r), B = 0; then x0.01 (clip_whisper_model.py:973-1000)."""
    g = _Gen(device, torch.float32, seed + 4)
    sd = {}
    for i in range(c.layers):
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            sd[f"layers.{i}.{nm}.lora_A"] = g.n((l.r, c.hidden), (1.0 / l.r) * 0.01)
            rows = (c.kv_heads or c.heads) * c.head_dim if nm in ("k_proj", "v_proj") else c.hidden
            sd[f"layers.{i}.{nm}.lora_B"] = torch.zeros(rows, l.r, device=device)
    return sd


def _cfg_from_hf_dir(path, kind):
    c = json.load(open(os.path.join(path, "config.json")))
    if kind == "whisper":
        return WhisperCfg(c["d_model"], c["encoder_attention_heads"], c["encoder_layers"], c["encoder_ffn_dim"],
                          c.get("num_mel_bins", 80), c.get("max_source_positions", 1500))
    if kind == "clip":
        v = c.get("vision_config", c)
        return ClipCfg(v["hidden_size"], v["num_attention_heads"], v["num_hidden_layers"], v["intermediate_size"],
                       v.get("image_size", 224), v["patch_size"], v.get("layer_norm_eps", 1e-5))
    rs = _rope_scaling(c.get("rope_scaling") or (c.get("rope_parameters") or {}))
    if c.get("head_dim") not in (None, c["hidden_size"] // c["num_attention_heads"]):
        raise NotImplementedError("head_dim != hidden_size / num_attention_heads is not implemented")
    rope = c.get("rope_theta", (c.get("rope_parameters") or {}).get("rope_theta", 10000.0))
    kvh = c.get("num_key_value_heads", c["num_attention_heads"])
    return LlamaCfg(c["hidden_size"], c["num_attention_heads"], c["num_hidden_layers"], c["intermediate_size"], c["vocab_size"],
                    c.get("rms_norm_eps", 1e-5), rope, 0 if kvh == c["num_attention_heads"] else kvh, rs)


def _rope_scaling(scaling):
    """config.json `rope_scaling` -> LlamaCfg.rope_scaling.  "llama3" (Llama-3.1 / 3.2, e.g. the reference decode.py's default
    checkpoints/Llama-3.2-1B) is implemented; other kinds (linear, dynamic, yarn, longrope) are refused."""
    kind = scaling.get("rope_type", scaling.get("type", "default")) if scaling else "default"
    if kind in ("default", None):
        return ()
    if kind == "llama3":
        return (float(scaling["factor"]), float(scaling["low_freq_factor"]), float(scaling["high_freq_factor"]), int(scaling["original_max_position_embeddings"]))
    raise NotImplementedError(f"rope scaling '{kind}' is not implemented: plain RoPE and the llama3 rule only")


def weights_from_reference_state_dict(sd):
    """`model_state_dict` of a checkpoint written by the REFERENCE trainer (trainer/clip_whisper_trainer.py:752-760: the full
    ClipWhisperModel.state_dict(), frozen encoders and LLM included, LoRA under peft's key names) -> the `weights=` dict of ClipWhisperModel:
    {"whisper", "clip", "llama"} in HuggingFace naming, {"lora"} as `layers.N.<module>.lora_A|B`, and the two connector state dicts.
    Sub-dicts that the checkpoint does not carry are simply absent (decode.py-style checkpoints hold the connectors only)."""
    W = {"whisper": {}, "clip": {}, "llama": {}, "lora": {}, "audio_connector": {}, "video_connector": {}}
    for k, v in sd.items():
        if k.startswith("whisper."):
            W["whisper"][k[len("whisper."):]] = v
        elif k.startswith("clip."):
            W["clip"][k[len("clip."):]] = v
        elif k.startswith("audio_connector."):
            W["audio_connector"][k[len("audio_connector."):]] = v
        elif k.startswith("video_connector."):
            W["video_connector"][k[len("video_connector."):]] = v
        elif k.startswith("llm."):
            n = k[len("llm."):]
            if n.startswith("base_model.model."):                      # peft: PeftModel.base_model (LoraModel) .model
                n = n[len("base_model.model."):]
            if ".lora_A." in n or ".lora_B." in n:                        # model.layers.N.self_attn.q_proj.lora_A.default.weight
                parts = n.split(".")
                i, mod = parts[parts.index("layers") + 1], parts[parts.index("self_attn") + 1]
                W["lora"][f"layers.{i}.{mod}.{'lora_A' if '.lora_A.' in n else 'lora_B'}"] = v
            else:
                W["llama"][n.replace(".base_layer.", ".")] = v            # peft keeps the frozen nn.Linear as `base_layer`
    return {k: v for k, v in W.items() if v}


def resolve_arch(llm_path, whisper_model, clip_model, config, weights, seed, lora_r, lora_alpha, use_lora, p_llm, p_whisper,
                 p_clip, device, dtype, synthetic_weights=False):
    """-> (ModelCfg, weights).  A component's tensors come from `weights[kind]`, a `_provided_*` module, or a local HF directory with
    safetensors.  Seeded random tensors of the named architecture are an explicit opt-in (`synthetic_weights=True`: bench, tests, the
    scripts' --synthetic-weights); without it a path that is not a local checkpoint raises FileNotFoundError instead of silently
    training / decoding a random model."""
    W = dict(weights or {})
    cfg = config
    if cfg is None:
        parts = {}
        for kind, path, table, prov in (("whisper", whisper_model, WHISPER, p_whisper), ("clip", clip_model, CLIP, p_clip),
                                        ("llama", llm_path, LLAMA, p_llm)):
            if prov is not None and hasattr(prov, "config"):
                hc = prov.config
                if kind == "whisper":
                    parts[kind] = WhisperCfg(hc.d_model, hc.encoder_attention_heads, hc.encoder_layers, hc.encoder_ffn_dim,
                                             hc.num_mel_bins, hc.max_source_positions)
                elif kind == "clip":
                    hc = getattr(hc, "vision_config", hc)
                    parts[kind] = ClipCfg(hc.hidden_size, hc.num_attention_heads, hc.num_hidden_layers, hc.intermediate_size,
                                          hc.image_size, hc.patch_size, hc.layer_norm_eps)
                else:
                    kvh = getattr(hc, "num_key_value_heads", None) or hc.num_attention_heads
                    rope = getattr(hc, "rope_theta", None) or (getattr(hc, "rope_parameters", None) or {}).get("rope_theta", 10000.0)
                    parts[kind] = LlamaCfg(hc.hidden_size, hc.num_attention_heads, hc.num_hidden_layers, hc.intermediate_size,
                                           hc.vocab_size, hc.rms_norm_eps, rope, 0 if kvh == hc.num_attention_heads else kvh,
                                           _rope_scaling(getattr(hc, "rope_scaling", None) or {}))
            elif isinstance(path, str) and os.path.isdir(path) and os.path.exists(os.path.join(path, "config.json")):
                parts[kind] = _cfg_from_hf_dir(path, kind)
            else:
                parts[kind] = _from_name(path, table, kind)
        cfg = ModelCfg(parts["whisper"], parts["clip"], parts["llama"], LoraCfg(lora_r, float(lora_alpha)))
    else:
        cfg.lora = LoraCfg(lora_r, float(lora_alpha)) if not hasattr(cfg, "lora") or cfg.lora is None else cfg.lora
    for kind, path, prov, synth, c in (("whisper", whisper_model, p_whisper, synth_whisper, cfg.whisper),
                                       ("clip", clip_model, p_clip, synth_clip, cfg.clip),
                                       ("llama", llm_path, p_llm, synth_llama, cfg.llama)):
        if kind in W:
            continue
        if prov is not None:
            sd = {k: v.detach() for k, v in prov.state_dict().items()}
        elif isinstance(path, str) and os.path.isdir(path) and glob.glob(os.path.join(path, "*.safetensors")):
            sd = _load_dir(path)
        elif synthetic_weights:
            logging.warning("%s: no local checkpoint at %r -- using SEEDED RANDOM weights of that architecture (synthetic_weights=True)", kind, path)
            sd = synth(c, device, dtype, seed)
        else:
            raise FileNotFoundError(f"{kind}: {path!r} is not a local directory with *.safetensors (there is no network access to fetch it by "
                                    "name); pass a checkpoint directory, `_provided_*` modules or `weights=`, or opt in to seeded random "
                                    "weights with synthetic_weights=True / --synthetic-weights")
        if kind == "whisper":
            sd = _strip(sd, "model.")
        if kind == "clip":
            sd = _strip(sd, "vision_model.")
        if kind == "llama" and "lm_head.weight" not in sd and "model.embed_tokens.weight" in sd:
            sd["lm_head.weight"] = sd["model.embed_tokens.weight"]          # tie_word_embeddings checkpoints store one copy
        W[kind] = sd
    if use_lora and "lora" not in W:
        W["lora"] = synth_lora(cfg.llama, cfg.lora, device, seed)
    return cfg, W
