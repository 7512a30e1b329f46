"""Model configurations and deterministic synthetic weights (TEST INFRASTRUCTURE).

No checkpoints exist offline, so every parity test and the bench use
seeded random weights at the architecture's true shapes.  Tensors are keyed by the
HuggingFace state-dict names of the modules the reference instantiates
(reference: src/clip_whisper/models/clip_whisper_model.py:864-907 `_load_whisper_model`,
`_load_clip_model`; :909-1019 `_load_llm`), so the same dictionaries load into
`transformers` modules (oracle/make_golden.py) and into the HIP engine.

Each tensor is drawn from its own `torch.Generator` seeded by crc32(name) ^ seed, so the
values do not depend on creation order and can be regenerated anywhere from the seed.
"""
from __future__ import annotations

import math
import zlib
from dataclasses import dataclass, field, asdict

import torch


@dataclass
class WhisperCfg:
    d_model: int = 768
    heads: int = 12
    layers: int = 12
    ffn: int = 3072
    n_mels: int = 80
    n_ctx: int = 1500          # max_source_positions; input is always 2*n_ctx mel frames


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
    def tokens(self) -> int:
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
    rope_scaling: tuple = ()    # () = plain RoPE; (factor, low_freq_factor, high_freq_factor, original_max_position_embeddings) = "llama3" rule

    @property
    def head_dim(self) -> int:
        return self.hidden // self.heads


@dataclass
class LoraCfg:
    r: int = 16
    alpha: float = 32.0
    targets: tuple = ("q_proj", "k_proj", "v_proj", "o_proj")

    @property
    def scale(self) -> float:
        return self.alpha / self.r


@dataclass
class ModelCfg:
    whisper: WhisperCfg = field(default_factory=WhisperCfg)
    clip: ClipCfg = field(default_factory=ClipCfg)
    llama: LlamaCfg = field(default_factory=LlamaCfg)
    lora: LoraCfg = field(default_factory=LoraCfg)
    max_seq_len: int = 512
    fusion_scale: float = 0.5
    pad_token_id: int = 2
    max_prompt_len: int = 32   # clip_whisper_model.py:469

    def to_dict(self):
        return asdict(self)


def config2() -> ModelCfg:
    """BASELINE.json configs[1]: Whisper-small + CLIP ViT-B/16 -> Llama-2-7B, LoRA r16."""
    return ModelCfg()


def tiny() -> ModelCfg:
    """Small model that keeps the head dims the HIP attention kernels are built for
    (64 for the encoders, 128 for the LLM) so the golden vectors exercise the shipped kernels."""
    return ModelCfg(
        whisper=WhisperCfg(d_model=128, heads=2, layers=2, ffn=256),
        clip=ClipCfg(hidden=128, heads=2, layers=2, mlp=256, image=48, patch=16),
        llama=LlamaCfg(hidden=256, heads=2, layers=2, ffn=512, vocab=256),
        lora=LoraCfg(r=16, alpha=32.0),
        max_seq_len=512,
    )


def _gen(name: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _randn(name, seed, shape, std, mean=0.0, dtype=torch.float32):
    t = torch.randn(shape, generator=_gen(name, seed), dtype=torch.float32)
    return (t * std + mean).to(dtype)


def whisper_weights(c: WhisperCfg, seed: int = 0) -> dict:
    d, f = c.d_model, c.ffn
    sd = {}
    P = "whisper.encoder."
    sd["encoder.conv1.weight"] = _randn(P + "conv1.w", seed, (d, c.n_mels, 3), 1.0 / math.sqrt(3 * c.n_mels))
    sd["encoder.conv1.bias"] = _randn(P + "conv1.b", seed, (d,), 0.05)
    sd["encoder.conv2.weight"] = _randn(P + "conv2.w", seed, (d, d, 3), 1.0 / math.sqrt(3 * d))
    sd["encoder.conv2.bias"] = _randn(P + "conv2.b", seed, (d,), 0.05)
    sd["encoder.embed_positions.weight"] = _randn(P + "pos", seed, (c.n_ctx, d), 0.1)
    for i in range(c.layers):
        L = f"encoder.layers.{i}."
        for nm, has_bias in (("q_proj", True), ("k_proj", False), ("v_proj", True), ("out_proj", True)):
            sd[L + f"self_attn.{nm}.weight"] = _randn("whisper." + L + nm + ".w", seed, (d, d), 1.0 / math.sqrt(d))
            if has_bias:
                sd[L + f"self_attn.{nm}.bias"] = _randn("whisper." + L + nm + ".b", seed, (d,), 0.05)
        for nm in ("self_attn_layer_norm", "final_layer_norm"):
            sd[L + nm + ".weight"] = _randn("whisper." + L + nm + ".w", seed, (d,), 0.1, 1.0)
            sd[L + nm + ".bias"] = _randn("whisper." + L + nm + ".b", seed, (d,), 0.05)
        sd[L + "fc1.weight"] = _randn("whisper." + L + "fc1.w", seed, (f, d), 1.0 / math.sqrt(d))
        sd[L + "fc1.bias"] = _randn("whisper." + L + "fc1.b", seed, (f,), 0.05)
        sd[L + "fc2.weight"] = _randn("whisper." + L + "fc2.w", seed, (d, f), 1.0 / math.sqrt(f))
        sd[L + "fc2.bias"] = _randn("whisper." + L + "fc2.b", seed, (d,), 0.05)
    sd["encoder.layer_norm.weight"] = _randn(P + "ln.w", seed, (d,), 0.1, 1.0)
    sd["encoder.layer_norm.bias"] = _randn(P + "ln.b", seed, (d,), 0.05)
    return sd


def clip_weights(c: ClipCfg, seed: int = 0) -> dict:
    d, f = c.hidden, c.mlp
    sd = {}
    sd["embeddings.class_embedding"] = _randn("clip.cls", seed, (d,), 0.5)
    sd["embeddings.patch_embedding.weight"] = _randn("clip.patch", seed, (d, 3, c.patch, c.patch),
                                                      1.0 / math.sqrt(3 * c.patch * c.patch))
    sd["embeddings.position_embedding.weight"] = _randn("clip.pos", seed, (c.tokens, d), 0.1)
    for nm in ("pre_layrnorm", "post_layernorm"):
        sd[nm + ".weight"] = _randn("clip." + nm + ".w", seed, (d,), 0.1, 1.0)
        sd[nm + ".bias"] = _randn("clip." + nm + ".b", seed, (d,), 0.05)
    for i in range(c.layers):
        L = f"encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[L + f"self_attn.{nm}.weight"] = _randn("clip." + L + nm + ".w", seed, (d, d), 1.0 / math.sqrt(d))
            sd[L + f"self_attn.{nm}.bias"] = _randn("clip." + L + nm + ".b", seed, (d,), 0.05)
        for nm in ("layer_norm1", "layer_norm2"):
            sd[L + nm + ".weight"] = _randn("clip." + L + nm + ".w", seed, (d,), 0.1, 1.0)
            sd[L + nm + ".bias"] = _randn("clip." + L + nm + ".b", seed, (d,), 0.05)
        sd[L + "mlp.fc1.weight"] = _randn("clip." + L + "fc1.w", seed, (f, d), 1.0 / math.sqrt(d))
        sd[L + "mlp.fc1.bias"] = _randn("clip." + L + "fc1.b", seed, (f,), 0.05)
        sd[L + "mlp.fc2.weight"] = _randn("clip." + L + "fc2.w", seed, (d, f), 1.0 / math.sqrt(f))
        sd[L + "mlp.fc2.bias"] = _randn("clip." + L + "fc2.b", seed, (d,), 0.05)
    return sd


def llama_weights(c: LlamaCfg, seed: int = 0, dtype=torch.float32) -> dict:
    d, f = c.hidden, c.ffn
    sd = {}
    sd["model.embed_tokens.weight"] = _randn("llama.embed", seed, (c.vocab, d), 0.5, dtype=dtype)
    for i in range(c.layers):
        L = f"model.layers.{i}."
        dkv = (c.kv_heads or c.heads) * c.head_dim
        for nm in ("q_proj", "k_proj", "v_proj", "o_proj"):
            rows = dkv if nm in ("k_proj", "v_proj") else d
            sd[L + f"self_attn.{nm}.weight"] = _randn("llama." + L + nm, seed, (d, d), 1.0 / math.sqrt(d), dtype=dtype)[:rows].contiguous()
        sd[L + "mlp.gate_proj.weight"] = _randn("llama." + L + "gate", seed, (f, d), 1.0 / math.sqrt(d), dtype=dtype)
        sd[L + "mlp.up_proj.weight"] = _randn("llama." + L + "up", seed, (f, d), 1.0 / math.sqrt(d), dtype=dtype)
        sd[L + "mlp.down_proj.weight"] = _randn("llama." + L + "down", seed, (d, f), 1.0 / math.sqrt(f), dtype=dtype)
        sd[L + "input_layernorm.weight"] = _randn("llama." + L + "ln1", seed, (d,), 0.1, 1.0, dtype=dtype)
        sd[L + "post_attention_layernorm.weight"] = _randn("llama." + L + "ln2", seed, (d,), 0.1, 1.0, dtype=dtype)
    sd["model.norm.weight"] = _randn("llama.norm", seed, (d,), 0.1, 1.0, dtype=dtype)
    sd["lm_head.weight"] = _randn("llama.head", seed, (c.vocab, d), 1.0 / math.sqrt(d), dtype=dtype)
    return sd


def lora_weights(c: LlamaCfg, l: LoraCfg, seed: int = 0, b_std: float = 0.0) -> dict:
    """peft `lora.Linear` parameters for every target module.

    Reference init (clip_whisper_model.py:973-1000): LoraConfig(init_lora_weights="gaussian")
    => A ~ Normal(0, 1/r), B = 0, then every `lora_` parameter is multiplied by 0.01.
    `b_std` > 0 draws a non-zero B so that tests exercise the adapter path.
    Keys: `layers.{i}.{module}.lora_A` [r, in], `layers.{i}.{module}.lora_B` [out, r].
    """
    d = c.hidden
    dkv = (c.kv_heads or c.heads) * c.head_dim
    sd = {}
    for i in range(c.layers):
        for nm in l.targets:
            rows = dkv if nm in ("k_proj", "v_proj") else d
            sd[f"layers.{i}.{nm}.lora_A"] = _randn(f"lora.{i}.{nm}.A", seed, (l.r, d), (1.0 / l.r) * 0.01)
            if b_std > 0:
                sd[f"layers.{i}.{nm}.lora_B"] = _randn(f"lora.{i}.{nm}.B", seed, (d, l.r), b_std)[:rows].contiguous()
            else:
                sd[f"layers.{i}.{nm}.lora_B"] = torch.zeros(rows, l.r)
    return sd


def connector_weights(in_dim: int, out_dim: int, name: str, seed: int = 0) -> dict:
    """SimpleModalityConnector (modality_connector.py:25-44): xavier-uniform W, zero bias.
    `bias_std` is non-zero in tests so the bias path is exercised."""
    bound = math.sqrt(6.0 / (in_dim + out_dim))
    w = (torch.rand((out_dim, in_dim), generator=_gen(name + ".w", seed)) * 2 - 1) * bound
    b = _randn(name + ".b", seed, (out_dim,), 0.02)
    return {"linear.weight": w, "linear.bias": b}


def all_weights(cfg: ModelCfg, seed: int = 0, lora_b_std: float = 0.05) -> dict:
    return {
        "whisper": whisper_weights(cfg.whisper, seed),
        "clip": clip_weights(cfg.clip, seed),
        "llama": llama_weights(cfg.llama, seed),
        "lora": lora_weights(cfg.llama, cfg.lora, seed, lora_b_std),
        "audio_connector": connector_weights(cfg.whisper.d_model, cfg.llama.hidden, "conn.audio", seed),
        "video_connector": connector_weights(cfg.clip.hidden, cfg.llama.hidden, "conn.video", seed),
    }


def synthetic_batch(cfg: ModelCfg, batch: int, frames: int, seed: int = 1234, label_len: int = 256):
    """Synthetic LRS3-shaped batch: mel ~ N(0,1) [B,80,3000]; CLIP-normalised
    frames from uint8 noise [B,F,3,H,W]; labels [B,label_len] = BOS + n random ids + pad;
    prompt = labels[:, :32] (mirrors trainer/clip_whisper_trainer.py:669-671)."""
    g = torch.Generator().manual_seed(seed)
    audio = torch.randn(batch, cfg.whisper.n_mels, 2 * cfg.whisper.n_ctx, generator=g)
    u8 = torch.randint(0, 256, (batch, frames, 3, cfg.clip.image, cfg.clip.image), generator=g, dtype=torch.uint8)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073]).view(1, 1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711]).view(1, 1, 3, 1, 1)
    video = (u8.float() / 255.0 - mean) / std
    V = cfg.llama.vocab
    labels = torch.full((batch, label_len), cfg.pad_token_id, dtype=torch.long)
    for b in range(batch):
        n = int(torch.randint(8, 41, (1,), generator=g))
        labels[b, 0] = 1
        labels[b, 1:1 + n] = torch.randint(3, V, (n,), generator=g)
    prompt = labels[:, : cfg.max_prompt_len].clone()
    return audio, video, labels, prompt
