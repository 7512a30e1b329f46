"""ClipWhisperModel -- host-side mirror of the reference class of the same name
(src/clip_whisper/models/clip_whisper_model.py:24-1451): same constructor keywords, attributes, methods,
argument meaning and error behaviour; every tensor op runs in libavllm.so (HIP, gfx950).

Differences that are deliberate and documented:
  * precision: `use_fp16=True` selects the MI355X reduced-precision mode (bf16 storage / fp32 accumulate);
    `use_fp16=False` is strict fp32 (fp32 MFMA).  `precision="bf16"|"fp32"` overrides.
  * weights come from local HF checkpoints (safetensors) when the given paths exist, from `_provided_*`
    modules' state_dict()s, from `weights=`, or -- only with `synthetic_weights=True` -- from a seeded synthetic
    initialisation of the named architecture (a path that is none of these raises FileNotFoundError).
  * freeze_encoders=False / use_4bit=True are refused (out of scope).
"""
from __future__ import annotations

import logging
import os
import types

import torch

from . import lib as L
from . import ops
from .arch import ModelCfg, resolve_arch
from .connector import ModalityConnector, create_modality_connector
from .engine import ClipEngine, LlamaEngine, WhisperEngine
from .tokenizer import load_tokenizer


class _LoraLoss(torch.autograd.Function):
    """Makes `out["loss"].backward()` (trainer/clip_whisper_trainer.py:454) drive avllm_llama_lora_bwd."""

    @staticmethod
    def forward(ctx, lora_param, model, loss_value):
        ctx.model = model
        ctx.gen = model.llm_engine.gen
        return loss_value.clone()

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        eng = m.llm_engine
        if eng.gen != ctx.gen:
            raise RuntimeError("loss.backward(): another training forward ran since this loss was computed; its activations were overwritten")
        eng.lora_g.zero_()
        eng.bwd(grad_scale=float(grad_out))
        return eng.lora_g.clone(), None, None


class ClipWhisperModel:
    def __init__(self, llm_path="meta-llama/Llama-2-7b-hf", whisper_model="openai/whisper-small",
                 clip_model="openai/clip-vit-base-patch16", device="cuda", use_fp16=False, use_4bit=False, use_lora=True,
                 lora_r=16, lora_alpha=32, lora_dropout=0.05, freeze_encoders=True, freeze_llm=False, modality="both",
                 max_seq_len=256, fusion_scale=0.5, connector_type="simple", _provided_tokenizer=None, _provided_llm=None,
                 _provided_whisper=None, _provided_clip=None, *, precision=None, config: ModelCfg | None = None,
                 weights: dict | None = None, seed: int = 0, synthetic_weights: bool = False):
        if use_4bit:
            raise NotImplementedError("use_4bit (bitsandbytes nf4) is out of scope of the MI355X hot path (SURVEY.md §8)")
        if not freeze_encoders:
            raise NotImplementedError("freeze_encoders=False is not supported: the hot path trains LoRA only (SURVEY.md fact 4)")
        if modality not in ("audio", "video", "both"):
            raise ValueError(f"modality must be audio|video|both, got {modality}")
        L.load()                                   # fail loudly when the HIP library is missing
        self.device = device
        self.use_fp16, self.use_4bit, self.use_lora = use_fp16, use_4bit, use_lora
        if use_lora and "llama" not in str(llm_path).lower():
            # clip_whisper_model.py:966-970 picks target modules ["query","key","value","dense"] for such paths; no Llama/Mistral
            # module carries those names, so peft raises and the reference "continues without LoRA" (:1002-1005).  This build
            # attaches the q/k/v/o adapters to every Llama-architecture model instead.
            logging.warning("LLM path %r does not contain 'llama': the reference would train without LoRA here; "
                            "this build applies LoRA to q_proj/k_proj/v_proj/o_proj", llm_path)
        self.lora_r, self.lora_alpha, self.lora_dropout = lora_r, lora_alpha, lora_dropout
        self.freeze_encoders, self.freeze_llm = freeze_encoders, freeze_llm
        self.modality, self.max_seq_len, self.fusion_scale = modality, max_seq_len, fusion_scale
        self.connector_type = connector_type
        precision = precision or ("bf16" if use_fp16 else "fp32")
        if precision not in ("fp32", "bf16", "fp8"):
            raise ValueError(f"precision must be fp32|bf16|fp8, got {precision}")
        # "fp8" (BASELINE config 5): bf16 storage, with every frozen-weight projection of the encoders and of the LLM's training forward
        # on the block-scaled fp8 matrix pipe; attention, norms, LoRA terms, loss and the whole backward pass as in "bf16"
        self.fp8 = precision == "fp8"
        self.precision = precision
        self.dtype = torch.float32 if precision == "fp32" else torch.bfloat16
        self.training = True
        self._drop_step = 0
        self._seed = int(seed or 0)

        cfg, W = resolve_arch(llm_path, whisper_model, clip_model, config, weights, seed, lora_r, lora_alpha, use_lora,
                              _provided_llm, _provided_whisper, _provided_clip, device, self.dtype, synthetic_weights)
        cfg.max_seq_len, cfg.fusion_scale = max_seq_len, fusion_scale
        self.cfg = cfg
        # explicit `weights=` / `config=` builds (tests, bench) are synthetic by construction: they carry no tokenizer either
        self.tokenizer = _provided_tokenizer or load_tokenizer(llm_path, cfg.llama.vocab, synthetic=synthetic_weights or weights is not None)
        if getattr(self.tokenizer, "pad_token_id", None) is None:
            self.tokenizer.pad_token_id = getattr(self.tokenizer, "eos_token_id", 2)      # pad = eos (:955-959)
        cfg.pad_token_id = self.tokenizer.pad_token_id
        self.whisper_processor = None
        self.clip_processor = None
        self.audio_dim, self.video_dim, self.llm_dim = cfg.whisper.d_model, cfg.clip.hidden, cfg.llama.hidden

        self.whisper_engine = WhisperEngine(W["whisper"], cfg.whisper, self.dtype, device, fp8=self.fp8) if modality in ("audio", "both") else None
        self.clip_engine = ClipEngine(W["clip"], cfg.clip, self.dtype, device, fp8=self.fp8) if modality in ("video", "both") else None
        self.llm_engine = LlamaEngine(W["llama"], cfg.llama, cfg.lora if use_lora else None, W.get("lora"), self.dtype, device,
                                      training=True, fp8=self.fp8)
        self._setup_projections(W)
        self.lora_param = None
        if use_lora:
            self.lora_param = torch.nn.Parameter(self.llm_engine.lora_p, requires_grad=not freeze_llm)
        self.eos_token_id = getattr(self.tokenizer, "eos_token_id", 2)

    # ------------------------------------------------------------------ plumbing mirrored from nn.Module
    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def to(self, *a, **k):
        return self

    def parameters(self):
        ps = [p for c in (self.audio_connector, self.video_connector) for p in c.parameters()]
        if self.lora_param is not None:
            ps.append(self.lora_param)
        return ps

    def named_parameters(self):
        out = [(f"audio_connector.{n}", p) for n, p in self.audio_connector.named_parameters()]
        out += [(f"video_connector.{n}", p) for n, p in self.video_connector.named_parameters()]
        if self.lora_param is not None:
            out.append(("llm.lora_flat", self.lora_param))
        return out

    def state_dict(self):
        """Trainable/connector tensors under the reference's key names (decode.py:237-238 matches by substring;
        LoRA uses the peft naming `llm.base_model.model.model.layers.N.self_attn.X.lora_A.default.weight`)."""
        sd = {f"audio_connector.{k}": v.detach().clone() for k, v in self.audio_connector.state_dict().items()}
        sd.update({f"video_connector.{k}": v.detach().clone() for k, v in self.video_connector.state_dict().items()})
        if self.use_lora:
            for k, v in self.llm_engine.lora_views().items():
                _, i, mod, ab = k.split(".")
                sd[f"llm.base_model.model.model.layers.{i}.self_attn.{mod}.{ab}.default.weight"] = v.detach().clone()
        return sd

    def load_state_dict(self, sd, strict=False):
        a = {k.split("audio_connector.")[1]: v for k, v in sd.items() if "audio_connector." in k}
        v_ = {k.split("video_connector.")[1]: v for k, v in sd.items() if "video_connector." in k}
        if a:
            self.audio_connector.load_state_dict(a)
        if v_:
            self.video_connector.load_state_dict(v_)
        lora = {}
        for k, t in sd.items():
            if ".lora_A." in k or ".lora_B." in k:
                parts = k.split(".")
                i = parts[parts.index("layers") + 1]
                mod = parts[parts.index("self_attn") + 1]
                ab = "lora_A" if ".lora_A." in k else "lora_B"
                lora[f"layers.{i}.{mod}.{ab}"] = t
        if lora and self.use_lora:
            self.llm_engine.load_lora(lora)
            self.llm_engine.pack_lora()
        return types.SimpleNamespace(missing_keys=[], unexpected_keys=[])

    def save_pretrained(self, output_dir):
        """Directory layout of clip_whisper_model.py:738-798 for the tensors this build owns."""
        os.makedirs(output_dir, exist_ok=True)
        torch.save(self.audio_connector.state_dict(), os.path.join(output_dir, "audio_connector.pt"))
        torch.save(self.video_connector.state_dict(), os.path.join(output_dir, "video_connector.pt"))
        import json
        cfg = dict(modality=self.modality, max_seq_len=self.max_seq_len, fusion_scale=self.fusion_scale, use_lora=self.use_lora,
                   lora_r=self.lora_r, lora_alpha=self.lora_alpha, connector_type=self.connector_type,
                   audio_dim=self.audio_dim, video_dim=self.video_dim, llm_dim=self.llm_dim)
        json.dump(cfg, open(os.path.join(output_dir, "config.json"), "w"), indent=2)
        torch.save(cfg, os.path.join(output_dir, "config.pt"))
        if self.use_lora:
            os.makedirs(os.path.join(output_dir, "llm"), exist_ok=True)
            torch.save({k: v for k, v in self.state_dict().items() if "lora_" in k}, os.path.join(output_dir, "llm", "adapter_model.pt"))

    @classmethod
    def from_pretrained(cls, model_dir, tokenizer=None, **kwargs):
        """clip_whisper_model.py:821-864.  The reference's own pair is not closed: its save_pretrained writes `config.pt` / `config.json` while its
        from_pretrained demands `model_config.json` (and `whisper/`, `clip/` directories that are only written with freeze_encoders=False), so it
        cannot reload what it saved.  This one reloads what save_pretrained above wrote -- connectors, adapters, the saved settings -- on top of
        base checkpoints named by the usual constructor keywords (`llm_path=`, `whisper_model=`, `clip_model=`, or `config=` + `synthetic_weights=`)."""
        import json
        if not os.path.exists(model_dir):
            raise ValueError(f"Model directory {model_dir} does not exist")
        cfg_path = os.path.join(model_dir, "config.json")
        if not os.path.exists(cfg_path):
            raise ValueError(f"Model config {cfg_path} does not exist")
        cfg = json.load(open(cfg_path))
        for k in ("modality", "max_seq_len", "fusion_scale", "use_lora", "lora_r", "lora_alpha", "connector_type"):
            if k in cfg:
                kwargs.setdefault(k, cfg[k])
        model = cls(_provided_tokenizer=tokenizer, **kwargs)
        model.audio_connector.load_state_dict(torch.load(os.path.join(model_dir, "audio_connector.pt"), map_location=model.device, weights_only=True))
        model.video_connector.load_state_dict(torch.load(os.path.join(model_dir, "video_connector.pt"), map_location=model.device, weights_only=True))
        adapters = os.path.join(model_dir, "llm", "adapter_model.pt")
        if model.use_lora and os.path.exists(adapters):
            model.load_state_dict(torch.load(adapters, map_location="cpu", weights_only=True))
        return model

    def _setup_projections(self, W=None):
        """clip_whisper_model.py:1159-1190: both connectors always exist."""
        # xavier init as in the reference, but drawn from a private generator state derived from `seed` (the global RNG is left alone):
        # two builds with the same seed are the same model, which the reference's unseeded nn.Linear init does not give
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed(0x5EED + int(self._seed))
            self.audio_connector = create_modality_connector(self.connector_type, self.audio_dim, self.llm_dim, self.device, self.dtype)
            self.video_connector = create_modality_connector(self.connector_type, self.video_dim, self.llm_dim, self.device, self.dtype)
        if W is not None:
            if "audio_connector" in W:
                self.audio_connector.load_state_dict(W["audio_connector"])
            if "video_connector" in W:
                self.video_connector.load_state_dict(W["video_connector"])

    def _get_llm_dim(self):
        return self.cfg.llama.hidden

    # ------------------------------------------------------------------ encoders (+ connectors)
    def encode_audio(self, audio, attention_mask=None, rows=None):
        """clip_whisper_model.py:1067-1106.  `rows`: project only the first `rows` frames (identical result for the
        rows that survive encode()'s truncation, 3x fewer connector FLOPs at L=512)."""
        if audio is None:
            raise ValueError("Audio input cannot be None")
        nm = self.cfg.whisper.n_mels                               # the reference hard-codes 80 (:1074); lifted for whisper-large-v3 (128)
        if audio.dim() != 2 and (audio.dim() != 3 or audio.shape[1] != nm):
            raise ValueError(f"Audio input should have shape [batch_size, sequence_length] or [batch_size, {nm}, time_steps], but got {audio.shape}")
        if audio.dim() == 2:
            raise ValueError(f"raw-waveform audio is not supported: pass WhisperFeatureExtractor mel features [B,{nm},3000]")
        if self.whisper_engine is None:
            raise ValueError("audio encoder not loaded (modality=video)")
        h = self.whisper_engine.forward(audio)                       # [B,1500,d]
        return self.audio_connector(h if rows is None else h[:, :rows])

    def encode_video(self, video, attention_mask=None, rows=None):
        """clip_whisper_model.py:1108-1146: CLS of last_hidden_state per frame (no post_layernorm) -> connector."""
        if video.dim() != 5 or video.shape[2] != 3:
            raise ValueError(f"Video input should have shape [batch_size, frames, 3, height, width], but got {video.shape}")
        if self.clip_engine is None:
            raise ValueError("video encoder not loaded (modality=audio)")
        B, Fr = video.shape[:2]
        cls = self.clip_engine.forward(video.reshape(B * Fr, 3, video.shape[3], video.shape[4])).view(B, Fr, -1)
        return self.video_connector(cls if rows is None else cls[:, :rows])

    def _embed_prompt(self, prompt):
        if prompt is None:
            return None
        if isinstance(prompt, str):
            prompt = torch.tensor([self.tokenizer.encode(prompt)[:32]], dtype=torch.long)
        ids = prompt.to(self.device)[:, :32].contiguous()            # max_prompt_len = 32 (:469)
        return ops.embedding(self.llm_engine.embed, ids)

    def _features(self, audio, video):
        """(a_feat, v_feat, L) with the encode() modality rules (clip_whisper_model.py:407-445)."""
        a = v = None
        use_a = self.modality in ("audio", "both") and audio is not None
        use_v = self.modality in ("video", "both") and video is not None
        if use_a and use_v:
            if self.connector_type not in ("conv", "attention", "adaptive"):
                # per-token connectors (simple, deep = every unknown name): projecting only the rows that survive the truncation below is the
                # same function at a third of the connector FLOPs
                Ta, Tv = self.cfg.whisper.n_ctx, video.shape[1]
                Lc = min(self.max_seq_len, max(Ta, Tv))
                a = self.encode_audio(audio, rows=min(Ta, Lc))
                v = self.encode_video(video, rows=min(Tv, Lc))
                return a, v, Lc
            # sequence-mixing connectors (conv / attention / adaptive) see the whole sequence, and `adaptive` changes its length (T > 512 -> T / 4):
            # lengths are taken AFTER the connectors, as encode() does (:424-430)
            a, v = self.encode_audio(audio), self.encode_video(video)
            return a, v, min(self.max_seq_len, max(a.shape[1], v.shape[1]))
        if use_a:
            a = self.encode_audio(audio)
            return a, None, a.shape[1]
        if use_v:
            v = self.encode_video(video)
            return None, v, v.shape[1]
        raise ValueError("No valid inputs provided - both audio and video are None")

    def _llm_inputs(self, audio, video, prompt, S_out=None):
        a, v, Lc = self._features(audio, video)
        pe = self._embed_prompt(prompt)
        P = pe.shape[1] if pe is not None else 0
        B = (a if a is not None else v).shape[0]
        S = P + Lc if S_out is None else S_out
        return ops.fuse_pool(a, v, pe, Lc, S, self.fusion_scale, self.llm_dim, B)

    def encode(self, audio=None, video=None, prompt=None):
        """-> (inputs_embeds [B,P+L,D], attention_mask ones int64 [B,P+L])  (clip_whisper_model.py:407-462)."""
        x = self._llm_inputs(audio, video, prompt)
        return x, torch.ones(x.shape[0], x.shape[1], dtype=torch.long, device=x.device)

    # ------------------------------------------------------------------ forward / loss
    def _dropout_args(self):
        """lora_dropout is active in train() mode only; every training forward draws a fresh mask seed."""
        if not (self.training and self.use_lora and self.lora_dropout):
            return {"dropout": 0.0, "seed": 0}
        self._drop_step += 1
        rank = 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            rank = torch.distributed.get_rank()           # data-parallel ranks draw independent masks, as independent processes would
        return {"dropout": float(self.lora_dropout), "seed": (self._drop_step * 0x9E3779B1 + rank * 0x85EBCA6B + 12345) & 0xFFFFFFFF}

    def _prep_labels(self, labels):
        if isinstance(labels, list):
            if all(isinstance(t, torch.Tensor) for t in labels):
                labels = torch.stack(labels)
            else:
                labels = torch.tensor(labels)
        labels = labels.to(self.device)
        # :569-570 `labels[labels == pad] = -100` as one fresh tensor (no boolean-index assignment: that form syncs with the host and
        # cannot be captured in a graph)
        return torch.where(labels == self.tokenizer.pad_token_id, torch.full_like(labels, -100), labels)

    def forward(self, audio=None, video=None, prompt=None, labels=None, return_loss=True):
        """clip_whisper_model.py:489-619 -> {"loss","logits"} | {"logits"}."""
        if labels is not None and return_loss:
            try:
                labels = self._prep_labels(labels)
            except Exception as e:                                     # :541-558 dummy-loss behaviour
                logging.error(f"Failed to convert labels to tensor: {e}")
                return {"loss": torch.tensor(1.0, device=self.device, requires_grad=True), "logits": None}
        else:
            labels = None
        if self.training and labels is not None:
            x = self._llm_inputs(audio, video, prompt, S_out=labels.shape[1])       # adaptive pool / interpolate (:577-585)
            logits = self.llm_engine.fwd_loss(x, labels, want_logits=True, **self._dropout_args())
            acc = self.llm_engine.acc
            loss = acc[0] / acc[1]
            if self.lora_param is not None and self.lora_param.requires_grad:
                loss = _LoraLoss.apply(self.lora_param, self, loss)
            return {"loss": loss, "logits": logits}
        x = self._llm_inputs(audio, video, prompt)
        B, S, _ = x.shape
        kc, vc = self.llm_engine.alloc_cache(B, S)
        _, logits = self.llm_engine.prefill(x, kc, vc, all_logits=True)
        if labels is None:
            return {"logits": logits}
        if labels.shape[1] > S:                                           # :586-598
            labels = labels[:, :S]
        elif labels.shape[1] < S:
            pad = torch.full((B, S - labels.shape[1]), -100, dtype=labels.dtype, device=labels.device)
            labels = torch.cat([labels, pad], dim=1)
        _, acc = ops.ce_fwd(logits, labels.contiguous())
        return {"loss": acc[0] / acc[1], "logits": logits}

    __call__ = forward

    # ------------------------------------------------------------------ generation
    @torch.no_grad()
    def generate(self, audio=None, video=None, prompt=None, pixel_values=None, max_new_tokens=100, do_sample=False,
                 temperature=1.0, top_p=0.9, max_length=None):
        """clip_whisper_model.py:1240-1348 -> GenerationMixin greedy search; returns NEW tokens only [B, <=max_new_tokens]."""
        if video is None and pixel_values is not None:
            video = pixel_values
        if max_new_tokens is None:
            max_new_tokens = max_length if max_length is not None else 100
        if do_sample:
            raise NotImplementedError("do_sample=True: the reference's decode path is greedy (decode.py:544-549)")
        original = self.modality
        if audio is not None and video is not None:
            self.modality = "both"
        elif audio is not None:
            self.modality = "audio"
        elif video is not None:
            self.modality = "video"
        try:
            x = self._llm_inputs(audio, video, prompt)
        finally:
            self.modality = original
        eng = self.llm_engine
        B, S, _ = x.shape
        kc, vc = eng.alloc_cache(B, S + max_new_tokens)
        logits, _ = eng.prefill(x, kc, vc)
        eos, pad = self.eos_token_id, self.tokenizer.pad_token_id
        unfinished = torch.ones(B, dtype=torch.bool, device=x.device)
        out = []
        for step in range(max_new_tokens):
            nxt = ops.argmax_rows(logits)
            if eos is not None:
                nxt = torch.where(unfinished, nxt, torch.full_like(nxt, pad))
                unfinished = unfinished & (nxt != eos)
            out.append(nxt)
            if step + 1 == max_new_tokens or (eos is not None and not bool(unfinished.any())):
                break
            logits = eng.decode_step(nxt, S + step, kc, vc)
        return torch.stack(out, dim=1)
