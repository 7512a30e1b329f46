"""Host-side owners of device weights/workspaces for the three model-level C-ABI engines.

Weights arrive as HuggingFace-named state dicts (the modules the reference instantiates:
clip_whisper_model.py:864-1019), are re-laid-out once for the HIP kernels (fused qkv / gate-up, im2col-ordered
conv weights, transposed images for the dX GEMMs) and stay resident in HBM.  All compute is in libavllm.so.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import lib as L
from . import ops


class Workspace:
    """One growable byte buffer handed to the C ABI as scratch (the library never allocates).

    `Workspace.generation` counts reallocations of ANY workspace: a captured hipGraph holds raw pointers into these buffers, so whoever
    replays one compares the generation it captured under with the current one and drops its graphs when they differ
    (ClipWhisperTrainer.train_step).  Growth is monotonic, so a mix of input signatures settles after each has been seen once."""

    generation = 0

    def __init__(self, device):
        self.device = device
        self.buf = None

    def get(self, nbytes: int):
        if self.buf is None or self.buf.numel() < nbytes:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("workspace would grow inside a hipGraph capture (the eager warm-up step must size it first)")
            self.buf = None
            self.buf = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            Workspace.generation += 1
        return self.buf


def _dev(t, dtype, device):
    return t.detach().to(device=device, dtype=dtype).contiguous()


def _pad_cols(w2d, kpad):
    if w2d.shape[1] == kpad:
        return w2d
    out = torch.zeros(w2d.shape[0], kpad, dtype=w2d.dtype, device=w2d.device)
    out[:, : w2d.shape[1]] = w2d
    return out


def _fp8_images(w, keep):
    """Block-scaled fp8 image of a weight matrix [N,K] (weight-side layout) -> (codes ptr, scales ptr); tensors parked in `keep`."""
    q, s = ops.mx_quantize(w, 1)
    keep += [q, s]
    return q.data_ptr(), s.data_ptr()


def _enc_layers(sd, prefix_fmt, n, names, dtype, device, keep, fp8=False):
    """Build the EncLayer array for a pre-LN encoder; `names` maps our fields to HF suffixes."""
    arr = (L.EncLayer * n)()
    for i in range(n):
        p = prefix_fmt.format(i)
        g = lambda k: sd[p + k]
        q, k_, v = g(names["q"] + ".weight"), g(names["k"] + ".weight"), g(names["v"] + ".weight")
        d = q.shape[0]
        bq = g(names["q"] + ".bias")
        bk = sd.get(p + names["k"] + ".bias")
        if bk is None:
            bk = torch.zeros(d, dtype=bq.dtype, device=bq.device)     # Whisper k_proj has no bias (HF whisper :276)
        bv = g(names["v"] + ".bias")
        t = {
            "ln1_w": g(names["ln1"] + ".weight"), "ln1_b": g(names["ln1"] + ".bias"),
            "wqkv": torch.cat([q, k_, v], 0), "bqkv": torch.cat([bq, bk, bv], 0),
            "wo": g(names["o"] + ".weight"), "bo": g(names["o"] + ".bias"),
            "ln2_w": g(names["ln2"] + ".weight"), "ln2_b": g(names["ln2"] + ".bias"),
            "w1": g(names["fc1"] + ".weight"), "b1": g(names["fc1"] + ".bias"),
            "w2": g(names["fc2"] + ".weight"), "b2": g(names["fc2"] + ".bias"),
        }
        for k, val in t.items():
            dv = _dev(val, dtype, device)
            keep.append(dv)
            setattr(arr[i], k, dv.data_ptr())
            if fp8 and k in ("wqkv", "wo", "w1", "w2"):
                q8, s8 = _fp8_images(dv, keep)
                f = {"wqkv": ("wqkv8", "sqkv8"), "wo": ("wo8", "so8"), "w1": ("w18", "s18"), "w2": ("w28", "s28")}[k]
                setattr(arr[i], f[0], q8); setattr(arr[i], f[1], s8)
    return arr


class WhisperEngine:
    """avllm_whisper_encoder_fwd: mel f32 [B,80,2*n_ctx] -> [B,n_ctx,d]."""

    def __init__(self, sd, cfg, dtype=torch.bfloat16, device="cuda", fp8=False):
        self.cfg, self.dtype, self.device = cfg, dtype, device
        self.keep = []
        d = cfg.d_model
        k1pad = (3 * cfg.n_mels + 63) // 64 * 64
        w = L.Whisper()
        w.dtype, w.d, w.heads, w.layers, w.ffn, w.n_mels, w.n_ctx, w.k1pad = (
            L.F32 if dtype == torch.float32 else L.BF16, d, cfg.heads, cfg.layers, cfg.ffn, cfg.n_mels, cfg.n_ctx, k1pad)
        c1 = _pad_cols(sd["encoder.conv1.weight"].reshape(d, 3 * cfg.n_mels), k1pad)         # column c*3+kw
        c2 = sd["encoder.conv2.weight"].permute(0, 2, 1).reshape(d, 3 * d)                   # column kw*d+c
        for name, val in (("conv1_w", c1), ("conv1_b", sd["encoder.conv1.bias"]), ("conv2_w", c2),
                          ("conv2_b", sd["encoder.conv2.bias"]), ("pos", sd["encoder.embed_positions.weight"]),
                          ("lnf_w", sd["encoder.layer_norm.weight"]), ("lnf_b", sd["encoder.layer_norm.bias"])):
            dv = _dev(val, dtype, device)
            self.keep.append(dv)
            setattr(w, name, dv.data_ptr())
        names = dict(q="self_attn.q_proj", k="self_attn.k_proj", v="self_attn.v_proj", o="self_attn.out_proj",
                     ln1="self_attn_layer_norm", ln2="final_layer_norm", fc1="fc1", fc2="fc2")
        self.layers = _enc_layers(sd, "encoder.layers.{}.", cfg.layers, names, dtype, device, self.keep, fp8)
        w.layer = C.cast(self.layers, C.POINTER(L.EncLayer))
        w.fp8 = int(bool(fp8))
        self.desc = w
        self.ws = Workspace(device)

    def forward(self, mel):
        if mel.dim() != 3 or mel.shape[1] != self.cfg.n_mels:
            raise ValueError(f"Audio input should have shape [batch_size, {self.cfg.n_mels}, time_steps], but got {tuple(mel.shape)}")
        if mel.shape[-1] != 2 * self.cfg.n_ctx:
            raise ValueError(f"Whisper expects the mel input features to be of length {2 * self.cfg.n_ctx}, but found {mel.shape[-1]}")
        lib = L.load()
        mel = mel.to(device=self.device, dtype=torch.float32).contiguous()
        B = mel.shape[0]
        out = torch.empty(B, self.cfg.n_ctx, self.cfg.d_model, device=self.device, dtype=self.dtype)
        n = lib.avllm_whisper_workspace_bytes(C.byref(self.desc), B)
        ws = self.ws.get(n)
        L.check(lib.avllm_whisper_encoder_fwd(C.byref(self.desc), L.ptr(mel), B, L.ptr(out), L.ptr(ws), ws.numel(), L.stream_ptr()))
        return out


class ClipEngine:
    """avllm_clip_vision_cls_fwd: frames f32 [N,3,S,S] -> CLS of last_hidden_state [N,d] (no post_layernorm)."""

    def __init__(self, sd, cfg, dtype=torch.bfloat16, device="cuda", chunk_frames=0, fp8=False):
        sd = {k[len("vision_model."):] if k.startswith("vision_model.") else k: v for k, v in sd.items()}
        self.cfg, self.dtype, self.device = cfg, dtype, device
        self.chunk = chunk_frames
        self.keep = []
        d = cfg.hidden
        kpad = (3 * cfg.patch * cfg.patch + 63) // 64 * 64
        c = L.Clip()
        c.dtype, c.d, c.heads, c.layers, c.ffn, c.image, c.patch, c.tokens = (
            L.F32 if dtype == torch.float32 else L.BF16, d, cfg.heads, cfg.layers, cfg.mlp, cfg.image, cfg.patch, cfg.tokens)
        c.eps = cfg.eps
        pw = _pad_cols(sd["embeddings.patch_embedding.weight"].reshape(d, -1), kpad)
        for name, val in (("patch_w", pw), ("class_emb", sd["embeddings.class_embedding"]),
                          ("pos", sd["embeddings.position_embedding.weight"]), ("pre_ln_w", sd["pre_layrnorm.weight"]),
                          ("pre_ln_b", sd["pre_layrnorm.bias"])):
            dv = _dev(val, dtype, device)
            self.keep.append(dv)
            setattr(c, name, dv.data_ptr())
        names = dict(q="self_attn.q_proj", k="self_attn.k_proj", v="self_attn.v_proj", o="self_attn.out_proj",
                     ln1="layer_norm1", ln2="layer_norm2", fc1="mlp.fc1", fc2="mlp.fc2")
        self.layers = _enc_layers(sd, "encoder.layers.{}.", cfg.layers, names, dtype, device, self.keep, fp8)
        c.layer = C.cast(self.layers, C.POINTER(L.EncLayer))
        c.fp8 = int(bool(fp8))
        self.desc = c
        self.ws = Workspace(device)

    def forward(self, frames):
        if frames.dim() != 4 or frames.shape[1] != 3:
            raise ValueError(f"frames should have shape [N, 3, H, W], but got {tuple(frames.shape)}")
        if frames.shape[-1] != self.cfg.image or frames.shape[-2] != self.cfg.image:
            raise ValueError(f"Input image size ({frames.shape[-2]}*{frames.shape[-1]}) doesn't match model ({self.cfg.image}*{self.cfg.image}).")
        lib = L.load()
        # fp32 pixel_values as the reference's CLIPProcessor hands them over, or bf16 (device-side preprocessing, avllm.preprocess.ClipFrames)
        frames = frames.to(device=self.device, dtype=torch.bfloat16 if frames.dtype == torch.bfloat16 else torch.float32).contiguous()
        self.desc.frames_bf16 = int(frames.dtype == torch.bfloat16)
        N = frames.shape[0]
        out = torch.empty(N, self.cfg.hidden, device=self.device, dtype=self.dtype)
        step = self.chunk if self.chunk and self.chunk < N else N
        ws = self.ws.get(lib.avllm_clip_workspace_bytes(C.byref(self.desc), step))
        for s in range(0, N, step):
            n = min(step, N - s)
            L.check(lib.avllm_clip_vision_cls_fwd(C.byref(self.desc), L.ptr(frames[s:]), n, L.ptr(out[s:]), L.ptr(ws), ws.numel(), L.stream_ptr()))
        return out


LORA_TARGETS = ("q_proj", "k_proj", "v_proj", "o_proj")


class LlamaEngine:
    """Llama + LoRA: avllm_llama_lora_fwd_loss / _bwd (training) and prefill / decode_step (generation).

    Trainable state: one flat fp32 buffer `lora_p` holding, per layer, [A_q,B_q,A_k,B_k,A_v,B_v,A_o,B_o]
    (A [r,d], B [dout,r]; dout = d, or kv_heads*head_dim for k/v under grouped-query attention) so a layer's gradient bucket is one contiguous slice of `lora_g` for the DDP all-reduce.
    """

    def __init__(self, sd, cfg, lora_cfg=None, lora_sd=None, dtype=torch.bfloat16, device="cuda", training=True, fp8=False):
        self.cfg, self.lcfg, self.dtype, self.device, self.training = cfg, lora_cfg, dtype, device, training
        self.fp8 = bool(fp8)
        self.keep = []
        d, f = cfg.hidden, cfg.ffn
        self.use_lora = lora_cfg is not None
        r = lora_cfg.r if self.use_lora else 0
        self.r = r
        m = L.Llama()
        m.dtype, m.d, m.heads, m.layers, m.ffn, m.vocab, m.lora_r = (
            L.F32 if dtype == torch.float32 else L.BF16, d, cfg.heads, cfg.layers, f, cfg.vocab, r)
        kvh = getattr(cfg, "kv_heads", 0) or cfg.heads
        if cfg.heads % kvh:
            raise ValueError(f"heads={cfg.heads} is not a multiple of kv_heads={kvh}")
        m.kv_heads = kvh
        self.dkv = kvh * (d // cfg.heads)
        self.douts = (d, self.dkv, self.dkv, d)                  # output width of q, k, v, o (grouped-query: k/v are narrower)
        m.eps, m.theta, m.lora_scale = cfg.eps, cfg.theta, (lora_cfg.scale if self.use_lora else 0.0)
        rs = tuple(getattr(cfg, "rope_scaling", ()) or ())
        if rs:                                                   # Llama-3.1 / 3.2 "llama3" RoPE frequency scaling (include/avllm.h)
            m.rope_factor, m.rope_low_freq_factor, m.rope_high_freq_factor, m.rope_orig_ctx = float(rs[0]), float(rs[1]), float(rs[2]), int(rs[3])

        def put(val):
            dv = _dev(val, dtype, device)
            self.keep.append(dv)
            return dv

        self.embed = put(sd["model.embed_tokens.weight"])
        m.embed = self.embed.data_ptr()
        m.norm_w = put(sd["model.norm.weight"]).data_ptr()
        head = put(sd["lm_head.weight"])
        m.lm_head = head.data_ptr()
        if training:
            m.lm_head_t = put(head.t()).data_ptr()
        m.fp8 = int(self.fp8)
        if self.fp8:
            m.lm_head8, m.slm_head8 = _fp8_images(head, self.keep)
        self.layers = (L.LlamaLayer * cfg.layers)()
        # ---- LoRA masters / grads / padded operand images
        self.per_layer = sum(r * (d + do) for do in self.douts)
        n = cfg.layers * self.per_layer
        if self.use_lora:
            self.lora_p = torch.zeros(n, dtype=torch.float32, device=device)
            self.lora_g = torch.zeros(n, dtype=torch.float32, device=device)
            P = L.LORA_PAD
            self.img_A = torch.zeros(cfg.layers, 4, P, d, dtype=dtype, device=device)
            self.img_ATqkv = torch.zeros(cfg.layers, d, 3 * P, dtype=dtype, device=device)
            self.img_ATo = torch.zeros(cfg.layers, d, P, dtype=dtype, device=device)
            self.img_B = [[torch.zeros(do, P, dtype=dtype, device=device) for do in self.douts] for _ in range(cfg.layers)]
            self.img_BT = [[torch.zeros(P, do, dtype=dtype, device=device) for do in self.douts] for _ in range(cfg.layers)]
            if lora_sd is not None:
                self.load_lora(lora_sd)
        for i in range(cfg.layers):
            p = f"model.layers.{i}."
            ly = self.layers[i]
            wqkv = put(torch.cat([sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.k_proj.weight"],
                                  sd[p + "self_attn.v_proj.weight"]], 0))
            wo = put(sd[p + "self_attn.o_proj.weight"])
            wgu = put(torch.cat([sd[p + "mlp.gate_proj.weight"], sd[p + "mlp.up_proj.weight"]], 0))
            wdown = put(sd[p + "mlp.down_proj.weight"])
            ly.ln1_w = put(sd[p + "input_layernorm.weight"]).data_ptr()
            ly.ln2_w = put(sd[p + "post_attention_layernorm.weight"]).data_ptr()
            ly.wqkv, ly.wo, ly.wgu, ly.wdown = wqkv.data_ptr(), wo.data_ptr(), wgu.data_ptr(), wdown.data_ptr()
            if training:
                ly.wqkv_t, ly.wo_t = put(wqkv.t()).data_ptr(), put(wo.t()).data_ptr()
                ly.wgu_t, ly.wdown_t = put(wgu.t()).data_ptr(), put(wdown.t()).data_ptr()
            if self.fp8:
                ly.wqkv8, ly.sqkv8 = _fp8_images(wqkv, self.keep)
                ly.wo8, ly.so8 = _fp8_images(wo, self.keep)
                ly.wgu8, ly.sgu8 = _fp8_images(wgu, self.keep)
                ly.wdown8, ly.sdown8 = _fp8_images(wdown, self.keep)
            if self.use_lora:
                es = self.img_A.element_size()
                for j in range(4):
                    lm = ly.lora[j]
                    lm.A_pad = self.img_A[i, j].data_ptr()
                    lm.B_pad = self.img_B[i][j].data_ptr()
                    lm.BT_pad = self.img_BT[i][j].data_ptr()
                    if j < 3:
                        lm.AT_pad, lm.ld_at = self.img_ATqkv[i].data_ptr() + j * L.LORA_PAD * es, 3 * L.LORA_PAD
                    else:
                        lm.AT_pad, lm.ld_at = self.img_ATo[i].data_ptr(), L.LORA_PAD
                    a, b = self._slices(i, j)
                    lm.gA = self.lora_g[a[0]:a[1]].data_ptr()
                    lm.gB = self.lora_g[b[0]:b[1]].data_ptr()
        m.layer = C.cast(self.layers, C.POINTER(L.LlamaLayer))
        self.desc = m
        self.ws = Workspace(device)            # training: holds the saved activations between fwd_loss() and bwd()
        self.ws_infer = Workspace(device)      # prefill / decode_step scratch: an eval forward or generate() between a training
        self._last = None                      #   forward and its backward must not touch the saved activations
        self.acc = torch.zeros(2, dtype=torch.float32, device=device)     # [loss_sum, count]
        if self.use_lora:
            self.pack_lora()

    # flat-buffer offsets of module j (0..3 = q,k,v,o) in layer i: (A range, B range)
    def _slices(self, i, j):
        r, d = self.r, self.cfg.hidden
        base = i * self.per_layer + sum(r * (d + do) for do in self.douts[:j])
        return (base, base + r * d), (base + r * d, base + r * d + self.douts[j] * r)

    def lora_views(self, buf=None):
        """dict key -> fp32 view into the flat buffer, keys `layers.{i}.{module}.lora_A|B` (oracle naming)."""
        buf = self.lora_p if buf is None else buf
        r, d = self.r, self.cfg.hidden
        out = {}
        for i in range(self.cfg.layers):
            for j, nm in enumerate(LORA_TARGETS):
                a, b = self._slices(i, j)
                out[f"layers.{i}.{nm}.lora_A"] = buf[a[0]:a[1]].view(r, d)
                out[f"layers.{i}.{nm}.lora_B"] = buf[b[0]:b[1]].view(self.douts[j], r)
        return out

    def load_lora(self, lora_sd):
        v = self.lora_views()
        for k, t in lora_sd.items():
            v[k].copy_(t.to(device=self.device, dtype=torch.float32))

    def pack_lora(self):
        """Refresh the padded operand images from the fp32 masters (after every optimizer step): one launch for all adapters,
        driven by a device-side table of (master, image) pointers built on first use."""
        lib = L.load()
        if getattr(self, "_pack_table", None) is None:
            rows = []
            for i in range(self.cfg.layers):
                for j in range(4):
                    a, b = self._slices(i, j)
                    lm = self.layers[i].lora[j]
                    rows.append([self.lora_p[a[0]:a[1]].data_ptr(), self.lora_p[b[0]:b[1]].data_ptr(), lm.A_pad, lm.AT_pad, lm.B_pad, lm.BT_pad,
                                 lm.ld_at, self.douts[j]])
            self._pack_table = torch.tensor(rows, dtype=torch.int64, device=self.device)
        L.check(lib.avllm_lora_pack_batch(L.ptr(self._pack_table), self._pack_table.shape[0], self.r, self.cfg.hidden, self.desc.dtype,
                                          L.stream_ptr()))

    # ------------------------------------------------------------------ training
    def fwd_loss(self, x, labels, want_logits=False, dropout=0.0, seed=0, seed_dev=None):
        """x [B,S,d] (engine dtype), labels int64 [B,S] (-100 applied).  Leaves loss_sum,count in self.acc.
        `dropout`/`seed`: lora_dropout probability and the step's mask seed (kept in the descriptor for bwd()); `seed_dev`: device
        pointer (int) of a uint32 added to `seed` at run time -- the step state of a graph-replayable step."""
        lib = L.load()
        self.desc.lora_dropout = float(dropout) if self.use_lora else 0.0
        self.desc.dropout_seed = int(seed) & 0xFFFFFFFF
        self.desc.dropout_seed_dev = seed_dev
        B, S, _ = x.shape
        x = x.contiguous()
        labels = labels.contiguous()
        ws = self.ws.get(lib.avllm_llama_train_workspace_bytes(C.byref(self.desc), B, S))
        logits = torch.empty(B, S, self.cfg.vocab, device=self.device, dtype=self.dtype) if want_logits else None
        self.acc.zero_()
        L.check(lib.avllm_llama_lora_fwd_loss(C.byref(self.desc), L.ptr(x), L.ptr(labels), B, S, L.ptr(logits), L.ptr(self.acc),
                                              L.ptr(self.acc) + 4, L.ptr(ws), ws.numel(), L.stream_ptr()))
        self._last = (B, S, labels, ws.data_ptr(), ws.numel())
        self.gen = getattr(self, "gen", 0) + 1          # identifies whose activations the workspace holds (checked by _LoraLoss.backward)
        return logits

    def bwd(self, grad_scale=1.0, count=None, after_layer=None, layer_hi=None, layer_lo=0):
        """Accumulates LoRA grads into self.lora_g (zero it first).  `count` = device float tensor holding the
        (possibly all-reduced) number of scored tokens; defaults to this rank's.  layer_hi/layer_lo: run only that piece of the
        backward pass (decoder layers layer_hi .. layer_lo; pieces in descending order, avllm_llama_lora_bwd_layers)."""
        lib = L.load()
        if self._last is None:
            raise RuntimeError("LlamaEngine.bwd() without a preceding fwd_loss()")
        B, S, labels, ws_ptr, ws_len = self._last
        ws = self.ws.buf
        if ws is None or ws.data_ptr() != ws_ptr or ws.numel() != ws_len:
            raise RuntimeError("LlamaEngine.bwd(): the training workspace changed since fwd_loss() (the saved activations are gone)")
        cnt = self.acc[1:2] if count is None else count
        cb = L.LAYER_CB(lambda l, u: after_layer(l)) if after_layer is not None else L.LAYER_CB(0)
        hi = self.cfg.layers - 1 if layer_hi is None else layer_hi
        L.check(lib.avllm_llama_lora_bwd_layers(C.byref(self.desc), L.ptr(labels), B, S, L.ptr(cnt), grad_scale, L.ptr(ws), ws.numel(),
                                                hi, layer_lo, cb, None, L.stream_ptr()))

    # ------------------------------------------------------------------ inference
    def adapters_disabled(self):
        """Context manager: the LLM without its LoRA adapters, as scripts/clip_whisper/decode.py of the reference runs it (it rebuilds the
        model with use_lora=False and loads the connector weights only, decode.py:186-197,236-260)."""
        eng = self

        class _Ctx:
            def __enter__(self_):
                self_.saved = [[(ly.lora[j].A_pad, ly.lora[j].B_pad) for j in range(4)] for ly in eng.layers]
                for ly in eng.layers:
                    for j in range(4):
                        ly.lora[j].A_pad = None
                        ly.lora[j].B_pad = None

            def __exit__(self_, *a):
                for ly, sv in zip(eng.layers, self_.saved):
                    for j in range(4):
                        ly.lora[j].A_pad, ly.lora[j].B_pad = sv[j]
        return _Ctx()

    def frozen_weight_bytes(self):
        """Bytes of frozen weights ONE decode token step streams from HBM: every projection of every layer + lm_head (the embedding
        table contributes B rows, the norms 2 vectors per layer: both negligible and left out)."""
        d, f = self.cfg.hidden, self.cfg.ffn
        per_layer = (d + 2 * self.dkv) * d + d * d + 2 * f * d + d * f
        return (self.cfg.layers * per_layer + self.cfg.vocab * d) * (4 if self.dtype == torch.float32 else 2)

    def decode_is_fused(self, B):
        """True when a token step of B sequences takes the fused bf16 path (tests and benchmarks assert which path they measured)."""
        return bool(L.load().avllm_llama_decode_is_fused(C.byref(self.desc), B))

    def alloc_cache(self, B, Tmax):
        shape = (self.cfg.layers, B, Tmax, self.dkv)
        return torch.empty(shape, dtype=self.dtype, device=self.device), torch.empty(shape, dtype=self.dtype, device=self.device)

    def prefill(self, x, kc, vc, all_logits=False):
        lib = L.load()
        B, S, _ = x.shape
        x = x.contiguous()
        Tmax = kc.shape[2]
        ws = self.ws_infer.get(lib.avllm_llama_infer_workspace_bytes(C.byref(self.desc), B, S))
        last = torch.empty(B, self.cfg.vocab, device=self.device, dtype=torch.float32)
        full = torch.empty(B, S, self.cfg.vocab, device=self.device, dtype=self.dtype) if all_logits else None
        L.check(lib.avllm_llama_prefill(C.byref(self.desc), L.ptr(x), B, S, L.ptr(kc), L.ptr(vc), Tmax, L.ptr(last), L.ptr(full),
                                        L.ptr(ws), ws.numel(), L.stream_ptr()))
        return last, full

    def decode_step(self, ids, pos, kc, vc, pos_dev=None, logits=None):
        """One token per sequence at position `pos` (+ the int32 in device memory `pos_dev`: a captured step replays for every token).
        `logits`: optional preallocated [B, vocab] float32 output."""
        lib = L.load()
        B = ids.shape[0]
        ws = self.ws_infer.get(lib.avllm_llama_infer_workspace_bytes(C.byref(self.desc), B, 1))
        if logits is None:
            logits = torch.empty(B, self.cfg.vocab, device=self.device, dtype=torch.float32)
        L.check(lib.avllm_llama_decode_step_at(C.byref(self.desc), L.ptr(ids.contiguous()), B, pos, L.ptr(pos_dev), L.ptr(kc), L.ptr(vc), kc.shape[2],
                                               L.ptr(logits), L.ptr(ws), ws.numel(), L.stream_ptr()))
        return logits
