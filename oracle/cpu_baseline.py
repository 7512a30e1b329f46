"""CPU baseline leg of bench.py (TEST INFRASTRUCTURE: the oracle timed on the GPU box's host cores).

Times the fp32 CPU restatement (oracle/avsr_oracle.py) on a BOUNDED sample of the config-2 train step and
extrapolates by the layer/frame counts it skipped; the sample is reported verbatim in the JSON.  It is a
reported baseline ("port"), not an optimisation target.
"""
from __future__ import annotations

import os
import time

import torch

from . import avsr_oracle as O
from . import weights as Wt


def run(cfg: Wt.ModelCfg | None = None, frames: int = 125, sample_frames: int = 12, whisper_layers: int = 4, threads: int | None = None):
    cfg = cfg or Wt.config2()
    # a 1-GPU box's CPU share is 16 cores; more torch threads than that only adds contention
    threads = threads or min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    t_all = time.time()
    # ---- Whisper: conv stem + `whisper_layers` of the encoder layers, B=1
    wc = Wt.WhisperCfg(**{**vars(cfg.whisper), "layers": whisper_layers})
    Ww = Wt.whisper_weights(wc, 0)
    mel = torch.randn(1, 80, 3000)
    with torch.no_grad():
        t0 = time.time(); O.whisper_encoder(Ww, wc, mel); t_w = time.time() - t0
        wc0 = Wt.WhisperCfg(**{**vars(cfg.whisper), "layers": 0})
        t0 = time.time(); O.whisper_encoder(Ww, wc0, mel); t_stem = time.time() - t0
    per_wl = max(0.0, (t_w - t_stem) / whisper_layers)
    whisper_s = t_stem + per_wl * cfg.whisper.layers
    # ---- CLIP: full depth on `sample_frames` frames
    Wc = Wt.clip_weights(cfg.clip, 0)
    fr = torch.randn(sample_frames, 3, cfg.clip.image, cfg.clip.image)
    with torch.no_grad():
        t0 = time.time(); O.clip_vision_cls(Wc, cfg.clip, fr); t_c = time.time() - t0
    clip_s = t_c * frames / sample_frames
    # ---- Llama: ONE decoder layer + lm_head + CE, forward and backward (LoRA grads), S=256, B=1
    lc = Wt.LlamaCfg(**{**vars(cfg.llama), "layers": 1})
    Wl = Wt.llama_weights(lc, 0)
    lora = {k: v.requires_grad_(True) for k, v in Wt.lora_weights(lc, cfg.lora, 0, 0.01).items()}
    x = torch.randn(1, 256, lc.hidden)
    labels = torch.randint(3, lc.vocab, (1, 256))
    t0 = time.time()
    h = O.llama_hidden(Wl, lora, lc, cfg.lora, x)
    logits = h @ Wl["lm_head.weight"].T
    loss = O.causal_lm_loss(logits, labels)
    t_f = time.time() - t0
    t0 = time.time(); loss.backward(); t_b = time.time() - t0
    with torch.no_grad():
        t0 = time.time(); (h.detach() @ Wl["lm_head.weight"].T); t_head = time.time() - t0
    # the head appears once; a layer's share is what is left
    layer_f = max(0.0, t_f - t_head)
    layer_b = max(0.0, t_b - 2 * t_head)
    llama_s = (layer_f + layer_b) * cfg.llama.layers + 3 * t_head
    total = whisper_s + clip_s + llama_s
    return {
        "value": 1.0 / total, "unit": "samples/s", "cores": threads, "kind": "port",
        "sample": (f"B=1: Whisper stem+{whisper_layers}/{cfg.whisper.layers} layers, CLIP {sample_frames}/{frames} frames, "
                   f"1/{cfg.llama.layers} Llama layer fwd+bwd + lm_head/CE at S=256, fp32 torch CPU oracle; extrapolated by "
                   f"layer/frame counts"),
        "split_s": {"whisper": round(whisper_s, 3), "clip": round(clip_s, 3), "llama_fwd_bwd": round(llama_s, 3)},
        "measured_s": round(time.time() - t_all, 2),
    }


if __name__ == "__main__":
    import json
    print(json.dumps(run()))
