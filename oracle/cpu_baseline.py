"""CPU baseline leg of bench.py (TEST INFRASTRUCTURE: the oracle timed on the GPU box's host cores).

Times the fp32 CPU restatement (oracle/avsr_oracle.py) on a BOUNDED sample of the config-2 train step and
extrapolates by the layer/frame counts it skipped; the sample is reported verbatim in the JSON.  It is a
reported baseline ("port"), not an optimisation target.
"""
from __future__ import annotations

import os
import time

import torch

if __package__ in (None, ""):                       # `python oracle/cpu_baseline.py [--full]`
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import avsr_oracle as O
    from oracle import weights as Wt
else:
    from . import avsr_oracle as O
    from . import weights as Wt


def cpu_model():
    """The host CPU's model string ("core count and CPU model printed in the report")."""
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def run(cfg: Wt.ModelCfg | None = None, frames: int = 125, sample_frames: int = 32, whisper_layers: int = 4, threads: int | None = None):
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
        "value": 1.0 / total, "unit": "samples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
        "sample": (f"B=1: Whisper stem+{whisper_layers}/{cfg.whisper.layers} layers, CLIP {sample_frames}/{frames} frames, "
                   f"1/{cfg.llama.layers} Llama layer fwd+bwd + lm_head/CE at S=256, fp32 torch CPU oracle; extrapolated by "
                   f"layer/frame counts"),
        "split_s": {"whisper": round(whisper_s, 3), "clip": round(clip_s, 3), "llama_fwd_bwd": round(llama_s, 3)},
        "measured_s": round(time.time() - t_all, 2),
    }


def shared_depth_weights(cfg: Wt.ModelCfg, seed: int = 0, lora_b_std: float = 0.01, distinct_lora: bool = False):
    """Weights of the full-depth model with the decoder layers sharing ONE set of frozen tensors (same arithmetic and cache behaviour per
    layer, 1/layers of the 27 GB an fp32 Llama-2-7B would need on the host).  The adapters are per layer: copies of one draw, or -- with
    `distinct_lora` -- independent draws (tests/test_pin_bf16_gpu.py's full-depth pin)."""
    one = Wt.LlamaCfg(**{**vars(cfg.llama), "layers": 1})
    L1 = Wt.llama_weights(one, seed)
    llama = {k: v for k, v in L1.items() if "layers." not in k}
    lora = {}
    for i in range(cfg.llama.layers):
        for k, v in L1.items():
            if k.startswith("model.layers.0."):
                llama[k.replace("model.layers.0.", f"model.layers.{i}.")] = v
        for k, v in Wt.lora_weights(one, cfg.lora, seed + (i if distinct_lora else 0), lora_b_std).items():
            lora[k.replace("layers.0.", f"layers.{i}.")] = v.clone()
    return {"whisper": Wt.whisper_weights(cfg.whisper, seed), "clip": Wt.clip_weights(cfg.clip, seed), "llama": llama, "lora": lora,
            "audio_connector": Wt.connector_weights(cfg.whisper.d_model, cfg.llama.hidden, "conn.audio", seed),
            "video_connector": Wt.connector_weights(cfg.clip.hidden, cfg.llama.hidden, "conn.video", seed)}


def run_full(cfg: Wt.ModelCfg | None = None, frames: int = 125, threads: int | None = None):
    """ONE complete B=1 train step of the oracle at full depth and width (BASELINE.md §2 protocol, minus the optimizer's microseconds):
    every encoder layer, every frame, all decoder layers forward and backward.  The 32 decoder layers share ONE set of random tensors
    (shared_depth_weights)."""
    cfg = cfg or Wt.config2()
    threads = threads or min(16, os.cpu_count() or 1)
    torch.set_num_threads(threads)
    t_all = time.time()
    W = shared_depth_weights(cfg)
    audio, video, labels, prompt = Wt.synthetic_batch(cfg, 1, frames, seed=1234)
    t_setup = time.time() - t_all
    split = {}
    with torch.no_grad():
        t0 = time.time(); O.whisper_encoder(W["whisper"], cfg.whisper, audio); split["whisper"] = round(time.time() - t0, 3)
        t0 = time.time(); O.clip_vision_cls(W["clip"], cfg.clip, video.reshape(-1, 3, cfg.clip.image, cfg.clip.image)); split["clip"] = round(time.time() - t0, 3)
    t0 = time.time()
    O.train_step_grads(W, cfg, audio, video, prompt, labels)
    step_s = time.time() - t0
    split["llama_fwd_bwd"] = round(step_s - split["whisper"] - split["clip"], 3)
    return {"value": 1.0 / step_s, "unit": "samples/s", "cores": threads, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"one full B=1 train step ({frames} frames, all {cfg.whisper.layers}+{cfg.clip.layers}+{cfg.llama.layers} layers, forward + backward), fp32 torch CPU "
                      "oracle; the decoder layers share one set of random tensors", "split_s": split, "measured_s": round(time.time() - t_all, 2),
            "setup_s": round(t_setup, 2)}


if __name__ == "__main__":
    import json
    import sys
    print(json.dumps(run_full() if "--full" in sys.argv else run()))
