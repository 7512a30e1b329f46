#!/usr/bin/env python3
"""Where does a NaN in the mel input go?  Prints the NaN count after each stage of the tiny model (debug aid for the non-finite guard test)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-llm_amd")]
import torch
from oracle import weights as Wt
from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
from avllm.model import ClipWhisperModel

oc = Wt.tiny()
W = Wt.all_weights(oc, 0, lora_b_std=0.05)
audio, video, labels, prompt = Wt.synthetic_batch(oc, 2, 3, seed=3)
cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
for precision in ("fp32", "bf16"):
    m = ClipWhisperModel(device="cuda:0", lora_r=16, lora_alpha=32, lora_dropout=0.0, max_seq_len=512, config=cfg, weights=W, precision=precision).train()
    bad = audio.clone(); bad[0, 3, 100] = float("nan")
    dev = "cuda:0"
    h = m.whisper_engine.forward(bad.to(dev))
    print(precision, "whisper out nan rows:", int(torch.isnan(h.float()).any(-1).sum()), "of", h.shape[0] * h.shape[1])
    a = m.encode_audio(bad.to(dev), rows=512)
    print(precision, "connector out nan:", int(torch.isnan(a.float()).sum()))
    lab = m._prep_labels(labels)
    x = m._llm_inputs(bad.to(dev), video.to(dev), prompt.to(dev), S_out=lab.shape[1])
    print(precision, "llm inputs nan rows per clip:", torch.isnan(x.float()).any(-1).sum(-1).tolist())
    logits = m.llm_engine.fwd_loss(x, lab, want_logits=True)
    print(precision, "logits nan rows per clip:", torch.isnan(logits.float()).any(-1).sum(-1).tolist(), "acc", m.llm_engine.acc.tolist())
