#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (build container only): a checkpoint in the REFERENCE TRAINER's own format.

Builds a micro clip_whisper model out of the reference's own `ClipWhisperModel` glue (loaded by file path, as oracle/make_golden.py
does), hands it to the reference's own `ClipWhisperTrainer` methods -- `_setup_optimizer` (trainer/clip_whisper_trainer.py:171-232),
two optimizer steps exactly as the hot loop runs them (:433-464: forward, backward, clip_grad_norm_, AdamW step, zero_grad, cosine
scheduler step) and `_save_checkpoint(is_best=True)` (:725-794) -- and keeps what that wrote:

    tests/golden/n1_reference_model_best.pt      the file torch.save produced inside _save_checkpoint, untouched
    tests/golden/n1_expected.npz                 the reference's train loss / logits on a third batch from the restored state, and the
                                                 LoRA tensors after its third optimizer step (what a resumed run must reproduce)

peft is not installed here, so the LoRA wrap is a stand-in whose MODULE TREE spells peft's published key names
(`llm.base_model.model.model.layers.N.self_attn.q_proj.{base_layer.weight, lora_A.default.weight, lora_B.default.weight}`): the key
names in the file are therefore peft's, the arithmetic is the restated lora.Linear (parity unpinned against real peft).
The batch is not stored: oracle/weights.py regenerates it from its seed."""
import importlib.util
import os
import shutil
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import weights as Wt  # noqa: E402
from oracle.make_golden import build_reference_model, load_reference  # noqa: E402

TRAINER = "/root/reference/src/clip_whisper/trainer/clip_whisper_trainer.py"
OUT = os.path.join(ROOT, "tests", "golden")
LR, WD, CLIP, TOTAL_STEPS = 1e-3, 0.01, 0.5, 10


def micro_cfg():
    return Wt.ModelCfg(whisper=Wt.WhisperCfg(d_model=64, heads=1, layers=1, ffn=64), clip=Wt.ClipCfg(hidden=64, heads=1, layers=1, mlp=64, image=32, patch=16),
                       llama=Wt.LlamaCfg(hidden=128, heads=1, layers=1, ffn=128, vocab=64), lora=Wt.LoraCfg(r=8, alpha=16.0), max_seq_len=512)


class PeftLoraLinear(nn.Module):
    """Module tree of peft's lora.Linear: base_layer + lora_A/lora_B ModuleDicts keyed by the adapter name "default"."""

    def __init__(self, base, A, B, scale):
        super().__init__()
        self.base_layer = base
        base.weight.requires_grad_(False)
        self.lora_A = nn.ModuleDict({"default": nn.Linear(A.shape[1], A.shape[0], bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(B.shape[1], B.shape[0], bias=False)})
        self.lora_A["default"].weight.data.copy_(A)
        self.lora_B["default"].weight.data.copy_(B)
        self.scale = scale

    def forward(self, x):
        return self.base_layer(x) + self.lora_B["default"](self.lora_A["default"](x)) * self.scale


class PeftModelStandIn(nn.Module):
    """PeftModel(base_model=LoraModel(model=llm)): gives the state-dict prefix `base_model.model.` and forwards the calls the reference makes."""

    def __init__(self, llm):
        super().__init__()
        self.base_model = nn.Module()
        self.base_model.model = llm

    def forward(self, *a, **k):
        return self.base_model.model(*a, **k)

    def get_input_embeddings(self):
        return self.base_model.model.get_input_embeddings()

    def generate(self, *a, **k):
        return self.base_model.model.generate(*a, **k)

    @property
    def config(self):
        return self.base_model.model.config


def build(cw, cfg, W):
    m = build_reference_model(cw, cfg, W)                       # make_golden's LoraLinear wrap ...
    llm = m.llm
    for i, layer in enumerate(llm.model.layers):                 # ... re-wrapped with peft's module names
        for nm in cfg.lora.targets:
            old = getattr(layer.self_attn, nm)
            setattr(layer.self_attn, nm, PeftLoraLinear(old.base, old.lora_A.data, old.lora_B.data, cfg.lora.scale))
    m.llm = PeftModelStandIn(llm)
    return m


def main():
    # the trainer module first: it imports transformers' schedulers, whose import probes for a real `peft` package and trips over the stub
    # module that load_reference() registers for the model file
    spec = importlib.util.spec_from_file_location("ref_trainer", TRAINER)
    tr_mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tr_mod)
    T = tr_mod.ClipWhisperTrainer
    cw = load_reference()
    cfg = micro_cfg()
    W = Wt.all_weights(cfg, 31, lora_b_std=0.05)
    m = build(cw, cfg, W)
    m.train()
    tmp = tempfile.mkdtemp()
    t = T.__new__(T)
    t.model, t.learning_rate, t.weight_decay, t.max_epochs, t.warmup_steps = m, LR, WD, 1, 0
    t.train_dataloader = [None] * TOTAL_STEPS                    # only len() is read (T_max of the cosine schedule)
    t.output_dir, t.device, t.grad_clip, t.current_epoch = tmp, "cpu", CLIP, 0
    t.train_losses, t.val_losses = [], []
    T._setup_optimizer(t)

    def step(seed):
        audio, video, labels, prompt = Wt.synthetic_batch(cfg, 2, 3, seed=seed)
        loss = m(audio=audio, video=video, prompt=prompt, labels=labels)["loss"]          # trainer :705 model(**batch_dict)
        loss.backward()                                                                    # :454
        torch.nn.utils.clip_grad_norm_(m.parameters(), t.grad_clip)                        # :458
        t.optimizer.step(); t.optimizer.zero_grad(); t.scheduler.step()                    # :461-464
        return float(loss)

    t.train_losses = [step(101), step(102)]
    t.val_losses = [t.train_losses[-1]]
    path = T._save_checkpoint(t, is_best=True, checkpoint_dir=tmp)
    assert os.path.basename(path) == "model_best.pt"
    os.makedirs(OUT, exist_ok=True)
    dst = os.path.join(OUT, "n1_reference_model_best.pt")
    shutil.copy(path, dst)
    ck = torch.load(dst, map_location="cpu", weights_only=True)                            # the safe loader must accept the reference's own file
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "train_losses", "val_losses", "best_val_loss"}
    keys = list(ck["model_state_dict"])
    assert "llm.base_model.model.model.layers.0.self_attn.q_proj.lora_A.default.weight" in keys and "audio_connector.linear.weight" in keys
    # what a run resumed from that file must reproduce: the third batch's forward, and the parameters after the third step
    audio, video, labels, prompt = Wt.synthetic_batch(cfg, 2, 3, seed=103)
    with torch.no_grad():
        out = m(audio=audio, video=video, prompt=prompt, labels=labels)
    exp = {"train_loss_3": np.array(float(out["loss"])), "train_logits_3_sub": out["logits"][:, ::8].numpy().copy(),
           "losses_12": np.array(t.train_losses), "lr": np.array(LR), "wd": np.array(WD), "clip": np.array(CLIP), "total_steps": np.array(TOTAL_STEPS),
           "seeds": np.array([101, 102, 103]), "weights_seed": np.array(31)}
    step(103)
    for k, v in m.state_dict().items():
        if "lora_" in k:
            exp["after3." + k] = v.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "n1_expected.npz"), **exp)
    print("wrote", dst, os.path.getsize(dst), "bytes;", len(keys), "tensors; optimizer state entries:", len(ck["optimizer_state_dict"]["state"]),
          "param groups:", [len(g["params"]) for g in ck["optimizer_state_dict"]["param_groups"]], "scheduler keys:", sorted(ck["scheduler_state_dict"]))
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
