#!/usr/bin/env python3
"""Train entry point with the reference's flag names (scripts/clip_whisper/train.py:33-81); `--synthetic N` replaces the
LRS3 manifests with N seeded synthetic clips (no dataset offline).  Launch one process per GPU with torchrun for DDP."""
import argparse
import logging
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-llm_amd")):
    sys.path.insert(0, p) if p not in sys.path else None

import torch  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser()
    p.add_argument("--config", default=os.path.join(ROOT, "configs", "clip_whisper.yaml"))
    p.add_argument("--output_dir"); p.add_argument("--data_path"); p.add_argument("--llm_path")
    p.add_argument("--whisper_model"); p.add_argument("--clip_model")
    p.add_argument("--modality", choices=["audio", "video", "both"])
    p.add_argument("--batch_size", type=int); p.add_argument("--max_epochs", type=int)
    p.add_argument("--learning_rate", type=float); p.add_argument("--max_seq_len", type=int)
    p.add_argument("--fp16", action="store_true", default=None); p.add_argument("--use_4bit", action="store_true", default=None)
    p.add_argument("--no_lora", action="store_true"); p.add_argument("--lora_r", type=int); p.add_argument("--lora_alpha", type=int)
    p.add_argument("--connector_type", default=None); p.add_argument("--max_grad_norm", type=float)
    p.add_argument("--log_interval", type=int); p.add_argument("--save_every", type=int); p.add_argument("--resume_from")
    p.add_argument("--save_steps", type=int, default=None, help="stored by the trainer and never acted on, as in the reference (clip_whisper_trainer.py:94-102)")
    p.add_argument("--log_param_updates", action="store_true", help="accepted for command-line compatibility; the reference stores it and never reads it (:118)")
    p.add_argument("--seed", type=int); p.add_argument("--synthetic", type=int, default=0, help="train on N synthetic clips")
    p.add_argument("--frames", type=int, default=125); p.add_argument("--tiny", action="store_true")
    p.add_argument("--synthetic-weights", action="store_true",
                   help="seeded random weights + byte tokenizer when the model paths are not local checkpoints (implied by --tiny); "
                        "without it such a path is an error")
    return p.parse_args()


class SyntheticClips(torch.utils.data.Dataset):
    """Batches in the reference's 4-tuple layout (audio, video, texts, labels), simple_dataset.py:455."""
    def __init__(self, n, cfg, frames, tok, seed=0):
        self.n, self.cfg, self.frames, self.tok, self.seed = n, cfg, frames, tok, seed
    def __len__(self):
        return self.n
    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 100003 + i)
        audio = torch.randn(80, 2 * self.cfg.whisper.n_ctx, generator=g)
        video = torch.randn(self.frames, 3, self.cfg.clip.image, self.cfg.clip.image, generator=g)
        words = " ".join("w%d" % int(x) for x in torch.randint(0, 50, (int(torch.randint(3, 9, (1,), generator=g)),), generator=g))
        ids = self.tok([words], padding="max_length", max_length=256, truncation=True).input_ids[0]
        return audio, video, words, ids
    @staticmethod
    def collate(b):
        return torch.stack([x[0] for x in b]), torch.stack([x[1] for x in b]), [x[2] for x in b], torch.stack([x[3] for x in b])


def main():
    a = parse_args()
    from avllm.config import merged
    cfg = merged(a.config, {"output_dir": a.output_dir, "path": a.data_path, "llm_path": a.llm_path, "whisper_model": a.whisper_model,
                            "clip_model": a.clip_model, "modality": a.modality, "batch_size": a.batch_size, "num_epochs": a.max_epochs,
                            "learning_rate": a.learning_rate, "max_seq_len": a.max_seq_len, "use_fp16": a.fp16, "use_4bit": a.use_4bit,
                            "lora_r": a.lora_r, "lora_alpha": a.lora_alpha, "connector_type": a.connector_type,
                            "max_grad_norm": a.max_grad_norm, "log_interval": a.log_interval, "save_every": a.save_every, "seed": a.seed,
                            "save_steps": a.save_steps, "log_param_updates": a.log_param_updates or None})
    os.makedirs(cfg["output_dir"], exist_ok=True)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(levelname)s %(message)s",
                        handlers=[logging.StreamHandler(), logging.FileHandler(os.path.join(cfg["output_dir"], "training.log"))])
    world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    torch.manual_seed(cfg.get("seed", 42))
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer
    kw = {}
    if a.tiny:
        from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
        kw["config"] = ModelCfg(WhisperCfg(128, 2, 2, 256), ClipCfg(128, 2, 2, 256, 48, 16), LlamaCfg(256, 2, 2, 512, 512), LoraCfg(16, 32.0))
    model = ClipWhisperModel(llm_path=cfg["llm_path"], whisper_model=cfg["whisper_model"], clip_model=cfg["clip_model"], device=f"cuda:{local}",
                             use_fp16=bool(cfg.get("use_fp16")), use_4bit=bool(cfg.get("use_4bit")), use_lora=not a.no_lora,
                             lora_r=cfg.get("lora_r", 16), lora_alpha=cfg.get("lora_alpha", 32), lora_dropout=cfg.get("lora_dropout", 0.05),
                             freeze_encoders=cfg.get("freeze_encoders", True), modality=cfg.get("modality", "both"),
                             max_seq_len=cfg.get("max_seq_len", 256), fusion_scale=cfg.get("fusion_scale", 0.5),
                             connector_type=cfg.get("connector_type", "simple"), synthetic_weights=a.synthetic_weights or a.tiny, **kw)
    if a.synthetic:
        frames = a.frames if not a.tiny else 5
        ds = SyntheticClips(a.synthetic, model.cfg, frames, model.tokenizer, cfg.get("seed", 42))
        sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=True) if world > 1 else None
        dl = torch.utils.data.DataLoader(ds, batch_size=cfg.get("batch_size", 4), shuffle=sampler is None, sampler=sampler, collate_fn=ds.collate)
        vdl = torch.utils.data.DataLoader(SyntheticClips(max(2, a.synthetic // 8), model.cfg, frames, model.tokenizer, 7), batch_size=cfg.get("batch_size", 4), collate_fn=ds.collate)
    else:
        # LRS3-style manifests (configs/clip_whisper.yaml data.path / train_manifest / train_labels): raw samples, features on the device
        from avllm.data import AVSRDataset, create_dataloaders
        root = cfg.get("path") or "."
        mp = os.path.join(root, cfg.get("train_manifest", "train.tsv")); lp = os.path.join(root, cfg.get("train_labels", "train.wrd"))
        sampler = None
        if world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(AVSRDataset(mp, lp, root, model.tokenizer, modality=cfg.get("modality", "both")), shuffle=True)
        dl, vdl = create_dataloaders(mp, lp, root, model.tokenizer, batch_size=cfg.get("batch_size", 4), num_workers=cfg.get("num_workers", 0),
                                     modality=cfg.get("modality", "both"), max_video_length=cfg.get("max_video_length", 300), sampler=sampler)
    tr = ClipWhisperTrainer(model, dl, vdl, learning_rate=float(cfg.get("learning_rate", 5e-5)), weight_decay=float(cfg.get("weight_decay", 0.01)),
                            max_epochs=cfg.get("num_epochs", 10), output_dir=cfg["output_dir"], device=f"cuda:{local}", fp16=bool(cfg.get("use_fp16")),
                            grad_accum_steps=cfg.get("grad_accum_steps", 1), log_interval=cfg.get("log_interval", 10), save_every=cfg.get("save_every", 1),
                            grad_clip=float(cfg.get("max_grad_norm", 0.5)), warmup_steps=cfg.get("warmup_steps", 0),
                            save_steps=cfg.get("save_steps", None), log_param_updates=bool(cfg.get("log_param_updates", False)))
    if a.resume_from:
        tr.load_checkpoint(a.resume_from)
    print(tr.train())


if __name__ == "__main__":
    main()
