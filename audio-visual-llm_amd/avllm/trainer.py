"""ClipWhisperTrainer -- mirror of src/clip_whisper/trainer/clip_whisper_trainer.py:19-1017 for the hot loop
(`_train_epoch` :412-524, `_process_batch` :604-723, `_setup_optimizer` :171-232, `_validate` :526-603) with the
same constructor keywords, batch layouts, return dict and output files.  The step is:
  encoders (HIP) -> fuse/pool (HIP) -> Llama+LoRA fwd+loss (HIP) -> bwd (HIP) [-> RCCL all-reduce, overlapped]
  -> clip_grad_norm_ + AdamW fused on the flat LoRA buffer (HIP) -> cosine LR.
No `loss.item()` inside the step: the loss stays on the device and is read once per log interval.

The step is graph-replayable: everything that changes from step to step (learning rate, Adam's bias corrections, the LoRA dropout
seed) lives in a small device record advanced by a one-thread kernel at the top of the step (include/avllm.h avllm_step_state), so the
~1,400 kernel launches of a step are the same sequence every time.  With `use_graph` (default on) the step is captured once per input
signature into hipGraphs (torch.cuda.CUDAGraph owns capture + memory pool) and replayed: one graph when single-process; under data
parallelism forward / backward pieces / optimizer are separate graphs with the RCCL collectives issued between them, each backward
piece's gradient all-reduce running on the side stream under the next piece.
"""
from __future__ import annotations

import csv
import ctypes
import json
import logging
import math
import os
import time

import torch

from . import lib as L
from . import ops
from .dist import LoraGradReducer, is_dist
from .engine import Workspace


class ClipWhisperTrainer:
    def __init__(self, model, train_dataloader=None, val_dataloader=None, learning_rate=5e-5, weight_decay=0.01, max_epochs=10,
                 output_dir="outputs/clip_whisper", device="cuda", fp16=False, grad_accum_steps=1, log_interval=10, save_every=1,
                 save_steps=None, grad_clip=0.5, warmup_steps=0, log_param_updates=False, total_steps=None, use_graph=None,
                 bwd_pieces=None):
        self.model, self.train_dataloader, self.val_dataloader = model, train_dataloader, val_dataloader
        self.learning_rate, self.weight_decay, self.max_epochs = learning_rate, weight_decay, max_epochs
        self.output_dir, self.device, self.fp16 = output_dir, device, fp16
        self.grad_accum_steps = grad_accum_steps           # stored but unused, as in the reference
        self.log_interval, self.save_every, self.save_steps = log_interval, max(1, save_every or 1), save_steps
        self.grad_clip, self.warmup_steps = grad_clip, warmup_steps
        self.global_step = 0
        self.train_losses, self.val_losses, self.best_val_loss = [], [], float("inf")
        if total_steps is None:
            total_steps = max_epochs * (len(train_dataloader) if train_dataloader is not None else 1)
        self.total_steps = max(1, total_steps)
        eng = model.llm_engine
        self.m = torch.zeros_like(eng.lora_p)
        self.v = torch.zeros_like(eng.lora_p)
        self.sumsq = torch.zeros(1, device=eng.lora_p.device, dtype=torch.float32)
        self._sumsq_parts = torch.zeros(1024, device=eng.lora_p.device, dtype=torch.float32)      # fixed-order gradient norm: replicas stay bit-identical
        self.skipped = torch.zeros(1, device=eng.lora_p.device, dtype=torch.float32)      # optimizer steps skipped on a non-finite loss / gradient
        self.reducer = LoraGradReducer(eng.lora_g, eng.per_layer, eng.cfg.layers)
        dev = eng.lora_p.device
        # avllm_step_state on the device: step count, this step's lr / bias corrections / dropout seed
        self.state = torch.zeros(ctypes.sizeof(L.StepState), dtype=torch.uint8, device=dev)
        self._seed_ptr = self.state.data_ptr() + L.StepState.dropout_seed.offset
        self._loss_buf = torch.zeros((), dtype=torch.float32, device=dev)
        self._rank = torch.distributed.get_rank() if is_dist() else 0
        if use_graph is None:
            use_graph = os.environ.get("AVLLM_GRAPH", "1") != "0"
        self.use_graph = bool(use_graph) and dev.type == "cuda"
        # data parallel + graphs: the backward pass is replayed in pieces so that a piece's gradient all-reduce overlaps the next piece
        self.bwd_pieces = max(1, min(eng.cfg.layers, bwd_pieces if bwd_pieces is not None else (4 if is_dist() else 1)))
        self._graphs = {}

    # ---- _setup_optimizer :171-232: AdamW(beta 0.9/0.95, eps 1e-8); cosine (with optional linear warmup)
    def lr_at(self, step):
        if self.warmup_steps > 0:
            if step < self.warmup_steps:
                return self.learning_rate * step / max(1, self.warmup_steps)
            prog = (step - self.warmup_steps) / max(1, self.total_steps - self.warmup_steps)
            return self.learning_rate * max(0.0, 0.5 * (1.0 + math.cos(math.pi * prog)))
        return self.learning_rate * (1 + math.cos(math.pi * step / self.total_steps)) / 2

    # ---- _process_batch :604-680
    def _unpack(self, batch):
        if isinstance(batch, dict) and "raw" in batch:
            # raw samples from avllm.data.AVSRDataset: log-mel / layer norm / CLIP resize+normalise happen here, on the device
            from .preprocess import ClipFrames, WhisperLogMel, device_collate
            if getattr(self, "_featurizers", None) is None:
                dev = self.model.device
                self._featurizers = (WhisperLogMel(dev, n_mels=self.model.cfg.whisper.n_mels), ClipFrames(dev, image=self.model.cfg.clip.image))
            audio, video = device_collate(batch["raw"], *self._featurizers)
            tok = self.model.tokenizer(batch["texts"], return_tensors="pt", padding=True, truncation=True, max_length=self.model.max_seq_len)
            return audio, video, batch["labels"], tok.input_ids
        if isinstance(batch, dict):
            return batch.get("audio"), batch.get("video"), batch.get("labels"), batch.get("prompt")
        audio, video, texts, labels = batch
        tok = self.model.tokenizer(texts, return_tensors="pt", padding=True, truncation=True, max_length=self.model.max_seq_len)
        bs = labels.shape[0]
        for t in (audio, video):
            if t is not None and t.shape[0] != bs:
                raise AssertionError(f"Batch size mismatch: {t.shape[0]} vs labels {bs}")
        return audio, video, labels, tok.input_ids

    # ---- the step, in capture-safe parts (no host sync, no host-side per-step scalars)
    def _part_fwd(self, audio, video, labels, prompt):
        model, eng = self.model, self.model.llm_engine
        ops.step_advance(self.state, self.learning_rate, self.total_steps, self.warmup_steps, rank=self._rank)
        labels = model._prep_labels(labels)
        x = model._llm_inputs(audio, video, prompt, S_out=labels.shape[1])
        p = float(model.lora_dropout) if (model.training and model.use_lora and model.lora_dropout) else 0.0
        eng.fwd_loss(x, labels, dropout=p, seed=0, seed_dev=self._seed_ptr)

    def _piece_range(self, i):
        """Decoder layers (hi, lo) of backward piece i of self.bwd_pieces, last layer first."""
        n, k = self.model.llm_engine.cfg.layers, self.bwd_pieces
        hi = n - 1 - (i * n) // k
        lo = n - ((i + 1) * n) // k
        return hi, lo

    def _part_bwd(self, i, per_layer_cb=None):
        eng = self.model.llm_engine
        if i == 0:
            eng.lora_g.zero_()
        hi, lo = self._piece_range(i)
        eng.bwd(grad_scale=1.0, count=eng.acc[1:2], after_layer=per_layer_cb, layer_hi=hi, layer_lo=lo)

    def _part_opt(self):
        eng = self.model.llm_engine
        ops.grad_sumsq(eng.lora_g, self.sumsq, partials=self._sumsq_parts)
        # NaN/Inf guard of trainer :444-452 without a host sync: a non-finite (all-reduced) loss sum or gradient norm makes the update a
        # no-op on every rank alike (the all-reduce spreads the NaN), leaving lora_p, m and v untouched; `skipped_steps` counts them and
        # the device-side step count goes back by one, so neither the LR schedule nor Adam's bias corrections advance (as in the reference,
        # which runs neither optimizer.step() nor scheduler.step() on such a batch).
        ops.adamw_step(eng.lora_p, eng.lora_g, self.m, self.v, 0.0, 0, sumsq=self.sumsq, max_norm=self.grad_clip or 0.0, wd=self.weight_decay,
                       guard=eng.acc[0:1], skipped=self.skipped, state=self.state)
        eng.pack_lora()
        torch.div(eng.acc[0], eng.acc[1], out=self._loss_buf)

    def _eager_step(self, audio, video, labels, prompt):
        eng = self.model.llm_engine
        self._part_fwd(audio, video, labels, prompt)
        self.reducer.reduce_counts(eng.acc)
        if self.reducer.enabled and not self.use_graph:
            # graphs off on every rank (use_graph is configuration, identical across ranks): one bucket per decoder layer, launched from
            # the C callback as that layer's kernels are enqueued
            saved, self.bwd_pieces = self.bwd_pieces, 1
            try:
                self._part_bwd(0, per_layer_cb=self.reducer.layer_done)
            finally:
                self.bwd_pieces = saved
        else:
            # with graphs on, a rank may be in this eager step (first sight of an input signature, more signatures than graph slots, a
            # dropped graph) while another replays: both issue the SAME collectives -- one all-reduce per backward piece, same slices,
            # same order (_replay) -- so ranks never have to agree on eager vs replay.
            for i in range(self.bwd_pieces):
                self._part_bwd(i)
                hi, lo = self._piece_range(i)
                self.reducer.layers_done(lo, hi)
        self.reducer.finish()
        self._part_opt()

    def _capture(self, audio, video, labels, prompt):
        """Capture the step for this input signature.  Static input buffers are allocated here (the caller's tensors are copied into
        them before every replay -- or ARE them, see static_inputs())."""
        dev = self.model.llm_engine.lora_p.device
        st = {"inputs": [None if t is None else torch.empty(t.shape, dtype=t.dtype, device=dev) for t in (audio, video, labels, prompt)]}
        for dst, src in zip(st["inputs"], (audio, video, labels, prompt)):
            if dst is not None:
                dst.copy_(src)
        st["generation"] = Workspace.generation          # the graphs below hold raw pointers into the engines' workspaces as they are NOW
        pool = torch.cuda.graph_pool_handle()
        a, v, lab, pr = st["inputs"]
        torch.cuda.synchronize()
        if not self.reducer.enabled:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                self._part_fwd(a, v, lab, pr)
                for i in range(self.bwd_pieces):
                    self._part_bwd(i)
                self._part_opt()
            st["all"] = g
        else:
            st["fwd"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["fwd"], pool=pool):
                self._part_fwd(a, v, lab, pr)
            st["bwd"] = []
            for i in range(self.bwd_pieces):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, pool=pool):
                    self._part_bwd(i)
                st["bwd"].append(g)
            st["opt"] = torch.cuda.CUDAGraph()
            with torch.cuda.graph(st["opt"], pool=pool):
                self._part_opt()
        return st

    def _replay(self, st):
        if "all" in st:
            st["all"].replay()
            return
        eng = self.model.llm_engine
        st["fwd"].replay()
        self.reducer.reduce_counts(eng.acc)
        for i, g in enumerate(st["bwd"]):
            g.replay()
            hi, lo = self._piece_range(i)
            self.reducer.layers_done(lo, hi)
        self.reducer.finish()
        st["opt"].replay()

    def static_inputs(self, audio, video, labels, prompt):
        """The device buffers a captured step reads for inputs of this signature (None before it has been captured).  A feeder that
        writes its batches straight into them (H2D or a device-side producer) saves the per-step copy."""
        st = self._graphs.get(self._signature(audio, video, labels, prompt))
        return st["inputs"] if isinstance(st, dict) else None

    @staticmethod
    def _signature(*tensors):
        return tuple(None if t is None else (tuple(t.shape), t.dtype) for t in tensors)

    def train_step(self, audio, video, labels, prompt, graph=None):
        """One optimizer step on this rank's batch; returns the (global) mean loss as a device scalar.
        graph=False forces the eager launch sequence for this call (bench.py's instrumented steps)."""
        use_graph = self.use_graph if graph is None else (graph and self.use_graph)
        self.global_step += 1
        if isinstance(labels, list):
            labels = self.model._prep_labels(labels).cpu()
        st = None
        if use_graph:
            key = self._signature(audio, video, labels, prompt)
            st = self._graphs.get(key)
            if isinstance(st, dict) and st["generation"] != Workspace.generation:
                # some workspace was reallocated since capture (a larger batch / more frames / generate() in between): every captured
                # graph may point into freed memory.  Drop them all; each signature is re-captured at its next-but-one visit.
                self._graphs = {k: "warm" for k in self._graphs}
                st = "stale"                             # this call runs eager and re-sizes nothing (workspaces only grow)
            if st is None and len(self._graphs) < 4:
                self._graphs[key] = st = "warm"          # first sight of a signature: eager (sizes every workspace); captured at the second
            elif st == "warm":
                self._graphs[key] = st = self._capture(audio, video, labels, prompt)
                if st["generation"] != Workspace.generation:         # cannot happen (Workspace.get refuses to grow under capture); belt and braces
                    self._graphs[key] = st = "warm"
        if isinstance(st, dict):
            for dst, src in zip(st["inputs"], (audio, video, labels, prompt)):
                if dst is not None and src.data_ptr() != dst.data_ptr():
                    dst.copy_(src, non_blocking=True)
            self._replay(st)
        else:
            self._eager_step(audio, video, labels, prompt)
        return self._loss_buf.clone()

    @property
    def skipped_steps(self):
        return int(self.skipped.item())

    def _sync_step(self):
        """global_step := the optimizer steps actually taken (device-side count: skipped steps do not advance it).  One host sync; called
        where the host already waits for the device (log interval, checkpoint)."""
        self.global_step = int(self.state.view(torch.int32)[0].item())
        return self.global_step

    def _all_ranks_ok(self, ok):
        """Data-parallel runs: a batch is trained on only if EVERY rank could unpack its share -- a rank that skipped on its own would
        leave the others waiting in the step's all-reduces.  One 1-element MIN all-reduce per batch, only when distributed."""
        if not is_dist():
            return ok
        flag = torch.tensor([1.0 if ok else 0.0], device=self.model.llm_engine.lora_p.device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item() > 0.5)

    def _train_epoch(self, epoch):
        self.model.train()
        total, n, t0 = 0.0, 0, time.time()
        pending = []
        failures = unstable = worst = 0
        for i, batch in enumerate(self.train_dataloader):
            # the reference logs and skips a batch on ANY exception (:492-507).  Host-side preparation (everything before the first
            # collective) is where data errors surface; under DDP the skip is agreed between ranks first.
            unpacked, err = None, None
            try:
                unpacked = self._unpack(batch)
            except Exception as e:
                err = e
            if not self._all_ranks_ok(err is None):
                logging.error(f"Error in batch {i}: {err if err is not None else 'another rank failed to prepare its batch'}")
                failures += 1
                if failures > 5:
                    raise RuntimeError("more than 5 consecutive batches failed") from err
                continue
            try:
                loss = self.train_step(*unpacked)
            except Exception as e:                               # the reference logs and skips on ANY exception (:492-507)
                if is_dist():
                    raise                                        # mid-step: the other ranks are already inside the collectives
                logging.error(f"Error in batch {i}: {e}")
                failures += 1
                if failures > 5:
                    raise RuntimeError("more than 5 consecutive batches failed") from e
                continue
            failures = 0
            pending.append(loss)
            if (i + 1) % self.log_interval == 0 or i + 1 == len(self.train_dataloader):
                vals = torch.stack(pending).float().cpu()
                pending.clear()
                self._sync_step()
                fin = torch.isfinite(vals)
                ok = vals[fin]
                total += float(ok.sum()); n += int(ok.numel())
                # trainer :444-452: a NaN loss skips the batch; more than 5 unstable batches in a row stop the epoch.  The losses stay on
                # the device between log intervals (no per-step host sync), so the streak is examined here, over the interval just read.
                for f in fin.tolist():
                    unstable = 0 if f else unstable + 1
                    worst = max(worst, unstable)
                if not bool(fin.all()):
                    logging.warning(f"NaN loss detected in {int((~fin).sum())} batch(es) up to batch {i}; {self.skipped_steps} optimizer step(s) skipped so far")
                if worst > 5:
                    logging.error("Too many unstable batches. Stopping epoch.")
                    break
                logging.info(f"epoch {epoch} batch {i + 1}/{len(self.train_dataloader)} loss {float(vals[-1]):.4f} "
                             f"avg {total / max(1, n):.4f} lr {self.lr_at(self.global_step):.3e} {(time.time() - t0) / (i + 1):.3f}s/it")
        return total / max(1, n)

    @torch.no_grad()
    def _validate(self):
        if self.val_dataloader is None:
            return None
        self.model.eval()
        tot, cnt = 0.0, 0
        for batch in self.val_dataloader:
            audio, video, labels, prompt = self._unpack(batch)
            out = self.model(audio=audio, video=video, prompt=prompt, labels=labels, return_loss=True)
            bs = labels.shape[0]
            v = float(out["loss"])
            if math.isfinite(v):
                tot += v * bs; cnt += bs
        self.model.train()
        return tot / max(1, cnt)

    def _lora_keys(self):
        """peft-named LoRA keys in the order torch's optimizer indexes them (state-dict order: per layer q,k,v,o; A then B) == the order of
        the flat LoRA buffer."""
        eng = self.model.llm_engine
        from .engine import LORA_TARGETS
        return [f"llm.base_model.model.model.layers.{i}.self_attn.{nm}.{ab}.default.weight" for i in range(eng.cfg.layers) for nm in LORA_TARGETS
                for ab in ("lora_A", "lora_B")]

    def _save_checkpoint(self, epoch, name):
        """Same top-level keys as trainer:752-760.  model_state_dict holds the tensors this build owns (connectors + LoRA under peft's key
        names: decode.py:236-260 extracts the connectors by substring); the frozen encoders / LLM the reference also dumps into every
        checkpoint (13.5 GB for a 7B LLM) are not repeated -- they are the checkpoints the model was built from.  optimizer_state_dict /
        scheduler_state_dict follow torch.optim.AdamW / CosineAnnealingLR's own layout for the LoRA parameters, in optimizer index order."""
        if is_dist() and torch.distributed.get_rank() != 0:
            return
        os.makedirs(self.output_dir, exist_ok=True)
        self._sync_step()
        path = os.path.join(self.output_dir, name)
        eng = self.model.llm_engine
        state, n = {}, 0
        mv, vv = eng.lora_views(self.m), eng.lora_views(self.v)
        for i, key in enumerate(self._lora_keys()):
            _, _, _, _, _, li, _, mod, ab, _, _ = key.split(".")
            short = f"layers.{li}.{mod}.{ab}"
            state[i] = {"step": torch.tensor(float(self.global_step)), "exp_avg": mv[short].detach().cpu().clone(), "exp_avg_sq": vv[short].detach().cpu().clone()}
            n += 1
        group = {"lr": self.lr_at(self.global_step), "betas": (0.9, 0.95), "eps": 1e-8, "weight_decay": self.weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None, "initial_lr": self.learning_rate,
                 "params": list(range(n))}
        torch.save({"epoch": epoch, "model_state_dict": self.model.state_dict(),
                    "optimizer_state_dict": {"state": state, "param_groups": [group]},
                    "scheduler_state_dict": {"T_max": self.total_steps, "eta_min": 0.0, "base_lrs": [self.learning_rate], "last_epoch": self.global_step,
                                             "_step_count": self.global_step + 1, "_last_lr": [self.lr_at(self.global_step)]},
                    "train_losses": self.train_losses, "val_losses": self.val_losses, "best_val_loss": self.best_val_loss}, path)
        json.dump({"epoch": epoch, "global_step": self.global_step, "best_val_loss": self.best_val_loss},
                  open(path.replace(".pt", "_meta.json"), "w"))

    def load_checkpoint(self, path):
        """trainer:796-854.  Reads a checkpoint of this build AND one written by the reference trainer: connectors + LoRA from
        model_state_dict, Adam moments from torch's optimizer state (the only parameters with state are the ones that ever received a
        gradient -- the LoRA tensors, SURVEY.md fact 4 -- in state-dict order), the step count from the scheduler's last_epoch."""
        ck = torch.load(path, map_location="cpu", weights_only=True)
        self.model.load_state_dict(ck["model_state_dict"])
        eng = self.model.llm_engine
        o = ck.get("optimizer_state_dict") or {}
        step = None
        if "m" in o:                                             # round-1 layout of this build
            self.m.copy_(o["m"]); self.v.copy_(o["v"]); step = int(o.get("step", 0))
        elif "state" in o:
            ids = sorted(int(k) for k, st in o["state"].items() if isinstance(st, dict) and "exp_avg" in st)
            keys = self._lora_keys()
            if len(ids) != len(keys):
                raise ValueError(f"optimizer state has {len(ids)} tensors with moments, the model has {len(keys)} LoRA tensors")
            mv, vv = eng.lora_views(self.m), eng.lora_views(self.v)
            for idx, key in zip(ids, keys):
                st = o["state"][idx] if idx in o["state"] else o["state"][str(idx)]
                _, _, _, _, _, li, _, mod, ab, _, _ = key.split(".")
                short = f"layers.{li}.{mod}.{ab}"
                if tuple(st["exp_avg"].shape) != tuple(mv[short].shape):
                    raise ValueError(f"optimizer state {idx} has shape {tuple(st['exp_avg'].shape)}, {key} is {tuple(mv[short].shape)}")
                mv[short].copy_(st["exp_avg"]); vv[short].copy_(st["exp_avg_sq"])
                step = int(float(st["step"]))
        sched = ck.get("scheduler_state_dict") or {}
        if "last_epoch" in sched:
            step = int(sched["last_epoch"])
        elif "step" in sched:
            step = int(sched["step"])
        if step is not None:
            self.global_step = step
            self.state.view(torch.int32)[0] = step                 # the device-side step count drives lr / bias corrections / seeds
        self.train_losses, self.val_losses = list(ck.get("train_losses", [])), list(ck.get("val_losses", []))
        self.best_val_loss = ck.get("best_val_loss", float("inf"))
        return ck.get("epoch", 0)

    def train(self):
        os.makedirs(self.output_dir, exist_ok=True)
        t0 = time.time()
        log_csv = os.path.join(self.output_dir, "loss_log.csv")
        with open(log_csv, "w", newline="") as f:
            csv.writer(f).writerow(["epoch", "train_loss", "val_loss", "time_hours", "remaining_hours"])
        for epoch in range(self.max_epochs):
            tl = self._train_epoch(epoch)
            vl = self._validate()
            self.train_losses.append(tl)
            if vl is not None:
                self.val_losses.append(vl)
                if vl < self.best_val_loss:
                    self.best_val_loss = vl
                    self._save_checkpoint(epoch, "model_best.pt")
            if (epoch + 1) % self.save_every == 0:
                self._save_checkpoint(epoch, f"checkpoint_epoch_{epoch + 1}.pt")
            el = (time.time() - t0) / 3600
            with open(log_csv, "a", newline="") as f:
                csv.writer(f).writerow([epoch + 1, tl, vl, el, el / (epoch + 1) * (self.max_epochs - epoch - 1)])
        self._save_checkpoint(self.max_epochs - 1, "model_final.pt")
        return {"train_losses": self.train_losses, "val_losses": self.val_losses, "best_val_loss": self.best_val_loss,
                "training_hours": (time.time() - t0) / 3600}
