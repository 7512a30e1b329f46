"""Data-parallel gradient exchange for the LoRA buffer: one process per GPU, torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" for the CPU tests).

The reference is single-process (SURVEY.md §2 census: no collective anywhere), so the semantics to preserve are
"one process at batch N*B": token-mean cross entropy over the WHOLE global batch (HF:loss/loss_utils.py:32-46).
Ranks therefore back-propagate sum-CE divided by the ALL-REDUCED count of scored tokens and SUM-all-reduce the
gradients; plain per-rank-mean averaging would differ whenever ranks hold different numbers of scored labels.

The only payload is the flat fp32 LoRA gradient (16.8 M floats for Llama-2-7B r=16).  It is reduced in per-layer
buckets (524,288 floats = 2 MiB each) as the backward pass finishes each decoder layer (31 -> 0), on a side
stream, so the xGMI transfers overlap with the remaining layers' dX GEMMs.  xGMI is point-to-point and a ring
all-reduce is per-link bound (~153 GB/s): 2 MiB buckets cost ~25 us each, far below one layer's backward time.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class LoraGradReducer:
    def __init__(self, flat_grad: torch.Tensor, per_layer: int, layers: int, group=None, bucket_layers: int = 1):
        self.g, self.per_layer, self.layers, self.group = flat_grad, per_layer, layers, group
        self.bucket_layers = max(1, bucket_layers)
        self.cuda = flat_grad.is_cuda
        self.stream = torch.cuda.Stream() if self.cuda else None
        self.work = []
        self.enabled = is_dist()

    def reduce_counts(self, acc: torch.Tensor):
        """acc = [loss_sum, count] on every rank -> global sums (tiny, on the compute stream)."""
        if self.enabled:
            dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
        return acc

    def layer_done(self, layer: int):
        """Called right after layer `layer`'s backward kernels were enqueued (avllm_llama_lora_bwd callback)."""
        if not self.enabled:
            return
        if layer % self.bucket_layers != 0:
            return
        lo = layer * self.per_layer
        hi = min(self.layers, layer + self.bucket_layers) * self.per_layer
        bucket = self.g[lo:hi]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self.work.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.work.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def layers_done(self, lo: int, hi: int):
        """Graph-replayed backward pieces: all-reduce the contiguous gradient slice of decoder layers lo..hi (inclusive) on the side stream,
        ordered after everything enqueued on the compute stream so far (the piece that produced it)."""
        if not self.enabled:
            return
        bucket = self.g[lo * self.per_layer:(hi + 1) * self.per_layer]
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self.work.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        else:
            self.work.append(dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Join the side stream before the optimizer reads the gradients."""
        for w in self.work:
            w.wait()
        self.work.clear()
        if self.enabled and self.cuda:
            torch.cuda.current_stream().wait_stream(self.stream)
