"""GPU: the trainer's step machinery under the conditions real batches create:
* input signatures of different sizes interleaved with captured hipGraphs (a workspace that regrows must not leave a graph pointing
  into freed memory),
* a non-finite batch (the update is skipped AND neither the LR schedule nor Adam's bias-correction count advances, as the reference's
  `continue` before optimizer.step()/scheduler.step() does: trainer/clip_whisper_trainer.py:444-464),
* two data-parallel ranks whose batches have DIFFERENT shapes, so one rank replays a graph while the other is still eager."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import weights as Wt

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(precision="fp32", dropout=0.0):
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    oc = Wt.tiny()
    W = Wt.all_weights(oc, 0, lora_b_std=0.05)
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device="cuda:0", lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=dropout, max_seq_len=512, config=cfg,
                         weights=W, precision=precision).train()
    return oc, m


def _long(batch, oc):
    """Long transcripts: a NaN enters through the audio frames, i.e. behind the pooled prompt rows; with causal attention only label
    positions past them can see it (a short transcript scores none of them and trains normally -- as the reference would)."""
    audio, video, labels, prompt = batch
    labels = labels.clone()
    labels[:, 1:200] = torch.randint(3, oc.llama.vocab, (labels.shape[0], 199), generator=torch.Generator().manual_seed(5))
    return audio, video, labels, prompt


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def test_replay_after_a_larger_signature_regrew_the_workspaces(dev):
    """Small batch captured -> a larger batch (more clips, more frames) regrows every engine workspace -> the small signature again.
    The captured graphs of the small signature must be dropped, not replayed into freed memory: the run has to equal the eager run."""
    from avllm.engine import Workspace
    from avllm.trainer import ClipWhisperTrainer
    oc = Wt.tiny()
    small = [Wt.synthetic_batch(oc, 1, 2, seed=s) for s in (11, 12, 13, 14, 15, 16)]
    big = [Wt.synthetic_batch(oc, 3, 5, seed=s) for s in (21, 22, 23)]
    order = [("s", 0), ("s", 1), ("s", 2), ("b", 0), ("s", 3), ("b", 1), ("s", 4), ("b", 2), ("s", 5)]
    runs = {}
    for graph in (False, True):
        _, m = _model()
        tr = ClipWhisperTrainer(m, learning_rate=1e-3, total_steps=16, max_epochs=1, grad_clip=0.5, use_graph=graph)
        losses, events = [], []
        for kind, i in order:
            b = small[i] if kind == "s" else big[i]
            gen = Workspace.generation
            junk = torch.empty(64 << 20, dtype=torch.uint8, device=dev).fill_(0xFF)      # whatever was freed gets overwritten with NaN patterns
            del junk
            losses.append(float(tr.train_step(*[t.to(dev) for t in b])))
            events.append((kind, Workspace.generation != gen, {k[0][0][0]: type(v).__name__ for k, v in tr._graphs.items()}))
        runs[graph] = (losses, m.llm_engine.lora_p.cpu().clone(), events)
    (l0, p0, _), (l1, p1, ev) = runs[False], runs[True]
    assert ev[2][2][1] == "dict"                                 # the small signature was captured at its second visit and replayed at the third
    assert ev[3][1]                                              # the first big batch regrew a workspace ...
    assert all(np.isfinite(l1))
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 1e-5, (l0, l1)
    assert _rel(p1, p0) < 1e-4
    assert any(v == "dict" for v in ev[-1][2].values())           # ... and graphs are in use again at the end


def test_nonfinite_batch_skips_update_and_does_not_advance_schedule(dev):
    from avllm.trainer import ClipWhisperTrainer
    oc = Wt.tiny()
    good = [_long(Wt.synthetic_batch(oc, 2, 3, seed=s), oc) for s in (1, 2, 3, 4)]
    bad = [t.clone() if t is not None else None for t in good[1]]
    bad[0][0, 0, 0] = float("nan")                               # one NaN in the mel of clip 0
    for graph in (False, True):
        _, m = _model()
        tr = ClipWhisperTrainer(m, learning_rate=1e-3, total_steps=10, max_epochs=1, warmup_steps=2, grad_clip=0.5, use_graph=graph)
        _, mref = _model()
        ref = ClipWhisperTrainer(mref, learning_rate=1e-3, total_steps=10, max_epochs=1, warmup_steps=2, grad_clip=0.5, use_graph=graph)
        seq = [good[0], good[1], bad, good[2], bad, good[3]]
        for b in seq:
            loss = tr.train_step(*[t.to(dev) for t in b])
            if b is bad:
                assert not bool(torch.isfinite(loss))
        for b in (good[0], good[1], good[2], good[3]):            # the same run without the bad batches
            ref.train_step(*[t.to(dev) for t in b])
        st = tr.state.cpu().numpy()
        assert int(st.view(np.uint32)[0]) == 4 and tr.skipped_steps == 2 and tr._sync_step() == 4
        assert abs(float(st.view(np.float32)[2]) - tr.lr_at(3)) < 1e-9                  # lr of the 4th optimizer step, not of the 6th call
        assert _rel(m.llm_engine.lora_p.cpu(), mref.llm_engine.lora_p.cpu()) < 1e-5     # parameters == the run that never saw the bad batches
        assert _rel(tr.m.cpu(), ref.m.cpu()) < 1e-5 and _rel(tr.v.cpu(), ref.v.cpu()) < 1e-5


def test_train_epoch_stops_after_more_than_five_unstable_batches(dev, caplog):
    from avllm.trainer import ClipWhisperTrainer
    oc = Wt.tiny()
    good = _long(Wt.synthetic_batch(oc, 2, 3, seed=1), oc)
    bad = [t.clone() for t in good]
    bad[0][:] = float("nan")
    mk = lambda b: {"audio": b[0].to(dev), "video": b[1].to(dev), "labels": b[2].to(dev), "prompt": b[3].to(dev)}
    _, m = _model()
    loader = [mk(good)] + [mk(bad)] * 7 + [mk(good)] * 4
    tr = ClipWhisperTrainer(m, train_dataloader=loader, learning_rate=1e-3, max_epochs=1, log_interval=4)
    import logging
    with caplog.at_level(logging.ERROR):
        tr._train_epoch(0)
    assert "Too many unstable batches. Stopping epoch." in caplog.text
    assert tr.global_step == 1                                   # one real optimizer step; the epoch ended at the log interval that saw the streak


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd")); sys.path.insert(0, ROOT)
    from avllm.trainer import ClipWhisperTrainer
    oc, m = _model()
    tr = ClipWhisperTrainer(m, learning_rate=1e-3, grad_clip=0.5, total_steps=10, max_epochs=1, use_graph=True, bwd_pieces=2)
    assert tr.reducer.enabled and tr.use_graph
    modes = []
    for step, frames in enumerate(_FRAMES[rank]):
        a, v, lab, pr = Wt.synthetic_batch(oc, 2, frames, seed=100 + 10 * step + rank)
        tr.train_step(a.cuda(), v.cuda(), lab.cuda(), pr.cuda())
        modes.append(type(tr._graphs.get(tr._signature(a, v, lab, pr))).__name__)
    torch.cuda.synchronize()
    q.put((rank, modes, m.llm_engine.lora_p.cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


# rank 0 sees one signature throughout (replays from its third step on); rank 1 keeps meeting new ones (eager while rank 0 replays)
_FRAMES = {0: [3, 3, 3, 3, 3, 3], 1: [3, 4, 3, 5, 4, 3]}


def test_ranks_with_different_shapes_mix_eager_and_replay(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35700 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(2):
        rank, modes, params = q.get(timeout=600)
        got[rank] = (modes, torch.from_numpy(params))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0][0][2:] == ["dict"] * 4, got[0][0]               # rank 0 replays ...
    assert "str" in got[1][0][2:], got[1][0]                      # ... while rank 1 is eager at some of the same steps
    assert torch.equal(got[0][1], got[1][1])                      # replicated update: bit-identical parameters on both ranks
