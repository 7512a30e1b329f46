"""GPU: the real trainer step under torch.distributed with 2 ranks on one card (gloo carries the CUDA tensors, since RCCL
refuses two ranks on one device): bucketed per-layer all-reduce driven by the C callback, token-count-weighted loss, replicated
clip+AdamW.  Result must equal ONE process stepping on the concatenated batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(dev):
    sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd")); sys.path.insert(0, ROOT)
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer
    from oracle import weights as Wt
    oc = Wt.tiny()
    W = Wt.all_weights(oc, 0, lora_b_std=0.05)
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device=dev, lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=512, config=cfg, weights=W,
                         precision="fp32").train()
    tr = ClipWhisperTrainer(m, learning_rate=1e-3, grad_clip=0.5, total_steps=10, max_epochs=1)
    audio, video, labels, prompt = Wt.synthetic_batch(oc, 4, 3, seed=77)
    labels[0, 12:] = oc.pad_token_id                      # uneven numbers of scored tokens across the two halves
    return m, tr, audio, video, labels, prompt


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    m, tr, audio, video, labels, prompt = _build("cuda:0")
    assert tr.reducer.enabled
    sl = slice(rank * 2, rank * 2 + 2)
    losses = []
    for _ in range(2):
        losses.append(float(tr.train_step(audio[sl].cuda(), video[sl].cuda(), labels[sl].cuda(), prompt[sl].cuda())))
    torch.cuda.synchronize()
    if rank == 0:
        q.put((losses, m.llm_engine.lora_p.cpu().numpy()))      # by value: a tensor travels as an fd the parent may fetch after this process exits
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_process(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    losses, params = q.get(timeout=600)
    params = torch.from_numpy(params)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    m, tr, audio, video, labels, prompt = _build("cuda:0")
    ref_losses = [float(tr.train_step(audio.cuda(), video.cuda(), labels.cuda(), prompt.cuda())) for _ in range(2)]
    ref = m.llm_engine.lora_p.cpu()
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-4, (losses, ref_losses)
    init = _build("cuda:0")[0].llm_engine.lora_p.cpu()
    rel = ((params - ref).norm() / (ref - init).norm()).item()
    assert rel < 2e-2, rel                                   # relative to the size of the 2-step update


def _rccl_worker(port, q):
    """world_size 1 over the `nccl` backend (= RCCL): the code bench.py / train.py run on the 8-GPU node -- init with device_id,
    the token-count all-reduce on the compute stream, async gradient all-reduces on the side stream (one per backward piece, eager step and
    hipGraph replay alike), work.wait() + stream join before
    the optimizer -- executes for real, on the one GPU this box has."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    m, tr, audio, video, labels, prompt = _build("cuda:0")
    assert not tr.reducer.enabled                       # world_size 1 is not "distributed" ...
    tr.reducer.enabled = True                           # ... so the reducer is forced on: every collective of the N > 1 path is issued
    tr.bwd_pieces = 2                                   # and the data-parallel graph layout: forward | 2 backward pieces | optimizer
    calls = []
    o1, o2 = tr.reducer.layer_done, tr.reducer.layers_done
    tr.reducer.layer_done = lambda layer: (calls.append(("layer", layer)), o1(layer))[1]
    tr.reducer.layers_done = lambda lo, hi: (calls.append(("piece", lo, hi)), o2(lo, hi))[1]
    losses = [float(tr.train_step(audio.cuda(), video.cuda(), labels.cuda(), prompt.cuda())) for _ in range(3)]
    torch.cuda.synchronize()
    q.put((losses, m.llm_engine.lora_p.cpu().numpy(), calls, dist.get_backend(), tr.use_graph))
    dist.barrier()
    dist.destroy_process_group()


def test_rccl_backend_world_size_1_matches_non_distributed(dev):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(31700 + os.getpid() % 2000, q))
    p.start()
    losses, params, calls, backend, graphed = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and backend == "nccl" and graphed
    # step 1 eager, steps 2 and 3 replay graphs: the SAME collectives either way -- one bucket per backward piece, last piece first -- so
    # ranks that disagree on eager vs replay (different input shapes) still pair up their all-reduces
    assert calls == 3 * [("piece", 1, 1), ("piece", 0, 0)], calls
    m, tr, audio, video, labels, prompt = _build("cuda:0")
    ref_losses = [float(tr.train_step(audio.cuda(), video.cuda(), labels.cuda(), prompt.cuda())) for _ in range(3)]
    assert max(abs(a - b) for a, b in zip(losses, ref_losses)) < 2e-4, (losses, ref_losses)
    ref = m.llm_engine.lora_p.cpu()
    init = _build("cuda:0")[0].llm_engine.lora_p.cpu()
    rel = ((torch.from_numpy(params) - ref).norm() / (ref - init).norm()).item()
    assert rel < 2e-2, rel


def test_bench_two_rank_launch_rehearsal(dev):
    """bench.py under the driver's own launch line (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`), rehearsed on
    one card (AVLLM_BENCH_SHARED_GPU=1: both ranks on cuda:0, gloo): rank/world plumbing, barriers, MAX over ranks and the ONE JSON line."""
    import json
    import subprocess
    env = dict(os.environ, AVLLM_BENCH_SHARED_GPU="1", MASTER_ADDR="127.0.0.1")
    port = 33700 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--tiny", "--batch", "2", "--frames", "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 4 and out["config"]["parallelism"] == "dp2" and out["value"] > 0
    assert "cpu_baseline" not in out                     # rank 0 at N = 1 only
