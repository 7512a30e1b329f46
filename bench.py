#!/usr/bin/env python3
"""bench.py -- AV-clip train-step throughput of the clip_whisper hot path on MI355X (BASELINE.json metric).

One "step" = one full reference training step (trainer/clip_whisper_trainer.py:433-490) on a batch of synthetic
LRS3-shaped 5 s clips resident in HBM: Whisper-small encoder + CLIP ViT-B/16 on 125 frames + connectors + fusion +
adaptive pool to 256 + Llama-2-7B LoRA(r16, q/k/v/o) forward + backward + grad clip + AdamW.  Weights are seeded
random tensors of the true shapes (no checkpoints offline).  All arithmetic runs in libavllm.so.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract in the task statement): whole-job samples/s, plus
  roofline     -- the dominant kernel (bf16 MFMA GEMM): algorithmic FLOPs / summed HIP-event time, live, vs 2.5 PF/s
  cpu_baseline -- the CPU oracle timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "audio-visual-llm_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_BF16_PEAK = 2.5e15       # dense, /opt/skills/guides/MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
MFMA_FP8_PEAK = 5.0e15        # dense, same table "Peak FP8 MFMA" (block-scaled forms)
FLOP_PER_CLIP = 11.64e12      # Whisper 0.344 + CLIP 4.391 + connectors 0.010 + LLM fwd 3.417 + bwd 3.451 + LoRA 0.026 TF


def synthetic_batch(cfg, B, frames, seed, device):
    g = torch.Generator(device=device).manual_seed(seed)
    audio = torch.randn(B, cfg.whisper.n_mels, 2 * cfg.whisper.n_ctx, generator=g, device=device)
    u8 = torch.randint(0, 256, (B, frames, 3, cfg.clip.image, cfg.clip.image), generator=g, device=device, dtype=torch.uint8)
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=device).view(1, 1, 3, 1, 1)
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=device).view(1, 1, 3, 1, 1)
    video = (u8.float() / 255.0 - mean) / std
    labels = torch.full((B, 256), 2, dtype=torch.long, device=device)
    for b in range(B):
        n = int(torch.randint(8, 41, (1,), generator=g, device=device))
        labels[b, 0] = 1
        labels[b, 1:1 + n] = torch.randint(3, cfg.llama.vocab, (n,), generator=g, device=device)
    return audio, video, labels, labels[:, :32].clone()


PMC_SUMMARY = "profiles/r01_pmc_hbm_summary_b16.txt"
TIMING_EVERY = 10


def pmc_traffic_bytes():
    """PMC counters need their own rocprofv3 passes (no --pmc next to timing), so the per-launch HBM traffic of the dominant
    kernel is read from the committed summary of that pass (tools/pmc_summary.py); None when the file is absent."""
    try:
        n, b = 0, 0.0
        for line in open(os.path.join(os.path.dirname(os.path.abspath(__file__)), PMC_SUMMARY)):
            f = line.split()
            if f and f[0] in ("gemm_bf16_wp_kernel", "gemm_bf16_h_kernel", "gemm_bf16_w_kernel"):       # the 256x256 kernels: launch-weighted mean
                n += int(f[1]); b += int(f[1]) * (float(f[2]) + float(f[3])) * 2 ** 20
        return round(b / n) if n else None
    except OSError:
        pass
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="clips per GPU")
    ap.add_argument("--frames", type=int, default=125)
    ap.add_argument("--max-seq-len", type=int, default=512)
    ap.add_argument("--precision", default="bf16", help="bf16 (BASELINE configs[1..3]) | fp32 | fp8 (BASELINE configs[4]: block-scaled fp8 on the frozen "
                    "projections of the forward pass; use with --whisper openai/whisper-large-v3 --clip openai/clip-vit-large-patch14 "
                    "--llm mistralai/Mistral-7B-v0.1 --frames 750)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step eagerly instead of replaying the captured hipGraph")
    ap.add_argument("--tiny", action="store_true", help="tiny model (plumbing check)")
    ap.add_argument("--whisper", default="openai/whisper-small", help="audio encoder by name (BASELINE config = whisper-small; config 5: openai/whisper-large-v3)")
    ap.add_argument("--clip", default="openai/clip-vit-base-patch16", help="visual encoder by name (config 5: openai/clip-vit-large-patch14)")
    ap.add_argument("--llm", default="meta-llama/Llama-2-7b-hf", help="LLM architecture by name (BASELINE config = Llama-2-7B); e.g. "
                    "meta-llama/Meta-Llama-3-8B or TinyLlama/TinyLlama-1.1B for the grouped-query family")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # AVLLM_BENCH_SHARED_GPU=1: rehearsal of the multi-rank plumbing on a ONE-GPU box (every rank on cuda:0, gloo instead of RCCL);
    # the numbers it prints are not a measurement
    rehearsal = os.environ.get("AVLLM_BENCH_SHARED_GPU") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(dev))

    from avllm import lib as L
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer

    if args.tiny:
        cfg = ModelCfg(WhisperCfg(128, 2, 2, 256), ClipCfg(128, 2, 2, 256, 48, 16), LlamaCfg(256, 2, 2, 512, 256), LoraCfg(16, 32.0))
        model = ClipWhisperModel(device=dev, max_seq_len=args.max_seq_len, config=cfg, precision=args.precision, seed=0, synthetic_weights=True)
        name = "tiny"
    else:
        model = ClipWhisperModel(args.llm, args.whisper, args.clip, device=dev,
                                 max_seq_len=args.max_seq_len, precision=args.precision, seed=0, synthetic_weights=True)
        base_enc = args.whisper == "openai/whisper-small" and args.clip == "openai/clip-vit-base-patch16"
        name = ("whisper-small+clip-vit-b16" if base_enc else args.whisper.split("/")[-1] + "+" + args.clip.split("/")[-1]) + "->" + \
               ("llama-2-7b" if "llama-2-7b" in args.llm.lower() else args.llm) + " lora r16"
    default_llm = "llama-2-7b" in args.llm.lower() and (args.tiny or base_enc)
    cfg = model.cfg
    model.train()
    # graph replay needs two warm-up calls (the first sizes the workspaces eagerly, the second captures): with fewer the step stays eager
    use_graph = not args.no_graph and args.warmup >= 2
    trainer = ClipWhisperTrainer(model, learning_rate=5e-5, weight_decay=0.01, grad_clip=0.5, total_steps=max(1000, args.steps + args.warmup),
                                 use_graph=use_graph)
    audio, video, labels, prompt = synthetic_batch(cfg, args.batch, args.frames, 1234 + rank, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    loss = None
    for _ in range(args.warmup):
        loss = trainer.train_step(audio, video, labels, prompt)
    si = trainer.static_inputs(audio, video, labels, prompt)
    if si is not None:                 # the synthetic batch lives in the captured step's own input buffers from here on (no per-step copy)
        audio, video, labels, prompt = si
    lib = L.load()
    timing = not args.no_kernel_timing
    barrier()
    if timing:
        L.check(lib.avllm_profile_begin(4000 * args.steps))
    # GEMM launches are bracketed with HIP events in every TIMING_EVERY-th timed step (the first one included).  Those steps are launched
    # eagerly (events cannot bracket single kernels of a graph replay) and a bracketed launch costs ~6 us of idle GPU (two barrier
    # packets): they stay inside the timed region and cost the reported throughput ~0.5 %
    sampled = 0
    t0 = time.perf_counter()
    for i in range(args.steps):
        if timing:
            on = i % TIMING_EVERY == 0
            L.check(lib.avllm_profile_enable(1 if on else 0))
            sampled += on
            loss = trainer.train_step(audio, video, labels, prompt, graph=not on)
            continue
        loss = trainer.train_step(audio, video, labels, prompt)
    barrier()
    dt = time.perf_counter() - t0
    prof = (ctypes.c_double * 4)()
    if timing:
        L.check(lib.avllm_profile_end(prof))
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    final_loss = float(loss)
    if rank == 0:
        clips = args.batch * world * args.steps
        value = clips / dt
        out = {
            "metric": "AV-clip train-step samples/sec (clip_whisper -> " + ("Llama-2-7B" if default_llm else args.llm) + " LoRA), whole job",
            "value": round(value, 4), "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": ("BASELINE configs[1]: " if default_llm and not args.tiny else "variant (not the BASELINE config): ") + f"{name}, synthetic LRS3-shaped {args.frames / 25:g} s clips ({args.frames} frames), "
                                   f"max_seq_len {args.max_seq_len}, train seq 256", "per_gpu_batch": args.batch,
                       "global_batch": args.batch * world, "parallelism": f"dp{world}", "samples_per_s_per_gpu": round(value / world, 4),
                       "final_loss": round(final_loss, 5), "launch": "hipGraph replay" if use_graph else "eager"},
        }
        frac_e2e = value / world * FLOP_PER_CLIP / MFMA_BF16_PEAK
        peak = MFMA_FP8_PEAK if args.precision == "fp8" else MFMA_BF16_PEAK
        if timing and prof[2] > 0:
            ach = prof[1] / (prof[0] * 1e-3) / 1e12           # TFLOP/s over the GEMM launches only
            kern = ("avllm_gemm launches: every dense projection (dominant: gemm_bf16_wp_kernel, persistent 256x256 tiles, 4 waves x 128x128)" if args.precision != "fp8" else
                    "every dense projection launch: forward frozen-weight projections on gemm_f8_wp_kernel (block-scaled fp8, persistent 256x256 tiles), backward dX and LoRA GEMMs on the bf16 kernels; priced against the fp8 peak")
            out["roofline"] = {"bound": "mfma", "kernel": kern, "achieved": round(ach, 2),
                               "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(ach / (peak / 1e12), 4),
                               "traffic": pmc_traffic_bytes(), "traffic_note": "HBM bytes per 256x256-tile GEMM launch, launch-weighted over the 256x256 kernels (FETCH_SIZE x2-corrected + WRITE_SIZE) from the separate rocprofv3 --pmc passes summarised in " + PMC_SUMMARY,
                               "launches_per_step": int(prof[2] / sampled), "timed_steps": f"{sampled} of {args.steps} (every {TIMING_EVERY}th; launched eagerly, the others replay the captured hipGraph)" if use_graph else f"{sampled} of {args.steps} (every {TIMING_EVERY}th)",
                               "avg_launch_us": round(1000 * prof[0] / prof[2], 2), "gemm_ms_per_step": round(prof[0] / sampled, 3),
                               "gemm_tflop_per_step": round(prof[1] / sampled / 1e12, 3),
                               "end_to_end_frac": round(frac_e2e, 4) if default_llm and not args.tiny else None}     # FLOP_PER_CLIP is the BASELINE config's
        else:
            out["roofline"] = {"bound": "mfma", "achieved": round(value / world * FLOP_PER_CLIP / 1e12, 2), "peak": MFMA_BF16_PEAK / 1e12,
                               "unit": "TFLOP/s", "frac": round(frac_e2e, 4), "traffic": None}
        if world == 1 and not args.no_cpu_baseline:
            from oracle import cpu_baseline
            out["cpu_baseline"] = cpu_baseline.run(frames=args.frames)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
