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
  cpu_baseline -- the CPU oracle timed on this box's host cores: ONE complete B=1 train step at full depth (rank 0, N=1 only).
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


PMC_SUMMARY = "profiles/r03_pmc_hbm_summary_b16.txt"
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


def pctl(ms):
    """p10 / p50 / p90 of per-step GPU times (HIP events on the step's stream between consecutive steps)."""
    v = sorted(ms)
    q = lambda f: round(v[min(len(v) - 1, int(f * (len(v) - 1) + 0.5))], 3)
    return {"p10": q(0.1), "p50": q(0.5), "p90": q(0.9), "n": len(v)}


HBM_COPY_RATE = 6.29e12       # measured device-to-device copy rate of this part (MI355X_MICROARCH.md "HBM3E peak BW": float4 copy)
HBM_PEAK = 8.0e12             # spec peak, same line of the guide


def decode_bench(model, cfg, args, dev, B=8, new=48):
    """Greedy decode (clip_whisper_model.py:1337-1340): prefill on the 256 fused AV positions, then `new` single-token steps on the KV cache;
    once as scripts/clip_whisper/decode.py runs it (LLM without adapters) and once with the LoRA adapters attached (the trainer's own eval,
    `decode.py --load_lora`).  A token step streams every frozen weight once plus the KV cache rows written so far: its floor is those bytes /
    HBM rate, quoted against the 8.0 TB/s spec peak and against the 6.29 TB/s measured copy rate."""
    import contextlib
    eng = model.llm_engine
    g = torch.Generator(device=dev).manual_seed(7)
    x = (torch.randn(B, 256, cfg.llama.hidden, generator=g, device=dev) * 0.02).to(eng.dtype)
    ids = torch.randint(3, cfg.llama.vocab, (B,), generator=g, device=dev)
    wbytes = eng.frozen_weight_bytes()
    es = 4 if eng.dtype == torch.float32 else 2
    ctx = 260 + new // 2                                             # mean number of cache rows a timed step reads per sequence
    kvbytes = 2 * cfg.llama.layers * B * ctx * eng.dkv * es
    lora_bytes = 4 * cfg.llama.layers * (2 * 16 * cfg.llama.hidden) * es if eng.use_lora else 0      # the rank's own rows of the A and B images
    out = {"batch": B, "context": f"256 prefill + 4..{4 + new}", "weight_bytes_per_step": int(wbytes), "kv_bytes_per_step": int(kvbytes),
           "hbm_peak_tb_s": HBM_PEAK / 1e12, "hbm_copy_rate_tb_s": HBM_COPY_RATE / 1e12,
           "note": "hbm_frac_* = (frozen weight bytes + KV cache bytes [+ adapter bytes]) per token step / step time / rate"}
    for name, cm in (("no_adapters", eng.adapters_disabled), ("with_adapters", contextlib.nullcontext)):
        if name == "with_adapters" and not eng.use_lora:
            continue
        with cm():
            fused = eng.decode_is_fused(B)
            kc, vc = eng.alloc_cache(B, 256 + new + 8)
            eng.prefill(x, kc, vc)
            for w in range(4):
                eng.decode_step(ids, 256 + w, kc, vc)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for t in range(new):
                eng.decode_step(ids, 260 + t, kc, vc)          # ids stay fixed (random weights): the arithmetic per step does not depend on them
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / new
        byts = wbytes + kvbytes + (lora_bytes if name == "with_adapters" else 0)
        out[name] = {"ms_per_token_step": round(ms, 3), "tok_s": round(B * 1000 / ms, 1), "fused_token_step": fused,
                     "hbm_frac_of_peak": round(byts / (ms * 1e-3) / HBM_PEAK, 4), "hbm_frac_of_copy_rate": round(byts / (ms * 1e-3) / HBM_COPY_RATE, 4)}
    # the figures earlier rounds reported at top level (decode.py's path: no adapters)
    out["ms_per_token_step"] = out["no_adapters"]["ms_per_token_step"]
    out["tok_s"] = out["no_adapters"]["tok_s"]
    out["hbm_frac"] = out["no_adapters"]["hbm_frac_of_copy_rate"]
    return out


def host_inputs_leg(args, cfg, model, trainer, labels, prompt, dev, rank, world, barrier):
    """The step as the reference's DataLoader boundary feeds it (trainer/clip_whisper_trainer.py:604-723, data/simple_dataset.py:174-183,
    :235-256): each batch arrives in PINNED HOST memory as raw uint8 RGB frames [B,F,224,224,3] + 16 kHz float samples [B,80000]; a copy
    stream moves batch i+1 to the device (two staging sets) while step i computes; avllm_logmel / avllm_clip_preproc (bf16 out) write the
    captured step's input buffers; then the step.  Everything is inside the timed region."""
    from avllm.preprocess import ClipFrames, WhisperLogMel
    B, Fr, S = args.batch, args.frames, cfg.clip.image
    nsamp = int(16000 * Fr / 25)
    g = torch.Generator().manual_seed(99 + rank)
    host = []
    for _ in range(2):                                   # two distinct pinned batches, cycled (a DataLoader with pin_memory=True hands over such buffers)
        fr = torch.randint(0, 256, (B, Fr, S, S, 3), generator=g, dtype=torch.uint8).pin_memory()
        wv = (torch.randn(B, nsamp, generator=g) * 0.1).pin_memory()
        host.append((fr, wv))
    stage = [(torch.empty((B, Fr, S, S, 3), dtype=torch.uint8, device=dev), torch.empty((B, nsamp), dtype=torch.float32, device=dev)) for _ in range(2)]
    logmel = WhisperLogMel(device=dev, n_mels=cfg.whisper.n_mels)
    frames = ClipFrames(device=dev, image=S, dtype=torch.bfloat16)
    audio = torch.empty(B, cfg.whisper.n_mels, 3000, dtype=torch.float32, device=dev)
    video = torch.empty(B, Fr, 3, S, S, dtype=torch.bfloat16, device=dev)
    copy_stream = torch.cuda.Stream(device=dev)
    main = torch.cuda.current_stream()
    copied = [torch.cuda.Event() for _ in range(2)]
    consumed = [torch.cuda.Event() for _ in range(2)]

    def upload(i):
        k = i % 2
        with torch.cuda.stream(copy_stream):
            copy_stream.wait_event(consumed[k])          # the preprocessing that last read this staging set
            stage[k][0].copy_(host[k][0], non_blocking=True)
            stage[k][1].copy_(host[k][1], non_blocking=True)
            copied[k].record(copy_stream)

    def step(i, a, v):
        k = i % 2
        upload(i + 1)
        main.wait_event(copied[k])
        logmel(stage[k][1], out=a)
        frames(stage[k][0].view(B * Fr, S, S, 3), out=v)
        consumed[k].record(main)
        return trainer.train_step(a, v, labels, prompt)

    for k in range(2):
        consumed[k].record(main)
    upload(0)
    step(0, audio, video)                                # eager (sizes), capture, then the captured step's own buffers
    step(1, audio, video)
    si = trainer.static_inputs(audio, video, labels, prompt)
    if si is not None:
        audio, video, labels, prompt = si
    step(2, audio, video)
    n = args.steps
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    barrier()
    t0 = time.perf_counter()
    for i in range(n):
        marks[i].record()
        loss = step(3 + i, audio, video)
    marks[n].record()
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    copy_stream.synchronize()
    h2d = B * Fr * S * S * 3 + B * nsamp * 4
    return {"value": round(B * world * n / dt, 4), "unit": "samples/s", "ms_per_step": round(1000 * dt / n, 3),
            "step_ms": pctl([marks[i].elapsed_time(marks[i + 1]) for i in range(n)]), "steps": n,
            "h2d_bytes_per_step": h2d, "final_loss": round(float(loss), 5),
            "path": "pinned host uint8 frames [B,F,224,224,3] + f32 16 kHz samples -> H2D on a copy stream (2 staging sets, batch i+1 under step i) -> "
                    "avllm_logmel (f32) + avllm_clip_preproc (bf16 pixel_values) into the captured step's input buffers -> train step"}


def config5_leg(dev, steps=4, B=4, frames=750):
    """BASELINE configs[4] as a driver-timed variant: Whisper-large-v3 (128 mel bins, 32 x d1280) + CLIP ViT-L/14 (257 tokens per frame) ->
    Mistral-7B (grouped-query) with LoRA, 30 s utterances (750 frames), B clips per step, full train step.  ONE model built with
    precision="fp8" (bf16 weights + e4m3 images) runs the step in both arithmetics: fp8 = the frozen forward projections on the block-scaled
    fp8 matrix pipe, bf16 = the same launches with the engines' fp8 flag off.  Each: 2 warm-up steps (the second captures the hipGraph),
    `steps` replayed steps timed with a device sync on both sides, then one eager step with every GEMM launch bracketed by HIP events."""
    from avllm import lib as L
    from avllm.model import ClipWhisperModel
    from avllm.trainer import ClipWhisperTrainer
    lib = L.load()
    model = ClipWhisperModel("mistralai/Mistral-7B-v0.1", "openai/whisper-large-v3", "openai/clip-vit-large-patch14", device=dev, max_seq_len=512,
                             precision="fp8", seed=0, synthetic_weights=True).train()
    cfg = model.cfg
    batch = synthetic_batch(cfg, B, frames, 4321, dev)
    engines = [model.whisper_engine, model.clip_engine, model.llm_engine]
    lora0 = model.llm_engine.lora_p.clone()             # both arithmetics start from the same adapters (final_loss is then comparable)
    out = {"workload": f"BASELINE configs[4]: whisper-large-v3 + clip-vit-large-patch14 -> Mistral-7B lora r16, synthetic {frames / 25:g} s clips ({frames} frames), "
                       f"per_gpu_batch {B}, max_seq_len 512, train seq 256", "steps": steps, "warmup": 2}
    for mode in ("bf16", "fp8"):
        for e in engines:
            e.desc.fp8 = int(mode == "fp8")
        model.llm_engine.lora_p.copy_(lora0)
        model.llm_engine.pack_lora()
        tr = ClipWhisperTrainer(model, learning_rate=5e-5, weight_decay=0.01, grad_clip=0.5, total_steps=1000, use_graph=True)
        for _ in range(2):
            tr.train_step(*batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = tr.train_step(*batch)
        torch.cuda.synchronize()
        ms = 1000 * (time.perf_counter() - t0) / steps
        prof = (ctypes.c_double * 4)()
        L.check(lib.avllm_profile_begin(8000))
        L.check(lib.avllm_profile_enable(1))
        tr.train_step(*batch, graph=False)
        torch.cuda.synchronize()
        L.check(lib.avllm_profile_end(prof))
        peak = MFMA_FP8_PEAK if mode == "fp8" else MFMA_BF16_PEAK
        ach = prof[1] / (prof[0] * 1e-3) / 1e12 if prof[0] > 0 else 0.0
        out[mode] = {"ms_per_step": round(ms, 3), "samples_per_s": round(B * 1000 / ms, 4), "final_loss": round(float(loss), 5),
                     "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": peak / 1e12, "unit": "TFLOP/s", "frac": round(ach / (peak / 1e12), 4),
                                  "gemm_ms_per_step": round(prof[0], 3), "gemm_tflop_per_step": round(prof[1] / 1e12, 3), "launches_per_step": int(prof[2]),
                                  "note": "every avllm_gemm / avllm_gemm_f8 launch of one eager step, HIP events on the launching stream; fp8 is priced "
                                          "against the 5 PF/s fp8 peak although its backward dX and LoRA GEMMs run in bf16"}}
        del tr
    out["fp8_speedup_vs_bf16"] = round(out["bf16"]["ms_per_step"] / out["fp8"]["ms_per_step"], 4)
    del model
    torch.cuda.empty_cache()
    return out


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
    ap.add_argument("--cpu-baseline", default="full", choices=["sample", "full"], help="full (default, BASELINE.md §2): one complete B=1 train step of the oracle "
                    "at full depth, every frame, forward + backward (~20-35 s of host time); sample: a bounded slice of one B=1 step extrapolated by layer / "
                    "frame counts (~6 s; reads ~1.3x faster than the full step)")
    ap.add_argument("--no-host-inputs", action="store_true", help="skip the second leg (batches arriving in pinned host memory as raw uint8 frames + samples)")
    ap.add_argument("--no-decode", action="store_true", help="skip the greedy-decode leg")
    ap.add_argument("--no-config5", action="store_true", help="skip the variants.config5_fp8 leg (BASELINE configs[4] in fp8 and bf16, B=4 x 750 frames)")
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
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        if i:
            marks[i].record()
        if timing:
            on = i % TIMING_EVERY == 0
            L.check(lib.avllm_profile_enable(1 if on else 0))
            sampled += on
            loss = trainer.train_step(audio, video, labels, prompt, graph=not on)
            continue
        loss = trainer.train_step(audio, video, labels, prompt)
    marks[args.steps].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    prof = (ctypes.c_double * 4)()
    if timing:
        L.check(lib.avllm_profile_end(prof))
    tt = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt)
    final_loss = float(loss)
    host_leg = None
    if not args.no_host_inputs and not args.tiny and world == 1:      # secondary legs run at N = 1 only: the scaling runs time the contract's step and nothing else
        host_leg = host_inputs_leg(args, cfg, model, trainer, labels, prompt, dev, rank, world, barrier)
    decode_leg = None
    if rank == 0 and world == 1 and not args.no_decode and not args.tiny:
        decode_leg = decode_bench(model, cfg, args, dev)
    variants = None
    if rank == 0 and world == 1 and not args.no_config5 and not args.tiny and default_llm and args.precision == "bf16":
        # the main model's step graphs and activations go first: the variant is another 7B model with 30 s batches
        trainer._graphs = {}
        torch.cuda.empty_cache()
        variants = {"config5_fp8": config5_leg(dev)}
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
                       "final_loss": round(final_loss, 5), "launch": "hipGraph replay" if use_graph else "eager",
                       "step_ms": pctl(step_ms), "inputs": "resident in HBM: fp32 log-mel [B,80,3000] + fp32 pixel_values [B,frames,3,224,224] (the reference model's arguments)"},
        }
        if host_leg is not None:
            out["config"]["host_inputs"] = host_leg
        if decode_leg is not None:
            out["decode"] = decode_leg
        if variants is not None:
            out["variants"] = variants
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
            out["cpu_baseline"] = cpu_baseline.run_full(frames=args.frames) if args.cpu_baseline == "full" else cpu_baseline.run(frames=args.frames)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
