import sys, os, torch, time
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "audio-visual-llm_amd")]
from avllm.arch import ClipCfg
from avllm.engine import ClipEngine
from oracle import weights as Wt
c = Wt.ClipCfg()           # ViT-B/16 defaults
sd = {k: v for k, v in Wt.clip_weights(c, 0).items()}
N = 2000
frames = torch.randn(N, 3, 224, 224, device="cuda", dtype=torch.bfloat16)
for chunk in (0, 1000, 500, 250, 125):
    eng = ClipEngine(sd, ClipCfg(**vars(c)), torch.bfloat16, "cuda", chunk_frames=chunk)
    for _ in range(2): eng.forward(frames)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3): eng.forward(frames)
    e1.record(); torch.cuda.synchronize()
    print(f"chunk_frames={chunk:5d}: {e0.elapsed_time(e1)/3:8.2f} ms per CLIP forward of {N} frames", flush=True)
    del eng; torch.cuda.empty_cache()
