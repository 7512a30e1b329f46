#!/usr/bin/env python3
"""Times the device-side feature extraction on bench-shaped inputs: B clips of 5 s audio and 125 frames each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd"))
import torch
from avllm.preprocess import ClipFrames, WhisperLogMel


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev, B, F = "cuda:0", int(os.environ.get("B", "16")), 125
    wave = torch.randn(B, 80000, device=dev) * 0.1
    lm = WhisperLogMel(dev)
    ms = timed(lambda: lm(wave))
    print(f"log-mel + layer norm  B={B} (5 s clips -> [B,80,3000])   {ms:8.3f} ms   {B * 0.965 / ms:7.1f} GFLOP/ms f64-DFT equivalent")
    for H, W in ((96, 96), (224, 224), (480, 640)):
        fr = torch.randint(0, 256, (B * F, H, W, 3), dtype=torch.uint8, device=dev)
        for dt in (torch.float32, torch.bfloat16):
            cf = ClipFrames(dev, dtype=dt)
            ms = timed(lambda: cf(fr))
            byt = fr.numel() + B * F * 3 * 224 * 224 * (4 if dt == torch.float32 else 2) + 2 * B * F * H * 224 * 3
            print(f"clip frames {H}x{W} -> 224, {B * F} frames, {str(dt)[6:]:8s} {ms:8.3f} ms   {byt / ms / 1e9:6.2f} TB/s (in + tmp w/r + out)")
        del fr


if __name__ == "__main__":
    main()
