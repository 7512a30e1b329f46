"""Device-side feature extraction: raw 16 kHz samples and raw RGB uint8 frames in, the model's inputs out.

Mirrors what `AVSRDataset.__getitem__` does per sample on CPU workers (src/clip_whisper/data/simple_dataset.py:156-186,
:191-264) and what `collate_fn` does per batch (:317-460: stack audio, zero-pad video to the longest clip), on the GPU and per
batch: `WhisperLogMel` = `whisper_processor(audio, sampling_rate=16000).input_features` + `F.layer_norm(f, f.shape)`;
`ClipFrames` = `clip_processor(images=frame)["pixel_values"]`.  Decoding files (soundfile / cv2) stays with the caller."""
import ctypes as C

import numpy as np
import torch

from . import lib as L

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
N_SAMPLES, N_MELS, N_FRAMES = 480000, 80, 3000


class WhisperLogMel:
    def __init__(self, device="cuda:0", normalize=True, n_mels=N_MELS):
        """n_mels: 80 (Whisper tiny..large-v2) or 128 (large-v3's feature extractor, feature_size=128)."""
        self.device, self.normalize, self.n_mels = torch.device(device), normalize, int(n_mels)
        lib = L.load()
        self.table = torch.empty(lib.avllm_logmel_table_bytes(), dtype=torch.uint8, device=self.device)
        L.check(lib.avllm_logmel_table_init(L.ptr(self.table), self.n_mels))
        self._ws = None

    def pad_batch(self, waves):
        """list of 1-D float arrays/tensors (mono, 16 kHz, any length) -> one zero-padded [B, n] float32 device tensor."""
        waves = [torch.as_tensor(np.asarray(w, dtype=np.float32) if not torch.is_tensor(w) else w, dtype=torch.float32).reshape(-1)[:N_SAMPLES]
                 for w in waves]
        n = max(1, max(int(w.numel()) for w in waves))
        out = torch.zeros(len(waves), n, dtype=torch.float32, device=self.device)
        for i, w in enumerate(waves):
            out[i, : w.numel()] = w.to(self.device, non_blocking=True)
        return out

    def __call__(self, wave, out=None):
        """wave: [B, n] float32 (rows zero-padded) or a list of 1-D waveforms -> [B, n_mels, 3000] float32 on the device
        (written into `out` when given: a feeder filling a captured step's input buffer)."""
        if isinstance(wave, (list, tuple)):
            wave = self.pad_batch(wave)
        if wave.dim() == 1:
            wave = wave[None]
        if wave.dim() != 2:
            raise ValueError(f"waveform batch should have shape [batch_size, samples], but got {tuple(wave.shape)}")
        wave = wave.to(self.device, torch.float32).contiguous()
        B, n = wave.shape
        lib = L.load()
        need = lib.avllm_logmel_workspace_bytes(B, self.n_mels)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty(B, self.n_mels, N_FRAMES, dtype=torch.float32, device=self.device)
        elif tuple(out.shape) != (B, self.n_mels, N_FRAMES) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != wave.device:
            raise ValueError(f"out should be a contiguous float32 [{B}, {self.n_mels}, {N_FRAMES}] tensor on {wave.device}, but got {out.dtype} {tuple(out.shape)}")
        L.check(lib.avllm_logmel(L.ptr(self.table), L.ptr(wave), B, n, wave.stride(0), int(self.normalize), self.n_mels, L.ptr(out),
                                 L.ptr(self._ws), self._ws.numel(), L.stream_ptr()))
        return out


class ClipFrames:
    def __init__(self, device="cuda:0", image=224, mean=CLIP_MEAN, std=CLIP_STD, dtype=torch.float32):
        self.device, self.image, self.dtype = torch.device(device), image, dtype
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self._plans, self._ws = {}, None

    def _plan(self, H, W):
        key = (H, W)
        if key not in self._plans:
            lib = L.load()
            buf = torch.empty(lib.avllm_clip_preproc_plan_bytes(H, W, self.image), dtype=torch.uint8, device=self.device)
            L.check(lib.avllm_clip_preproc_plan_init(L.ptr(buf), H, W, self.image, self.mean, self.std))
            self._plans[key] = buf
        return self._plans[key]

    def __call__(self, frames, out=None):
        """frames: uint8 [N, H, W, 3] RGB (any H, W) -> pixel_values [N, 3, image, image] (written into `out` when given)."""
        frames = torch.as_tensor(frames)
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError(f"frames should be uint8 [frames, height, width, 3], but got {frames.dtype} {tuple(frames.shape)}")
        frames = frames.to(self.device).contiguous()
        N, H, W, _ = frames.shape
        lib = L.load()
        plan = self._plan(H, W)
        need = lib.avllm_clip_preproc_workspace_bytes(N, H, W, self.image)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out = torch.empty(N, 3, self.image, self.image, dtype=self.dtype, device=self.device)
        elif out.numel() != N * 3 * self.image * self.image or out.dtype != self.dtype or not out.is_contiguous() or out.device != frames.device:
            raise ValueError(f"out should be a contiguous {self.dtype} tensor of {N} x 3 x {self.image} x {self.image} elements on {frames.device}, "
                             f"but got {out.dtype} {tuple(out.shape)}")
        L.check(lib.avllm_clip_preproc(L.ptr(plan), L.ptr(frames), N, H, W, self.image, L.ptr(out), L.dt_of(out), L.ptr(self._ws),
                                       self._ws.numel(), L.stream_ptr()))
        return out


def device_collate(samples, logmel, clip_frames, max_video_length=300):
    """collate_fn (simple_dataset.py:317-460) for raw samples: `samples` = list of dicts with "wave" (1-D float, 16 kHz) and/or
    "frames" (uint8 [F,H,W,3]) + "labels"/"text".  -> (audio [B,80,3000] | None, video [B,Fmax,3,S,S] zero-padded | None)."""
    audio = video = None
    if all(s.get("wave") is not None for s in samples):
        audio = logmel([s["wave"] for s in samples])
    if all(s.get("frames") is not None for s in samples):
        clips = [clip_frames(torch.as_tensor(s["frames"])[:max_video_length]) for s in samples]
        Fmax = max(c.shape[0] for c in clips)
        video = torch.zeros(len(clips), Fmax, *clips[0].shape[1:], dtype=clips[0].dtype, device=clips[0].device)
        for i, c in enumerate(clips):
            video[i, : c.shape[0]] = c
    return audio, video
