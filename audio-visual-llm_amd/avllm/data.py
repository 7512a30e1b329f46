"""LRS3-style manifest dataset that hands RAW samples to the GPU (mirror of `AVSRDataset`, src/clip_whisper/data/simple_dataset.py).

Same manifest / label files and the same per-sample rules as the reference (`_load_manifest` :76-116: first line = root, then
`id \\t video_path \\t audio_path \\t num_frames \\t num_samples`; labels one per line :118-124; stereo -> mono by mean and
int16-range -> /32768 :160-170; at most `max_video_length` frames :201; labels tokenised to 256 ids with pad = eos :289-303; up
to 10 neighbouring samples are tried when files are missing :131-150; zero features as the last resort :279-285), but
`__getitem__` stops after DECODING: the waveform and the uint8 RGB frames travel to the device, where
`avllm.preprocess.device_collate` produces the log-mel / pixel-value tensors the reference's CPU workers would have made.

Decoders available without extra packages: audio `.wav` (stdlib `wave`, PCM 8/16/32) and `.npy`; video `.npy` / `.npz` holding
uint8 `[F,H,W,3]` RGB.  Container formats (`.mp4`, `.flac`...) go through `soundfile` / `cv2` only when those are importable."""
import logging
import os
import wave

import numpy as np
import torch


def read_audio(path):
    """-> (float array [n] or [n, channels], sample_rate); the same values soundfile.read would return for PCM wav."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        return np.load(path), 16000
    if ext == ".wav":
        with wave.open(path, "rb") as w:
            n, ch, sw, sr = w.getnframes(), w.getnchannels(), w.getsampwidth(), w.getframerate()
            raw = w.readframes(n)
        if sw == 2:
            x = np.frombuffer(raw, dtype="<i2").astype(np.float64) / 32768.0
        elif sw == 4:
            x = np.frombuffer(raw, dtype="<i4").astype(np.float64) / 2147483648.0
        elif sw == 1:
            x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
        else:
            raise ValueError(f"{path}: unsupported sample width {sw}")
        return (x.reshape(-1, ch) if ch > 1 else x), sr
    try:
        import soundfile as sf
    except ImportError as e:
        raise ValueError(f"{path}: only .wav/.npy audio can be decoded without the soundfile package") from e
    return sf.read(path)


def read_frames(path, max_frames):
    """-> uint8 [F,H,W,3] RGB, at most max_frames (simple_dataset.py:193-210)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        fr = np.array(np.load(path, mmap_mode="r")[:max_frames])       # copy the slice out of the read-only map
    elif ext == ".npz":
        z = np.load(path)
        fr = z[z.files[0]][:max_frames]
    else:
        try:
            import cv2
        except ImportError as e:
            raise ValueError(f"{path}: only .npy/.npz frame stacks can be read without the cv2 package") from e
        cap, out = cv2.VideoCapture(path), []
        while len(out) < max_frames:
            ok, f = cap.read()
            if not ok:
                break
            out.append(cv2.cvtColor(f, cv2.COLOR_BGR2RGB))
        cap.release()
        fr = np.stack(out) if out else np.zeros((0, 1, 1, 3), np.uint8)
    fr = np.asarray(fr)
    if fr.ndim != 4 or fr.shape[-1] != 3 or fr.dtype != np.uint8:
        raise ValueError(f"{path}: expected uint8 frames [F,H,W,3], got {fr.dtype} {fr.shape}")
    return np.ascontiguousarray(fr)


class AVSRDataset(torch.utils.data.Dataset):
    def __init__(self, manifest_path, label_path, root_dir, tokenizer, max_audio_length=30, max_video_length=300, sampling_rate=16000,
                 split="train", normalize=True, modality="both", label_length=256):
        self.root_dir, self.tokenizer, self.modality = root_dir, tokenizer, modality
        self.max_audio_length, self.max_video_length, self.sampling_rate = max_audio_length, max_video_length, sampling_rate
        self.split, self.normalize, self.label_length = split, normalize, label_length
        self.names, self.sizes = self._load_manifest(manifest_path)
        with open(label_path) as f:
            self.labels = [line.strip() for line in f]
        if len(self.names) != len(self.labels):
            logging.warning(f"Mismatch between manifest ({len(self.names)}) and labels ({len(self.labels)})")

    @staticmethod
    def _load_manifest(path):
        names, sizes = [], []
        with open(path) as f:
            f.readline()                                     # root line (the reference reads and ignores it too)
            for line in f:
                it = line.strip().split("\t")
                if len(it) < 5:
                    logging.warning(f"Skipping invalid line: {line.strip()}")
                    continue
                try:
                    int(it[3]); n = int(it[4])
                except ValueError:
                    logging.warning(f"Invalid frame/sample count in line: {line.strip()}")
                    continue
                names.append((it[1], it[2], it[0]))
                sizes.append(n)
        return names, sizes

    def __len__(self):
        return len(self.names)

    def _decode(self, idx):
        vp, ap, _ = self.names[idx]
        vp, ap = os.path.join(self.root_dir, vp), os.path.join(self.root_dir, ap)
        wave_, frames = None, None
        if self.modality in ("audio", "both") and os.path.exists(ap):
            try:
                a, _sr = read_audio(ap)
                a = np.asarray(a)
                if a.ndim > 1:
                    a = a.mean(axis=1)
                if self.normalize:
                    a = a.astype(np.float32) / 32768.0 if np.abs(a).max(initial=0.0) > 1.0 else a.astype(np.float32)
                wave_ = np.ascontiguousarray(a[: self.max_audio_length * self.sampling_rate], dtype=np.float32)
            except Exception as e:                            # noqa: BLE001  (the reference logs and moves on, :184-186)
                logging.error(f"Error loading audio {ap}: {e}")
        if self.modality in ("video", "both") and os.path.exists(vp):
            try:
                fr = read_frames(vp, self.max_video_length)
                frames = fr if len(fr) else None
            except Exception as e:                            # noqa: BLE001
                logging.error(f"Error loading video {vp}: {e}")
        return wave_, frames

    def __getitem__(self, idx):
        cur, wave_, frames = idx, None, None
        for attempt in range(10):
            cur = (idx + attempt) % len(self)
            wave_, frames = self._decode(cur)
            if {"audio": wave_ is not None, "video": frames is not None, "both": wave_ is not None and frames is not None}[self.modality]:
                break
        else:
            logging.warning("Failed to find valid sample after 10 attempts, returning dummy features")
            wave_ = np.zeros(1, np.float32) if self.modality in ("audio", "both") else None
            frames = np.zeros((10, 224, 224, 3), np.uint8) if self.modality in ("video", "both") else None
        text = self.labels[cur]
        ids = self.tokenizer([text], padding="max_length", truncation=True, max_length=self.label_length).input_ids[0]
        return {"wave": wave_, "frames": frames, "text": text, "labels": torch.as_tensor(ids, dtype=torch.long)}

    @staticmethod
    def collate_fn(batch):
        """Raw samples stay host lists (ragged); labels are stacked.  `ClipWhisperTrainer._unpack` finishes the job on the device."""
        return {"raw": [{"wave": b["wave"], "frames": b["frames"]} for b in batch], "texts": [b["text"] for b in batch],
                "labels": torch.stack([b["labels"] for b in batch])}


def create_dataloaders(manifest_path, label_path, root_dir, tokenizer, batch_size=4, num_workers=0, modality="both",
                       max_audio_length=30, max_video_length=300, sampler=None, shuffle=True):
    """simple_dataset.create_dataloaders :470-660 for the raw-sample dataset: train loader + val loader when the `val` files exist."""
    def make(mp, lp, sh, smp):
        ds = AVSRDataset(mp, lp, root_dir, tokenizer, max_audio_length, max_video_length, modality=modality)
        if len(ds) == 0:
            raise ValueError("Training dataset is empty after filtering")
        return torch.utils.data.DataLoader(ds, batch_size=batch_size, shuffle=sh and smp is None, sampler=smp, num_workers=num_workers,
                                           collate_fn=AVSRDataset.collate_fn)
    for p in (manifest_path, label_path, root_dir):
        if not os.path.exists(p):
            raise FileNotFoundError(f"not found: {p}")
    train = make(manifest_path, label_path, shuffle, sampler)
    vm, vl = manifest_path.replace("train", "val"), label_path.replace("train", "val")
    val = make(vm, vl, False, None) if vm != manifest_path and os.path.exists(vm) and os.path.exists(vl) else None
    return train, val
