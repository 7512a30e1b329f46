"""avllm.data.AVSRDataset: the reference's manifest / label formats and per-sample rules (simple_dataset.py:76-124, :156-170,
:193-210, :289-303) with decoding only -- features are made on the device (tests/test_data_gpu.py)."""
import os
import wave

import numpy as np
import torch

from avllm.data import AVSRDataset, create_dataloaders, read_audio
from avllm.tokenizer import ByteTokenizer


def write_wav(path, x, sr=16000, channels=1):
    with wave.open(str(path), "wb") as w:
        w.setnchannels(channels); w.setsampwidth(2); w.setframerate(sr)
        w.writeframes((np.clip(x, -1, 1) * 32767).astype("<i2").tobytes())


def make_set(root, n=5, missing=()):
    rs = np.random.RandomState(0)
    lines, labels = [str(root)], []
    for i in range(n):
        nfr, ns = 3 + i, 4000 + 1000 * i
        if i not in missing:
            if i % 2:
                write_wav(root / f"a{i}.wav", np.stack([rs.randn(ns) * 0.1, rs.randn(ns) * 0.1], 1), channels=2)      # stereo
            else:
                write_wav(root / f"a{i}.wav", rs.randn(ns) * 0.1)
            np.save(root / f"v{i}.npy", rs.randint(0, 256, (nfr, 24, 32, 3), dtype=np.uint8))
        lines.append(f"id{i}\tv{i}.npy\ta{i}.wav\t{nfr}\t{ns}")
        labels.append(f"hello world {i}")
    lines.insert(3, "broken line")
    (root / "train.tsv").write_text("\n".join(lines) + "\n")
    (root / "train.wrd").write_text("\n".join(labels) + "\n")
    return root / "train.tsv", root / "train.wrd"


def test_manifest_decode_and_collate(tmp_path):
    mp, lp = make_set(tmp_path)
    tok = ByteTokenizer(512)
    ds = AVSRDataset(str(mp), str(lp), str(tmp_path), tok, max_video_length=4)
    assert len(ds) == 5                                              # the malformed line is skipped
    s0, s1 = ds[0], ds[1]
    assert s0["wave"].dtype == np.float32 and s0["wave"].shape == (4000,) and s0["frames"].shape == (3, 24, 32, 3)
    assert s1["wave"].shape == (5000,) and abs(s1["wave"]).max() <= 1.0       # stereo -> mono by mean
    assert ds[4]["frames"].shape[0] == 4                              # max_video_length cap
    a, sr = read_audio(str(tmp_path / "a0.wav"))
    assert sr == 16000 and np.allclose(a, s0["wave"], atol=1e-7)
    assert s0["labels"].shape == (256,) and s0["text"] == "hello world 0"
    b = AVSRDataset.collate_fn([s0, s1])
    assert b["labels"].shape == (2, 256) and len(b["raw"]) == 2 and b["texts"] == ["hello world 0", "hello world 1"]


def test_missing_files_fall_through_to_neighbours(tmp_path):
    mp, lp = make_set(tmp_path, missing=(1,))
    ds = AVSRDataset(str(mp), str(lp), str(tmp_path), ByteTokenizer(512))
    s = ds[1]                                                         # sample 1 has no files: sample 2 is returned, with ITS label
    assert s["text"] == "hello world 2" and s["wave"].shape == (6000,)
    da = AVSRDataset(str(mp), str(lp), str(tmp_path), ByteTokenizer(512), modality="audio")
    assert da[0]["frames"] is None and da[0]["wave"] is not None
    tr, val = create_dataloaders(str(mp), str(lp), str(tmp_path), ByteTokenizer(512), batch_size=2, shuffle=False)
    assert val is None and len(tr) == 3
    assert next(iter(tr))["labels"].shape == (2, 256)
