#!/usr/bin/env python3
"""Writes tests/golden/g6_preprocess.npz: inputs are regenerated from seeds, expected outputs come from the third-party
code the reference calls for its feature extraction (simple_dataset.py:156-186, :191-264): transformers'
WhisperFeatureExtractor + torch F.layer_norm, and CLIPImageProcessor (PIL backend -> Pillow BICUBIC).  Run HERE only
(needs transformers + Pillow); the fixture is data.  Also asserts oracle/preprocess.py against those outputs."""
import os
import sys
import zlib

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import preprocess as P  # noqa: E402


def wave_case(seed, n):
    rs = np.random.RandomState(seed)
    t = np.arange(n) / 16000.0
    x = 0.05 * rs.randn(n) + 0.2 * np.sin(2 * np.pi * (200 + 50 * seed) * t) * (t < 2.5) + 0.01 * np.sin(2 * np.pi * 3100 * t)
    return x.astype(np.float32)


def frame_case(seed, h, w):
    rs = np.random.RandomState(seed)
    base = rs.randint(0, 256, (h // 8 + 2, w // 8 + 2, 3)).astype(np.float32)
    img = np.kron(base, np.ones((8, 8, 1), dtype=np.float32))[:h, :w] + rs.randint(-20, 21, (h, w, 3))
    return np.clip(img, 0, 255).astype(np.uint8)


WAVES = [(1, 80000), (2, 480000), (3, 600000), (4, 7)]           # 5 s, exactly 30 s, truncated, nearly empty
FRAMES = [(1, 96, 96), (2, 96, 128), (3, 300, 260), (4, 224, 224), (5, 480, 640), (6, 225, 223)]


def main():
    import PIL
    import transformers
    from transformers import CLIPImageProcessor, WhisperFeatureExtractor
    fe, ip = WhisperFeatureExtractor(), CLIPImageProcessor()
    out = {"versions": np.array([transformers.__version__, PIL.__version__, torch.__version__])}
    out["mel_filters"] = np.asarray(fe.mel_filters, dtype=np.float64)
    assert np.abs(P.mel_filter_bank() - out["mel_filters"]).max() < 1e-15
    for seed, n in WAVES:
        w = wave_case(seed, n)
        # what the reference executes: with torch installed WhisperFeatureExtractor takes its float32 torch.stft path (:135-168)
        ref = fe(w, sampling_rate=16000, return_tensors="pt").input_features.squeeze(0)
        ref_n = F.layer_norm(ref, ref.shape)
        # HF's float64 numpy definition of the same features (:105-133), which the oracle restates; HF documents 1e-5 between the two
        x = np.zeros(P.N_SAMPLES, dtype=np.float32); x[: min(n, P.N_SAMPLES)] = w[: P.N_SAMPLES]
        ref64 = fe._np_extract_fbank_features(x[None], "cpu")[0]
        mine = P.log_mel(w)
        assert np.array_equal(mine, ref64), np.abs(mine - ref64).max()
        d = np.abs(mine - ref.numpy()).max()
        dn = np.abs(P.whole_tensor_layer_norm(mine) - ref_n.numpy()).max()
        print(f"wave seed {seed} n {n}: oracle == HF float64 path; vs HF torch float32 path max diff {d:.2e}, after layer_norm {dn:.2e}")
        assert d < 5e-5 and dn < 5e-4          # the spread between HF's own two implementations (float32 STFT vs float64)
        out[f"wave{seed}_logmel64_sub"] = ref64[:, ::25].copy()
        out[f"wave{seed}_n"] = np.array([seed, n])
        out[f"wave{seed}_logmel_sub"] = ref.numpy()[:, ::25].copy()            # [80,120]
        out[f"wave{seed}_norm_sub"] = ref_n.numpy()[:, ::25].copy()
        out[f"wave{seed}_norm_sum"] = np.array([ref_n.double().sum().item(), ref_n.double().abs().sum().item()])
    lut = np.zeros((3, 256), dtype=np.float32)
    for v in range(256):
        px = ip(images=np.full((224, 224, 3), v, dtype=np.uint8), return_tensors="np")["pixel_values"][0]
        lut[:, v] = px[:, 0, 0]
    out["normalize_lut"] = lut
    assert np.array_equal(P.clip_normalize_lut(), lut), np.abs(P.clip_normalize_lut() - lut).max()
    for seed, h, w in FRAMES:
        fr = frame_case(seed, h, w)
        ref = ip(images=fr, return_tensors="np")["pixel_values"][0]
        u8 = P.clip_resized_crop_u8(fr)
        mine = P.clip_pixel_values(fr)
        print(f"frame seed {seed} {h}x{w}: oracle vs CLIPImageProcessor max diff {np.abs(mine - ref).max():.2e}")
        assert np.array_equal(mine, ref)
        out[f"frame{seed}_hw"] = np.array([seed, h, w])
        out[f"frame{seed}_u8_crc"] = np.array([zlib.crc32(np.ascontiguousarray(u8).tobytes())], dtype=np.int64)
        out[f"frame{seed}_u8_sub"] = u8[::7, ::7].copy()
        out[f"frame{seed}_px_sub"] = ref[:, ::7, ::7].copy()
    path = os.path.join(ROOT, "tests", "golden", "g6_preprocess.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    # ---- g8: 128 mel bins (openai/whisper-large-v3's feature extractor = WhisperFeatureExtractor(feature_size=128); BASELINE config 5)
    fe128 = WhisperFeatureExtractor(feature_size=128)
    o8 = {"versions": out["versions"], "mel_filters_128": np.asarray(fe128.mel_filters, dtype=np.float64)}
    assert np.abs(P.mel_filter_bank(n_mels=128) - o8["mel_filters_128"]).max() < 1e-15
    for seed, n in WAVES[:3]:
        w = wave_case(seed, n)
        ref = fe128(w, sampling_rate=16000, return_tensors="pt").input_features.squeeze(0)
        x = np.zeros(P.N_SAMPLES, dtype=np.float32); x[: min(n, P.N_SAMPLES)] = w[: P.N_SAMPLES]
        ref64 = fe128._np_extract_fbank_features(x[None], "cpu")[0]
        mine = P.log_mel(w, 128)
        assert mine.shape == (128, 3000) and np.array_equal(mine, ref64), np.abs(mine - ref64).max()
        d = np.abs(mine - ref.numpy()).max()
        print(f"128 bins, wave seed {seed}: oracle == HF float64 path; vs HF torch float32 path {d:.2e}")
        assert d < 5e-5
        o8[f"wave{seed}_n"] = np.array([seed, n])
        o8[f"wave{seed}_logmel64_sub"] = ref64[:, ::25].copy()
        o8[f"wave{seed}_logmel_sub"] = ref.numpy()[:, ::25].copy()
    path8 = os.path.join(ROOT, "tests", "golden", "g8_logmel128.npz")
    np.savez_compressed(path8, **o8)
    print("wrote", path8, os.path.getsize(path8), "bytes")


if __name__ == "__main__":
    main()
