"""Oracle for the input features (oracle/preprocess.py) against the fixture made from transformers' WhisperFeatureExtractor,
torch's F.layer_norm and CLIPImageProcessor/Pillow (tests/golden/g6_preprocess.npz, oracle/make_golden_preproc.py).
Integer work (the Pillow resize) must be bit-exact; the log-mel must equal HF's float64 definition exactly and sit within the
spread of HF's own float32 torch path (the one the reference executes when torch is installed)."""
import zlib

import numpy as np
import pytest

from oracle import preprocess as P
from oracle.make_golden_preproc import FRAMES, WAVES, frame_case, wave_case


@pytest.fixture(scope="module")
def g6(golden_dir):
    return np.load(f"{golden_dir}/g6_preprocess.npz")


def test_mel_filters_and_lut(g6):
    assert np.abs(P.mel_filter_bank() - g6["mel_filters"]).max() < 1e-15
    assert np.array_equal(P.clip_normalize_lut(), g6["normalize_lut"])


@pytest.mark.parametrize("seed,n", WAVES)
def test_log_mel(g6, seed, n):
    w = wave_case(seed, n)
    f = P.log_mel(w)
    assert f.shape == (80, 3000) and f.dtype == np.float32
    assert np.array_equal(f[:, ::25], g6[f"wave{seed}_logmel64_sub"])                  # HF float64 path: exact
    assert np.abs(f[:, ::25] - g6[f"wave{seed}_logmel_sub"]).max() < 5e-5             # HF float32 torch path: its own spread
    fn = P.whole_tensor_layer_norm(f)
    assert np.abs(fn[:, ::25] - g6[f"wave{seed}_norm_sub"]).max() < 5e-4
    s = g6[f"wave{seed}_norm_sum"]
    assert abs(fn.astype(np.float64).sum() - s[0]) < 0.5 and abs(np.abs(fn).astype(np.float64).sum() / s[1] - 1) < 1e-4
    assert abs(fn.mean()) < 1e-4 and abs(fn.std() - 1) < 1e-2


@pytest.mark.parametrize("seed,h,w", FRAMES)
def test_clip_frames_bit_exact(g6, seed, h, w):
    fr = frame_case(seed, h, w)
    u8 = P.clip_resized_crop_u8(fr)
    assert u8.shape == (224, 224, 3)
    assert zlib.crc32(np.ascontiguousarray(u8).tobytes()) == int(g6[f"frame{seed}_u8_crc"][0])
    assert np.array_equal(u8[::7, ::7], g6[f"frame{seed}_u8_sub"])
    assert np.array_equal(P.clip_pixel_values(fr)[:, ::7, ::7], g6[f"frame{seed}_px_sub"])


def test_resize_shape_rule():
    assert P.clip_resize_shape(96, 128) == (224, 298) and P.clip_resize_shape(300, 260) == (258, 224)
    assert P.clip_resize_shape(224, 224) == (224, 224) and P.clip_resize_shape(225, 223) == (226, 224)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_log_mel_128_bins(golden_dir, seed):
    """128 mel bins (openai/whisper-large-v3's extractor, BASELINE config 5) against tests/golden/g8_logmel128.npz, generated from
    transformers' WhisperFeatureExtractor(feature_size=128): exact on HF's float64 definition, inside HF's own spread on its torch path."""
    g8 = np.load(f"{golden_dir}/g8_logmel128.npz")
    _, n = (int(v) for v in g8[f"wave{seed}_n"])
    from oracle.make_golden_preproc import wave_case
    f = P.log_mel(wave_case(seed, n), 128)
    assert f.shape == (128, 3000) and f.dtype == np.float32
    assert np.array_equal(f[:, ::25], g8[f"wave{seed}_logmel64_sub"])
    assert np.abs(f[:, ::25] - g8[f"wave{seed}_logmel_sub"]).max() < 5e-5
    assert np.abs(P.mel_filter_bank(n_mels=128) - g8["mel_filters_128"]).max() < 1e-15
