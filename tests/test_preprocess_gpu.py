"""Device-side feature extraction (csrc/preprocess.hip through the C ABI) against the CPU oracle and the HF/Pillow fixture.
Bars: the resize/crop is integer work -> the float32 pixel_values must be BIT-IDENTICAL to the oracle (the 256-entry
normalisation table is injective, so this also proves the uint8 image identical); the log-mel is float64 arithmetic rounded to
float32 where HF rounds -> within 2e-6 of the oracle (a last-ulp float32 flip), 2e-5 after the whole-tensor layer norm."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import preprocess as P  # noqa: E402
from oracle.make_golden_preproc import FRAMES, WAVES, frame_case, wave_case  # noqa: E402


@pytest.fixture(scope="module")
def g6(golden_dir):
    return np.load(f"{golden_dir}/g6_preprocess.npz")


def test_log_mel_batch_vs_oracle_and_fixture(dev, g6):
    from avllm.preprocess import WhisperLogMel
    waves = [wave_case(s, n) for s, n in WAVES]
    raw = WhisperLogMel(dev, normalize=False)(waves).cpu().numpy()             # ragged batch: 5 s, 30 s, 37.5 s (truncated), 7 samples
    nrm = WhisperLogMel(dev, normalize=True)(waves).cpu().numpy()
    assert raw.shape == (4, 80, 3000)
    for i, (s, n) in enumerate(WAVES):
        ref = P.log_mel(waves[i])
        assert np.abs(raw[i] - ref).max() < 2e-6, (s, np.abs(raw[i] - ref).max())
        assert (raw[i] != ref).mean() < 0.02                                     # nearly all elements bit-equal
        assert np.abs(raw[i][:, ::25] - g6[f"wave{s}_logmel_sub"]).max() < 5e-5  # what the reference's torch path produces
        refn = P.whole_tensor_layer_norm(ref)
        assert np.abs(nrm[i] - refn).max() < 2e-5, (s, np.abs(nrm[i] - refn).max())
        assert np.abs(nrm[i][:, ::25] - g6[f"wave{s}_norm_sub"]).max() < 5e-4


def test_log_mel_silence_and_single_row(dev):
    from avllm.preprocess import WhisperLogMel
    lm = WhisperLogMel(dev, normalize=False)
    z = lm(torch.zeros(1, 16000)).cpu().numpy()[0]
    assert np.array_equal(z, P.log_mel(np.zeros(16000, dtype=np.float32)))       # floor everywhere: (-10 + 4) / 4
    assert np.all(z == np.float32(-1.5))
    w = wave_case(9, 48000)
    assert np.abs(lm(torch.from_numpy(w)).cpu().numpy()[0] - P.log_mel(w)).max() < 2e-6


@pytest.mark.parametrize("seed,h,w", FRAMES + [(7, 8, 8), (8, 50, 1000), (9, 1000, 50), (10, 223, 640), (11, 16, 5200)])
def test_clip_frames_bit_exact(dev, seed, h, w):
    from avllm.preprocess import ClipFrames
    fr = np.stack([frame_case(seed, h, w), frame_case(seed + 100, h, w), 255 - frame_case(seed, h, w)])
    out = ClipFrames(dev)(torch.from_numpy(fr)).cpu().numpy()
    assert out.shape == (3, 3, 224, 224) and out.dtype == np.float32
    for i in range(3):
        assert np.array_equal(out[i], P.clip_pixel_values(fr[i])), (seed, i, np.abs(out[i] - P.clip_pixel_values(fr[i])).max())


def test_clip_frames_fixture_and_bf16(dev, g6):
    from avllm.preprocess import ClipFrames
    for seed, h, w in FRAMES:
        fr = frame_case(seed, h, w)
        out = ClipFrames(dev)(torch.from_numpy(fr[None]))[0].cpu().numpy()
        assert np.array_equal(out[:, ::7, ::7], g6[f"frame{seed}_px_sub"])          # CLIPImageProcessor's own output
    fr = frame_case(3, 300, 260)
    o16 = ClipFrames(dev, dtype=torch.bfloat16)(torch.from_numpy(fr[None]))[0].float().cpu()
    assert torch.equal(o16, torch.from_numpy(P.clip_pixel_values(fr)).bfloat16().float())


def test_plan_mismatch_is_refused(dev):
    from avllm import lib as L
    from avllm.preprocess import ClipFrames
    cf = ClipFrames(dev)
    plan = cf._plan(96, 96)
    fr = torch.zeros(1, 64, 64, 3, dtype=torch.uint8, device=dev)
    out = torch.empty(1, 3, 224, 224, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    rc = L.load().avllm_clip_preproc(L.ptr(plan), L.ptr(fr), 1, 64, 64, 224, L.ptr(out), L.F32, L.ptr(ws), ws.numel(), L.stream_ptr())
    assert rc != 0
    with pytest.raises(ValueError):
        cf(torch.zeros(1, 3, 64, 64, dtype=torch.uint8))


def test_model_from_raw_inputs_equals_model_from_oracle_features(dev):
    """Raw samples -> device_collate -> train forward == oracle features -> the same forward (tiny model, CLIP image 48)."""
    from oracle import weights as Wt
    from test_model_gpu import make_model
    from avllm.preprocess import ClipFrames, WhisperLogMel, device_collate
    oc = Wt.tiny()
    W = Wt.all_weights(oc, 11, lora_b_std=0.05)
    m = make_model(oc, W, "fp32").train()
    S = oc.clip.image
    samples = [{"wave": wave_case(21, 40000), "frames": np.stack([frame_case(30 + i, 40, 56) for i in range(3)])},
               {"wave": wave_case(22, 90000), "frames": np.stack([frame_case(40 + i, 40, 56) for i in range(5)])}]
    audio, video = device_collate(samples, WhisperLogMel(dev), ClipFrames(dev, image=S))
    assert audio.shape == (2, 80, 3000) and video.shape == (2, 5, 3, S, S) and float(video[0, 3:].abs().max()) == 0.0
    ref_a = torch.from_numpy(np.stack([P.audio_features(s["wave"]) for s in samples]))
    ref_v = torch.zeros(2, 5, 3, S, S)
    for i, s in enumerate(samples):
        ref_v[i, : len(s["frames"])] = torch.from_numpy(np.stack([P.clip_pixel_values(f, S) for f in s["frames"]]))
    assert torch.equal(video.cpu(), ref_v) and (audio.cpu() - ref_a).abs().max() < 2e-5
    _, _, labels, _ = Wt.synthetic_batch(oc, 2, 5, seed=3)
    o1 = m(audio=audio, video=video, prompt=None, labels=labels.to(dev))
    o2 = m(audio=ref_a.to(dev), video=ref_v.to(dev), prompt=None, labels=labels.to(dev))
    assert (o1["logits"] - o2["logits"]).abs().max() < 1e-3 and abs(float(o1["loss"].detach()) - float(o2["loss"].detach())) < 1e-4


def test_log_mel_128_bins_vs_oracle_and_fixture(dev, golden_dir):
    """The device log-mel with 128 bins (Whisper-large-v3, BASELINE config 5): kernel vs the numpy oracle and the HF-generated fixture g8."""
    from avllm.preprocess import WhisperLogMel
    from oracle.make_golden_preproc import wave_case
    g8 = np.load(f"{golden_dir}/g8_logmel128.npz")
    lm = WhisperLogMel(dev, normalize=False, n_mels=128)
    cases = [tuple(int(v) for v in g8[f"wave{s}_n"]) for s in (1, 2, 3)]
    waves = [wave_case(s, n) for s, n in cases]
    raw = lm(waves).cpu().numpy()
    assert raw.shape == (3, 128, 3000)
    for i, (s, n) in enumerate(cases):
        assert np.abs(raw[i] - P.log_mel(waves[i], 128)).max() < 2e-6
        assert np.abs(raw[i][:, ::25] - g8[f"wave{s}_logmel_sub"]).max() < 5e-5
    normed = WhisperLogMel(dev, normalize=True, n_mels=128)(waves[:1]).cpu().numpy()[0]
    assert np.abs(normed - P.audio_features(waves[0], n_mels=128)).max() < 2e-5
    with pytest.raises(ValueError):
        WhisperLogMel(dev, n_mels=100)
