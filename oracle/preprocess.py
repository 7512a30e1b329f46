"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement, in numpy, of the reference's per-sample feature
extraction -- the producer of the hot path's inputs.

Reference call sites: src/clip_whisper/data/simple_dataset.py:156-186 (audio: `whisper_processor(audio, sampling_rate=16000)`
then `F.layer_norm(features, features.shape)`), :191-264 (video: per-frame `clip_processor(images=frame)` on RGB uint8 frames).
The arithmetic itself lives in third-party code that is NOT under /root/reference and that the reference does not pin:
  * transformers (requirements.txt: >=4.30.0; 5.15.0 in this image): WhisperFeatureExtractor._np_extract_fbank_features
    (models/whisper/feature_extraction_whisper.py:105-133) -> audio_utils.spectrogram (:809-1017), mel_filter_bank (:638-730),
    window_function (:745-806); CLIPImageProcessor (PIL backend): resize(shortest_edge=224, BICUBIC) -> center_crop(224) ->
    rescale(1/255) -> normalize(mean, std).
  * Pillow (unpinned; 12.2.0 in this image): Image.resize(..., BICUBIC) = src/libImaging/Resample.c (precompute_coeffs,
    normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc): 8-bit fixed point, PRECISION_BITS = 22,
    horizontal pass first with a uint8 intermediate.
WhisperFeatureExtractor has two implementations of the same features: the float64 numpy one restated here (bit-identical, see
make_golden_preproc.py) and, when torch is importable -- as in the reference's environment -- a float32 torch.stft one
(:135-168) that HF documents as agreeing to 1e-5; the fixture holds both, measured spread 2e-6 .. 1.4e-5 (1.4e-4 after the
dataset's layer norm).  The Pillow resize is integer arithmetic and is restated bit-exactly.
Pinned by tests/golden/g6_preprocess.npz, generated here by oracle/make_golden_preproc.py from those libraries themselves."""
import math

import numpy as np

SAMPLE_RATE, N_FFT, HOP, N_MELS, N_SAMPLES, N_FRAMES = 16000, 400, 160, 80, 480000, 3000
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# --------------------------------------------------------------------------------------------- audio
def hann_window(n=N_FFT):
    """audio_utils.window_function("hann", periodic=True): np.hanning(n + 1)[:-1]."""
    return np.hanning(n + 1)[:-1]


def _hz_to_mel_slaney(f):
    """audio_utils.hertz_to_mel(mel_scale="slaney") :448-481."""
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * logstep, mels)


def _mel_to_hz_slaney(m):
    """audio_utils.mel_to_hertz(mel_scale="slaney") :484-517."""
    m = np.asarray(m, dtype=np.float64)
    logstep = np.log(6.4) / 27.0
    return np.where(m >= 15.0, 1000.0 * np.exp(logstep * (m - 15.0)), 200.0 * m / 3.0)


def mel_filter_bank(n_freqs=N_FFT // 2 + 1, n_mels=N_MELS, fmin=0.0, fmax=8000.0, sr=SAMPLE_RATE):
    """audio_utils.mel_filter_bank(norm="slaney", mel_scale="slaney") :638-730 -> [n_freqs, n_mels] float64."""
    mel_freqs = np.linspace(_hz_to_mel_slaney(fmin), _hz_to_mel_slaney(fmax), n_mels + 2)
    filter_freqs = _mel_to_hz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, n_freqs)
    diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    fb = np.maximum(0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2: n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def log_mel(wave, n_mels=N_MELS):
    """WhisperFeatureExtractor(feature_size=n_mels).__call__ on one mono 16 kHz waveform -> float32 [n_mels, 3000] (80 bins: Whisper
    tiny..large-v2; 128 bins: large-v3, BASELINE config 5): truncate / zero-pad to 30 s,
    reflect-pad 200, 3001 Hann frames hop 160, rfft in float64 STORED AS complex64 (audio_utils.py:966), |.|^2 in float64,
    slaney mel, floor 1e-10, log10, cast float32, drop the last frame, clamp to max-8, (x+4)/4."""
    x = np.zeros(N_SAMPLES, dtype=np.float32)
    w = np.asarray(wave, dtype=np.float32)[:N_SAMPLES]
    x[: w.size] = w
    xp = np.pad(x, (N_FFT // 2, N_FFT // 2), mode="reflect").astype(np.float64)
    win = hann_window()
    nfr = 1 + (xp.size - N_FFT) // HOP
    idx = np.arange(N_FFT)[None, :] + HOP * np.arange(nfr)[:, None]
    spec = np.fft.rfft(xp[idx] * win[None, :], axis=1).astype(np.complex64)
    power = np.abs(spec, dtype=np.float64) ** 2.0
    mel = np.maximum(1e-10, mel_filter_bank(n_mels=n_mels).T @ power.T)
    ls = np.log10(mel).astype(np.float32)[:, :-1]
    ls = np.maximum(ls, ls.max() - 8.0)
    return ((ls + 4.0) / 4.0).astype(np.float32)


def whole_tensor_layer_norm(f, eps=1e-5):
    """simple_dataset.py:183 `F.layer_norm(features, features.shape)`: zero mean / unit (biased) variance over ALL 80x3000."""
    f = np.asarray(f, dtype=np.float32)
    mu = f.mean(dtype=np.float64)
    var = ((f.astype(np.float64) - mu) ** 2).mean()
    return ((f - mu) / math.sqrt(var + eps)).astype(np.float32)


def audio_features(wave, normalize=True, n_mels=N_MELS):
    f = log_mel(wave, n_mels)
    return whole_tensor_layer_norm(f) if normalize else f


# --------------------------------------------------------------------------------------------- video
def _bicubic(x, a=-0.5):
    """Resample.c bicubic_filter."""
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


PRECISION_BITS = 32 - 8 - 2


def pil_coeffs(in_size, out_size, support=2.0, filt=_bicubic):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc -> (bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize])."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    sup = support * filterscale
    ksize = int(math.ceil(sup)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - sup + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + sup + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [filt((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        if ww != 0.0:
            w = [v / ww for v in w]
        for x, v in enumerate(w):
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis0_u8(img, out_size):
    """One 8-bit pass along axis 0: ss = 2^(P-1) + sum(pixel * k); clip8(ss >> P)."""
    bounds, kk = pil_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], dtype=np.uint8)
    src = img.astype(np.int64)
    for y in range(out_size):
        y0, n = bounds[y]
        ss = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for j in range(n):
            ss += src[y0 + j] * int(kk[y, j])
        out[y] = np.clip(ss >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def pil_bicubic_resize_u8(img, out_h, out_w):
    """Image.resize((out_w, out_h), BICUBIC) on an RGB uint8 HWC array: horizontal pass, then vertical, each rounding to uint8
    (ImagingResample: a pass is skipped when that dimension does not change)."""
    img = np.asarray(img, dtype=np.uint8)
    if out_w != img.shape[1]:
        img = _resample_axis0_u8(img.transpose(1, 0, 2), out_w).transpose(1, 0, 2)
    if out_h != img.shape[0]:
        img = _resample_axis0_u8(img, out_h)
    return img


def clip_resize_shape(h, w, size=224):
    """image_transforms.get_resize_output_image_size(shortest_edge): short -> size, long -> int(size * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)


def clip_resized_crop_u8(frame, size=224):
    """The uint8 image CLIPImageProcessor normalises: resize(shortest_edge) then center_crop(size, size) -> [size, size, 3]."""
    h, w = frame.shape[:2]
    nh, nw = clip_resize_shape(h, w, size)
    r = pil_bicubic_resize_u8(frame, nh, nw)
    top, left = (nh - size) // 2, (nw - size) // 2
    return r[top: top + size, left: left + size]


def clip_normalize_lut(mean=CLIP_MEAN, std=CLIP_STD):
    """rescale then normalize as the PIL backend does them (image_transforms.rescale :89-: float64 product rounded to float32;
    normalize :384-: float32 (x - mean) / std), for every uint8 value and channel -> float32 [3, 256]."""
    v = (np.arange(256, dtype=np.uint8).astype(np.float64) * (1 / 255)).astype(np.float32)
    return np.stack([(v - np.float32(mean[c])) / np.float32(std[c]) for c in range(3)]).astype(np.float32)


def clip_pixel_values(frame, size=224):
    """One RGB uint8 HWC frame -> float32 [3, size, size] (clip_processor(images=frame)["pixel_values"][0])."""
    u8 = clip_resized_crop_u8(frame, size)
    lut = clip_normalize_lut()
    return np.stack([lut[c][u8[:, :, c]] for c in range(3)])
