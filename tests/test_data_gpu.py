"""Files on disk -> avllm.data.AVSRDataset -> device-side features -> trainer steps, against the oracle fed with features the
oracle's own preprocessing made from the same files."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import avsr_oracle as O  # noqa: E402
from oracle import preprocess as P  # noqa: E402
from oracle import weights as Wt  # noqa: E402
from test_data_cpu import make_set  # noqa: E402
from test_model_gpu import make_model  # noqa: E402


def test_training_step_from_files_matches_oracle(dev, tmp_path):
    from avllm.data import AVSRDataset, create_dataloaders
    from avllm.trainer import ClipWhisperTrainer
    oc = Wt.tiny()
    W = Wt.all_weights(oc, 21, lora_b_std=0.05)
    m = make_model(oc, W, "fp32").train()
    mp, lp = make_set(tmp_path, n=4)
    dl, _ = create_dataloaders(str(mp), str(lp), str(tmp_path), m.tokenizer, batch_size=2, shuffle=False, max_video_length=4)
    tr = ClipWhisperTrainer(m, dl, None, learning_rate=1e-3, max_epochs=1, total_steps=10, grad_clip=0.5)
    batch = next(iter(dl))
    audio, video, labels, prompt = tr._unpack(batch)
    S = oc.clip.image
    assert audio.shape == (2, 80, 3000) and video.shape == (2, 4, 3, S, S)
    # the oracle's view of the same two files
    ds = AVSRDataset(str(mp), str(lp), str(tmp_path), m.tokenizer, max_video_length=4)
    ref_a = torch.from_numpy(np.stack([P.audio_features(ds[i]["wave"]) for i in range(2)]))
    ref_v = torch.zeros(2, 4, 3, S, S)
    for i in range(2):
        fr = ds[i]["frames"]
        ref_v[i, : len(fr)] = torch.from_numpy(np.stack([P.clip_pixel_values(f, S) for f in fr]))
    assert torch.equal(video.cpu(), ref_v) and (audio.cpu() - ref_a).abs().max() < 2e-5
    ref_loss, _, _ = O.train_step_grads(W, oc, ref_a, ref_v, prompt, labels)
    loss = tr.train_step(audio, video, labels, prompt)
    assert abs(float(loss) - float(ref_loss)) < 2e-4


def test_cli_train_then_decode_from_manifests(dev, tmp_path):
    """scripts/clip_whisper/train.py and decode.py on a toy LRS3-style directory: files -> device features -> LoRA steps ->
    model_final.pt with the reference's checkpoint keys -> greedy decode -> results / WER files (train.py:35-80, decode.py:42-66)."""
    import glob
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = tmp_path / "toy"
    data.mkdir()
    mp, lp = make_set(data, n=4)
    (data / "test.tsv").write_text(mp.read_text()); (data / "test.wrd").write_text(lp.read_text())
    out, dec = tmp_path / "out", tmp_path / "dec"
    env = dict(os.environ, PYTHONPATH=root)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts/clip_whisper/train.py"), "--tiny", "--data_path", str(data), "--batch_size", "2",
                        "--max_epochs", "1", "--output_dir", str(out), "--save_steps", "100", "--log_param_updates"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    ck = torch.load(out / "model_final.pt", map_location="cpu", weights_only=True)
    keys = ck["model_state_dict"].keys()
    assert "audio_connector.linear.weight" in keys and any(k.endswith("self_attn.q_proj.lora_A.default.weight") for k in keys)
    assert os.path.exists(out / "training.log")
    # the reference's own command line (decode.py:42-66), flag for flag; --tiny / --data_path are this build's additions (no checkpoints
    # offline; the toy media sit next to the manifest rather than one directory above it)
    r = subprocess.run([sys.executable, os.path.join(root, "scripts/clip_whisper/decode.py"), "--model_path", str(out / "model_final.pt"),
                        "--test_data", str(data / "test.tsv"), "--test_wrd", str(data / "test.wrd"), "--output_dir", str(dec), "--modality", "both",
                        "--batch_size", "2", "--max_new_tokens", "4", "--temperature", "1.0", "--device", "cuda", "--seed", "42", "--verbose",
                        "--whisper_model", "openai/whisper-small", "--clip_model", "openai/clip-vit-base-patch16", "--llm_model", "meta-llama/Llama-2-7b-hf",
                        "--output_file", "decode_results.json", "--tiny", "--data_path", str(data)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Overall WER:" in r.stdout and "UTT: id0" in r.stdout and "REF: hello world 0" in r.stdout
    res = open(glob.glob(str(dec / "results_*.txt"))[0]).read()
    assert res.startswith("Modality: both\nOverall WER: ") and "Utterance ID" in res and "hello world 3" in res
    assert open(glob.glob(str(dec / "wer_*.txt"))[0]).read().split("\n")[1] == "Total samples: 4"
    assert glob.glob(str(dec / "decode_*.log")) and os.path.exists(dec / "decode_results.json")
