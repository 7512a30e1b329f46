#!/usr/bin/env python3
"""Decode -> WER entry point with the reference's flag names (scripts/clip_whisper/decode.py:41-68): rebuild the model,
load ONLY connector tensors from the checkpoint (decode.py:236-260; pass --load_lora to also load the adapters), greedy
generate at max_seq_len 256, batch_decode, per-utterance and corpus WER, results_<ts>.txt / wer_<ts>.txt."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-llm_amd")):
    sys.path.insert(0, p) if p not in sys.path else None

import torch  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--model_path"); p.add_argument("--config", default=os.path.join(ROOT, "configs", "clip_whisper.yaml"))
    p.add_argument("--output_dir", default="outputs/decode"); p.add_argument("--modality", default="both")
    p.add_argument("--batch_size", type=int, default=4); p.add_argument("--max_new_tokens", type=int, default=100)
    p.add_argument("--temperature", type=float, default=1.0); p.add_argument("--load_lora", action="store_true")
    p.add_argument("--data_path"); p.add_argument("--test_manifest"); p.add_argument("--test_labels")
    p.add_argument("--synthetic", type=int, default=0); p.add_argument("--tiny", action="store_true"); p.add_argument("--frames", type=int, default=125)
    p.add_argument("--synthetic-weights", action="store_true", help="seeded random weights + byte tokenizer (implied by --tiny)")
    a = p.parse_args()
    from avllm.config import merged
    from avllm.model import ClipWhisperModel
    from avllm.wer import calculate_wer
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from train import SyntheticClips
    cfg = merged(a.config, {})
    kw = {}
    if a.tiny:
        from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
        kw["config"] = ModelCfg(WhisperCfg(128, 2, 2, 256), ClipCfg(128, 2, 2, 256, 48, 16), LlamaCfg(256, 2, 2, 512, 512), LoraCfg(16, 32.0))
    model = ClipWhisperModel(llm_path=cfg["llm_path"], whisper_model=cfg["whisper_model"], clip_model=cfg["clip_model"], device="cuda:0",
                             use_fp16=bool(cfg.get("use_fp16")), use_lora=a.load_lora, modality=a.modality, max_seq_len=256,
                             synthetic_weights=a.synthetic_weights or a.tiny, **kw).eval()
    if a.model_path:
        ck = torch.load(a.model_path, map_location="cpu", weights_only=True)
        sd = ck.get("model_state_dict", ck)
        keep = {k: v for k, v in sd.items() if "audio_connector" in k or "video_connector" in k or (a.load_lora and "lora_" in k)}
        model.load_state_dict(keep)
    if a.synthetic:
        ds = SyntheticClips(a.synthetic, model.cfg, 5 if a.tiny else a.frames, model.tokenizer, 11)
        batches = torch.utils.data.DataLoader(ds, batch_size=a.batch_size, collate_fn=ds.collate)
    else:
        # test manifest of the YAML (data.path / test_manifest / test_labels): raw samples, features on the device
        from avllm.data import create_dataloaders
        from avllm.preprocess import ClipFrames, WhisperLogMel, device_collate
        root = a.data_path or cfg.get("path") or "."
        mp = os.path.join(root, a.test_manifest or cfg.get("test_manifest", "test.tsv"))
        lp = os.path.join(root, a.test_labels or cfg.get("test_labels", "test.wrd"))
        dl, _ = create_dataloaders(mp, lp, root, model.tokenizer, batch_size=a.batch_size, modality=a.modality, shuffle=False)
        feats = (WhisperLogMel("cuda:0", n_mels=model.cfg.whisper.n_mels), ClipFrames("cuda:0", image=model.cfg.clip.image))

        def batches_from_files():
            for b in dl:
                audio, video = device_collate(b["raw"], *feats)
                yield audio, video, b["texts"], b["labels"]
        batches = batches_from_files()
    os.makedirs(a.output_dir, exist_ok=True)
    ts = time.strftime("%Y%m%d_%H%M%S")
    refs, hyps = [], []
    with open(os.path.join(a.output_dir, f"results_{ts}.txt"), "w") as f:
        for audio, video, texts, _ in batches:
            audio = audio if audio is None or a.modality == "video" else audio.cuda()
            video = video if video is None or a.modality == "audio" else video.cuda()
            ids = model.generate(audio=audio if a.modality != "video" else None, video=video if a.modality != "audio" else None,
                                 max_new_tokens=a.max_new_tokens, temperature=a.temperature)
            out = model.tokenizer.batch_decode(ids.cpu(), skip_special_tokens=True)
            for r, h in zip(texts, out):
                f.write(f"REF: {r}\nHYP: {h}\nWER: {calculate_wer([r], [h]):.4f}\n\n")
            refs += texts; hyps += out
    wer = calculate_wer(refs, hyps)
    open(os.path.join(a.output_dir, f"wer_{ts}.txt"), "w").write(f"WER: {wer:.4f}\nutterances: {len(refs)}\n")
    print(f"corpus WER {wer:.4f} over {len(refs)} utterances")


if __name__ == "__main__":
    main()
