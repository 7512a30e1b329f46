#!/usr/bin/env python3
"""Decode -> WER entry point.  Takes the reference's command line as given (scripts/clip_whisper/decode.py:42-66:
--model_path --test_data --test_wrd --output_dir --modality --batch_size --max_new_tokens --temperature --device --seed --config
--verbose --whisper_model --clip_model --llm_model --calculate_loss --text_key --output_file) and does what that script does:
rebuild the model WITHOUT adapters, load ONLY the connector tensors from the checkpoint (decode.py:236-260), map utterance ids of the
manifest to the lines of the .wrd file (:318-372, with the path-less id as a second key), greedy `generate` at max_seq_len 256,
`batch_decode`, per-utterance and corpus WER, and the same files: decode_<ts>.log, results_<ts>.txt (the reference's table),
wer_<ts>.txt ("Overall WER" / "Total samples").

Additions of this build (not reference flags): --load_lora (also load the adapters from the checkpoint), --synthetic N / --tiny /
--synthetic-weights / --frames (no dataset or checkpoints offline), --data_path (root for relative media paths; default = the
reference's rule dirname(dirname(test_data))), and --test_manifest / --test_labels as aliases of --test_data / --test_wrd.
`--output_file` (declared but never written by the reference) receives the per-utterance results as JSON."""
import argparse
import json
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "audio-visual-llm_amd")):
    sys.path.insert(0, p) if p not in sys.path else None

import torch  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Run inference with the ClipWhisperModel and calculate WER")
    p.add_argument("--model_path", type=str, help="Path to the trained checkpoint (.pt)")
    p.add_argument("--test_data", "--test_manifest", dest="test_data", type=str, help="Path to test data TSV file")
    p.add_argument("--test_wrd", "--test_labels", dest="test_wrd", type=str, help="Path to test word reference file")
    p.add_argument("--output_dir", type=str, default="outputs/clip_whisper_decoding")
    p.add_argument("--modality", type=str, choices=["audio", "video", "both"], default="both")
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--max_new_tokens", type=int, default=100)
    p.add_argument("--temperature", type=float, default=1.0)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--config", type=str, default=os.path.join(ROOT, "configs", "clip_whisper.yaml"))
    p.add_argument("--verbose", action="store_true")
    p.add_argument("--whisper_model", type=str, default=None, help="overrides the YAML's whisper_model")
    p.add_argument("--clip_model", type=str, default=None, help="overrides the YAML's clip_model")
    p.add_argument("--llm_model", type=str, default=None, help="overrides the YAML's llm_path")
    p.add_argument("--calculate_loss", action="store_true", help="also report the eval loss against the reference text")
    p.add_argument("--text_key", type=str, default="text")
    p.add_argument("--output_file", type=str, default="decode_results.json")
    # ---- this build's additions
    p.add_argument("--load_lora", action="store_true")
    p.add_argument("--data_path", type=str, default=None)
    p.add_argument("--synthetic", type=int, default=0)
    p.add_argument("--tiny", action="store_true")
    p.add_argument("--frames", type=int, default=125)
    p.add_argument("--synthetic-weights", action="store_true", help="seeded random weights + byte tokenizer (implied by --tiny)")
    return p.parse_args(argv)


def match_references(ids, texts):
    """utterance id -> reference text, ids and .wrd lines paired by position (decode.py:318-372); the last path component of an id is a
    second key.  `ids` are the manifest's well-formed entries (the dataset's own list, so ids and samples cannot drift apart: the
    reference script re-parses the TSV with a looser rule than its dataset and shifts every reference by one on a malformed line)."""
    if len(ids) != len(texts):
        logging.warning(f"Mismatch between number of utterance IDs ({len(ids)}) and reference texts ({len(texts)})")
    refs = {}
    for uid, t in zip(ids, texts):
        refs[uid] = t
        if "/" in uid:
            refs[uid.split("/")[-1]] = t
    return refs


def main(argv=None):
    a = parse_args(argv)
    torch.manual_seed(a.seed)
    os.makedirs(a.output_dir, exist_ok=True)
    ts = time.strftime("%Y%m%d_%H%M%S")
    root_logger = logging.getLogger()
    for h in root_logger.handlers[:]:
        root_logger.removeHandler(h)
    fh = logging.FileHandler(os.path.join(a.output_dir, f"decode_{ts}.log"))
    fh.setLevel(logging.DEBUG if a.verbose else logging.INFO)
    ch = logging.StreamHandler()
    ch.setLevel(logging.WARNING)
    root_logger.setLevel(logging.DEBUG if a.verbose else logging.INFO)
    root_logger.addHandler(fh); root_logger.addHandler(ch)
    logging.info(f"Model path: {a.model_path}  Modality: {a.modality}  Test data: {a.test_data}  Test references: {a.test_wrd}  Device: {a.device}")
    print("\n" + "=" * 80 + f"\nCLIP-WHISPER DECODING\nModel: {a.model_path}\nModality: {a.modality}\n" + "=" * 80 + "\n")

    from avllm.config import merged
    from avllm.model import ClipWhisperModel
    from avllm.wer import calculate_wer
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from train import SyntheticClips
    cfg = merged(a.config, {"llm_path": a.llm_model, "whisper_model": a.whisper_model, "clip_model": a.clip_model})
    dev = "cuda:0" if a.device in ("cuda", "gpu") else a.device
    if not dev.startswith("cuda"):
        raise ValueError(f"--device {a.device}: this build runs the path on an MI355X only (there is no CPU fallback)")
    kw = {}
    if a.tiny:
        from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
        kw["config"] = ModelCfg(WhisperCfg(128, 2, 2, 256), ClipCfg(128, 2, 2, 256, 48, 16), LlamaCfg(256, 2, 2, 512, 512), LoraCfg(16, 32.0))
    model = ClipWhisperModel(llm_path=cfg["llm_path"], whisper_model=cfg["whisper_model"], clip_model=cfg["clip_model"], device=dev,
                             use_fp16=bool(cfg.get("use_fp16")), use_lora=a.load_lora, modality=a.modality, max_seq_len=256,
                             synthetic_weights=a.synthetic_weights or a.tiny, **kw).eval()
    if a.model_path:
        ck = torch.load(a.model_path, map_location="cpu", weights_only=True)
        sd = ck.get("model_state_dict", ck)
        keep = {k: v for k, v in sd.items() if "audio_connector" in k or "video_connector" in k or (a.load_lora and "lora_" in k)}
        if not any("audio_connector" in k for k in keep):
            logging.warning("No audio connector weights found in checkpoint")
        if not any("video_connector" in k for k in keep):
            logging.warning("No video connector weights found in checkpoint")
        model.load_state_dict(keep)
        mc = ck.get("config") if isinstance(ck, dict) else None
        if isinstance(mc, dict):
            model.max_seq_len = mc.get("max_seq_len", 256); model.fusion_scale = mc.get("fusion_scale", 0.5)

    references, utt_ids = {}, None
    if a.synthetic:
        ds = SyntheticClips(a.synthetic, model.cfg, 5 if a.tiny else a.frames, model.tokenizer, 11)
        loader = torch.utils.data.DataLoader(ds, batch_size=a.batch_size, collate_fn=ds.collate)
        utt_ids = [f"synthetic_{i}" for i in range(len(ds))]
        references = {utt_ids[i]: ds[i][2] for i in range(len(ds))}

        def batches():
            for b in loader:
                yield b[0], b[1], b[2]
    else:
        droot = a.data_path or cfg.get("path") or "."
        test_data = a.test_data or os.path.join(droot, cfg.get("test_manifest", "test.tsv"))
        test_wrd = a.test_wrd or os.path.join(droot, cfg.get("test_labels", "test.wrd"))
        if not os.path.exists(test_data) or not os.path.exists(test_wrd):
            print("ERROR: Both --test_data and --test_wrd are required for batch decoding mode")
            logging.error("Both --test_data and --test_wrd are required for batch decoding mode")
            return 1
        from avllm.data import AVSRDataset
        from avllm.preprocess import ClipFrames, WhisperLogMel, device_collate
        root = a.data_path or os.path.dirname(os.path.dirname(os.path.abspath(test_data)))       # decode.py:394
        ds = AVSRDataset(test_data, test_wrd, root, model.tokenizer, max_audio_length=30, max_video_length=300, split="test", modality=a.modality)
        utt_ids = [n[2] for n in ds.names]
        references = match_references(utt_ids, ds.labels)
        loader = torch.utils.data.DataLoader(ds, batch_size=a.batch_size, shuffle=False, collate_fn=AVSRDataset.collate_fn)
        feats = (WhisperLogMel(dev, n_mels=model.cfg.whisper.n_mels), ClipFrames(dev, image=model.cfg.clip.image))

        def batches():
            for b in loader:
                audio, video = device_collate(b["raw"], *feats)
                yield audio, video, b["texts"]

    results, all_refs, all_hyps, losses = [], [], [], []
    seen = 0
    print(f"Starting decoding with modality: {a.modality}\nEach hypothesis will be shown as it's generated.\n")
    for bi, (audio, video, texts) in enumerate(batches()):
        try:
            audio = None if (audio is None or a.modality == "video") else audio.to(dev)
            video = None if (video is None or a.modality == "audio") else video.to(dev)
            ids = model.generate(audio=audio, video=video, max_new_tokens=a.max_new_tokens, temperature=a.temperature)
            out = model.tokenizer.batch_decode(ids.cpu(), skip_special_tokens=True)
            if a.calculate_loss:
                lab = model.tokenizer(list(texts), padding="max_length", truncation=True, max_length=256, return_tensors="pt").input_ids
                with torch.no_grad():
                    losses.append(float(model(audio=audio, video=video, labels=lab.to(dev), return_loss=True)["loss"]))
            for j, hyp in enumerate(out):
                uid = utt_ids[seen + j] if seen + j < len(utt_ids) else f"unknown_{bi}_{j}"
                hyp = hyp.strip()
                ref = references.get(uid, references.get(uid.split("/")[-1]) if "/" in uid else None)
                print(f"\nUTT: {uid}\nHYP: {hyp}\nREF: {ref if ref is not None else '[None]'}\n" + "-" * 40)
                if ref is not None:
                    all_refs.append(ref); all_hyps.append(hyp)
                    results.append({"utt_id": uid, "hypothesis": hyp, "reference": ref, "wer": calculate_wer([ref], [hyp])})
            seen += len(out)
        except Exception as e:                                   # noqa: BLE001  decode.py:647-650: log and go on with the next batch
            logging.error(f"Error processing batch {bi}: {e}")
            seen += len(texts)
            continue
    if not all_refs:
        logging.warning("No samples were successfully processed for WER calculation")
        print("\nNo samples were successfully processed for WER calculation")
        return 1
    wer = calculate_wer(all_refs, all_hyps)
    with open(os.path.join(a.output_dir, f"results_{ts}.txt"), "w") as f:
        f.write(f"Modality: {a.modality}\nOverall WER: {wer:.4f}\n\nDetailed Results:\n")
        f.write(f"{'Utterance ID':<20} {'WER':<10} {'Reference':<40} {'Hypothesis':<40}\n" + "-" * 110 + "\n")
        for r in results:
            f.write(f"{r['utt_id']:<20} {r['wer']:<10.4f} {r['reference'][:40]:<40} {r['hypothesis'][:40]:<40}\n")
    with open(os.path.join(a.output_dir, f"wer_{ts}.txt"), "w") as f:
        f.write(f"Overall WER: {wer:.4f}\nTotal samples: {len(all_refs)}\n")
    summary = {"modality": a.modality, "overall_wer": wer, "total_samples": len(all_refs), "results": results}
    if losses:
        summary["mean_loss"] = sum(losses) / len(losses)
    json.dump(summary, open(os.path.join(a.output_dir, os.path.basename(a.output_file)), "w"), indent=1)
    logging.info(f"Overall WER: {wer:.4f}")
    print("\n" + "=" * 80 + f"\nDECODING SUMMARY\nModality: {a.modality}\nOverall WER: {wer:.4f}\nTotal samples: {len(all_refs)}\n" + "=" * 80)
    return 0


if __name__ == "__main__":
    sys.exit(main())
