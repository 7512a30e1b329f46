"""CPU: host-side logic that needs no GPU -- config loading, tokenizer stand-in, WER, LR schedule, DDP reducer (gloo, ws=2)."""
import math
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_wer_product_matches_oracle():
    from avllm.wer import calculate_wer
    from oracle import avsr_oracle as O
    cases = [("a b c d", "a x c"), ("hello world", "hello world"), ("a", ""), ("the cat sat", "cat sat on the mat")]
    for r, h in cases:
        assert calculate_wer(r, h) == pytest.approx(O.wer(r, h))
    assert calculate_wer([c[0] for c in cases], [c[1] for c in cases]) == pytest.approx(O.wer([c[0] for c in cases], [c[1] for c in cases]))
    assert calculate_wer("a b c d", "a x c") == pytest.approx(0.5)


def test_config_yaml_is_honoured():
    from avllm.config import merged
    cfg = merged(os.path.join(ROOT, "configs", "clip_whisper.yaml"), {"learning_rate": 1e-4, "batch_size": None})
    assert cfg["max_seq_len"] == 512 and cfg["lora_r"] == 16 and cfg["whisper_model"] == "openai/whisper-small"
    assert cfg["learning_rate"] == 1e-4 and cfg["batch_size"] == 8 and cfg["train_manifest"] == "train.tsv"


def test_byte_tokenizer_roundtrip():
    from avllm.tokenizer import ByteTokenizer
    t = ByteTokenizer(512)
    out = t(["hello world", "a"], padding="max_length", max_length=16, truncation=True)
    assert out.input_ids.shape == (2, 16) and out.input_ids[0, 0] == 1 and out.input_ids[1, 5] == t.pad_token_id
    assert t.batch_decode(out.input_ids) == ["hello world", "a"]


def test_lr_schedule_matches_oracle():
    from avllm.trainer import ClipWhisperTrainer
    from oracle import avsr_oracle as O
    tr = ClipWhisperTrainer.__new__(ClipWhisperTrainer)
    tr.learning_rate, tr.warmup_steps, tr.total_steps = 5e-5, 0, 10
    for s in range(10):
        assert tr.lr_at(s) == pytest.approx(O.cosine_lr(5e-5, s, 10))


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.join(ROOT, "audio-visual-llm_amd")); sys.path.insert(0, ROOT)
    from avllm.dist import LoraGradReducer
    from oracle import avsr_oracle as O
    from oracle import weights as Wt
    torch.set_num_threads(2)
    oc = Wt.tiny()
    oc.llama = Wt.LlamaCfg(hidden=128, heads=1, layers=2, ffn=256, vocab=64)
    W = {"llama": Wt.llama_weights(oc.llama, 0), "lora": Wt.lora_weights(oc.llama, oc.lora, 0, 0.05)}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4, 32, 128, generator=g)
    labels = torch.randint(3, 64, (4, 32), generator=g)
    labels[0, 5:] = -100; labels[1, 20:] = -100; labels[2, 9:] = -100           # ranks see different token counts
    keys = sorted(W["lora"])
    per_layer = sum(W["lora"][k].numel() for k in keys if k.startswith("layers.0."))

    def grads(xs, ls, denom=None):
        lora = {k: v.clone().requires_grad_(True) for k, v in W["lora"].items()}
        h = O.llama_hidden(W["llama"], lora, oc.llama, oc.lora, xs)
        logits = h @ W["llama"]["lm_head.weight"].T
        shift = torch.cat([ls[:, 1:], torch.full((ls.shape[0], 1), -100)], 1)
        loss_sum = torch.nn.functional.cross_entropy(logits.view(-1, 64), shift.reshape(-1), ignore_index=-100, reduction="sum")
        cnt = (shift != -100).sum().float()
        (loss_sum / (denom if denom is not None else cnt)).backward()
        return torch.cat([lora[k].grad.flatten() for k in keys]), loss_sum.detach(), cnt

    # this rank's half of the global batch: sum-CE / ALL-REDUCED count, then SUM all-reduce in per-layer buckets
    sl = slice(rank * 2, rank * 2 + 2)
    _, ls_local, cnt_local = grads(x[sl], labels[sl])
    acc = torch.stack([ls_local, cnt_local])
    flat = torch.zeros(len(keys) and sum(W["lora"][k].numel() for k in keys))
    red = LoraGradReducer(flat, per_layer, oc.llama.layers)
    assert red.enabled
    red.reduce_counts(acc)
    gl, _, _ = grads(x[sl], labels[sl], denom=acc[1])
    # flat layout must be per-layer contiguous for the bucket slices: keys sorted = layers.0.*, layers.1.*
    flat.copy_(gl)
    for layer in reversed(range(oc.llama.layers)):
        red.layer_done(layer)
    red.finish()
    ref, ref_sum, ref_cnt = grads(x, labels)                 # single process on the concatenated batch
    if rank == 0:
        q.put((float((flat - ref).abs().max()), float(ref.abs().max()), float(acc[0] / acc[1]), float(ref_sum / ref_cnt)))
    dist.destroy_process_group()


def test_ddp_two_ranks_equal_single_process_on_concatenated_batch():
    """N-rank gradients (token-count-weighted, bucketed SUM all-reduce) == 1-rank gradients at batch N*B."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    err, scale, loss_ddp, loss_ref = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err <= 1e-5 * max(1.0, scale) + 1e-7, (err, scale)
    assert abs(loss_ddp - loss_ref) < 1e-5


def test_unresolvable_model_paths_raise_instead_of_random_weights(tmp_path):
    """A model path that is not a local checkpoint (the default YAML's `meta-llama/Llama-2-7b-hf`, or a typo) must not silently become a
    seeded-random model with a byte tokenizer: FileNotFoundError unless synthetic_weights=True was asked for."""
    import pytest
    from avllm.arch import resolve_arch
    from avllm.tokenizer import ByteTokenizer, load_tokenizer
    args = ("meta-llama/Llama-2-7b-hf", "openai/whisper-small", "openai/clip-vit-base-patch16", None, None, 0, 16, 32, True, None, None, None,
            "cpu", torch.float32)
    with pytest.raises(FileNotFoundError, match="synthetic_weights"):
        resolve_arch(*args)
    with pytest.raises(FileNotFoundError):
        load_tokenizer(str(tmp_path), 32000)                      # a directory without tokenizer files
    assert isinstance(load_tokenizer(str(tmp_path), 32000, synthetic=True), ByteTokenizer)
    (tmp_path / "tokenizer.json").write_text("{ not json")
    with pytest.raises(Exception) as ei:                            # a BROKEN tokenizer must surface, not turn into the byte stand-in
        load_tokenizer(str(tmp_path), 32000, synthetic=True)
    assert not isinstance(ei.value, FileNotFoundError) or "tokenizer files" not in str(ei.value)


def test_reference_trainer_checkpoint_maps_to_build_weights(golden_dir):
    """tests/golden/n1_reference_model_best.pt is the file the REFERENCE trainer's own _save_checkpoint wrote (oracle/make_golden_checkpoint.py).
    The safe loader accepts it; its full model_state_dict maps onto this build's `weights=` layout: frozen tensors equal the seeded weights
    the reference model was built from, LoRA tensors (two optimizer steps old) differ from their initial values, connectors untouched
    (no gradient reaches them)."""
    import numpy as np
    from avllm.arch import weights_from_reference_state_dict
    from oracle import weights as Wt
    from oracle.make_golden_checkpoint import micro_cfg
    ck = torch.load(f"{golden_dir}/n1_reference_model_best.pt", map_location="cpu", weights_only=True)
    assert {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "train_losses", "val_losses", "best_val_loss"} <= set(ck)
    W = weights_from_reference_state_dict(ck["model_state_dict"])
    cfg = micro_cfg()
    W0 = Wt.all_weights(cfg, int(np.load(f"{golden_dir}/n1_expected.npz")["weights_seed"]), lora_b_std=0.05)
    assert set(W) == {"whisper", "clip", "llama", "lora", "audio_connector", "video_connector"}
    for part in ("whisper", "clip", "llama", "audio_connector", "video_connector"):
        for k, v in W0[part].items():
            kk = k if k in W[part] else ("vision_model." + k if "vision_model." + k in W[part] else k)
            assert kk in W[part], (part, k, list(W[part])[:4])
            assert torch.equal(W[part][kk], v), (part, k)
    assert set(W["lora"]) == set(W0["lora"])
    assert any(not torch.equal(W["lora"][k], W0["lora"][k]) for k in W0["lora"])
    st = ck["optimizer_state_dict"]["state"]
    assert len([i for i in st if "exp_avg" in st[i]]) == len(W["lora"]) and ck["scheduler_state_dict"]["last_epoch"] == 2
