"""__graft_entry__.smoke(): one tiny train step of the hot path on cuda:0 through libavllm.so, checked against
the CPU oracle (the oracle is the checker here, never the thing run)."""
import torch


def run():
    from oracle import avsr_oracle as O
    from oracle import weights as Wt
    from .model import ClipWhisperModel
    from .arch import ModelCfg, WhisperCfg, ClipCfg, LlamaCfg, LoraCfg

    oc = Wt.tiny()
    W = Wt.all_weights(oc, 0, lora_b_std=0.05)
    audio, video, labels, prompt = Wt.synthetic_batch(oc, 2, 5, seed=7)
    cfg = ModelCfg(WhisperCfg(**vars(oc.whisper)), ClipCfg(**vars(oc.clip)), LlamaCfg(**vars(oc.llama)), LoraCfg(oc.lora.r, oc.lora.alpha))
    m = ClipWhisperModel(device="cuda:0", use_fp16=False, lora_r=oc.lora.r, lora_alpha=oc.lora.alpha, lora_dropout=0.0, max_seq_len=512,
                         config=cfg, weights=W, precision="fp32")
    m.train()
    out = m(audio=audio.cuda(), video=video.cuda(), prompt=prompt.cuda(), labels=labels.cuda())
    out["loss"].backward()
    torch.cuda.synchronize()
    loss, logits, grads = O.train_step_grads(W, oc, audio, video, prompt, labels)
    dl = (out["logits"].float().cpu() - logits).abs().max().item()
    assert dl < 1e-3, f"smoke: logits differ from the oracle by {dl}"
    assert abs(float(out["loss"].detach()) - float(loss)) < 1e-4
    gv = m.llm_engine.lora_views(m.lora_param.grad)
    for k, g in grads.items():
        d = (gv[k].cpu() - g).abs().max().item()
        assert d <= 1e-4 * max(1.0, g.abs().max().item()), (k, d)
    print(f"smoke OK: loss {float(out['loss']):.6f} (oracle {float(loss):.6f}), max|dlogits| {dl:.2e}")
