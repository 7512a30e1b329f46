"""CPU: the oracle (oracle/avsr_oracle.py) against the golden vectors generated from the REFERENCE's own
encode/forward/backward/generate (oracle/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import avsr_oracle as O
from oracle import weights as Wt


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_glue_golden(golden_dir):
    g = np.load(f"{golden_dir}/g1_glue.npz")
    x = T(g["pt_in"])
    for t in (20, 37, 50):
        assert torch.equal(O.pad_or_truncate(x, t), T(g[f"pt_{t}"]))
    for L, t in ((544, 256), (1532, 256), (300, 256), (257, 256)):
        assert (O.adaptive_projection(T(g[f"pool_{L}_{t}_in"]), t, True) - T(g[f"pool_{L}_{t}"])).abs().max() < 1e-6
    for L, t in ((100, 256), (33, 256)):
        xin = T(g[f"interp_{L}_{t}_in"])
        assert (O.adaptive_projection(xin, t, True) - T(g[f"interp_train_{L}_{t}"])).abs().max() < 2e-6
        assert (O.adaptive_projection(xin, t, False) - T(g[f"interp_eval_{L}_{t}"])).abs().max() < 2e-6
    m = torch.ones(2, 40, dtype=torch.long)
    for t in (30, 40, 64):
        assert torch.equal(O.adapt_mask(m, t), T(g[f"mask_{t}"]))


@pytest.fixture(scope="module")
def tiny(golden_dir):
    g = np.load(f"{golden_dir}/g2_tiny_e2e.npz")
    cfg = Wt.tiny()
    W = Wt.all_weights(cfg, int(g["seed"]), lora_b_std=0.05)
    audio, video, labels, _ = Wt.synthetic_batch(cfg, 2, int(g["frames"]), seed=int(g["batch_seed"]))
    assert torch.equal(labels, T(g["labels"]))
    return g, cfg, W, audio, video, labels, T(g["prompt"])


def test_encode_golden(tiny):
    g, cfg, W, audio, video, labels, prompt = tiny
    with torch.no_grad():
        enc, mask = O.encode(W, cfg, audio, video, None)
        assert enc.shape == (2, 512, cfg.llama.hidden) and mask.dtype == torch.long and mask.all()
        assert (enc[:, ::8] - T(g["encode_av_rows"])).abs().max() < 2e-4
        assert (O.encode(W, cfg, audio, None, None)[0][:, ::32] - T(g["encode_a_rows"])).abs().max() < 2e-4
        assert (O.encode(W, cfg, None, video, None)[0] - T(g["encode_v"])).abs().max() < 2e-4


def test_train_step_golden(tiny):
    g, cfg, W, audio, video, labels, prompt = tiny
    loss, logits, grads = O.train_step_grads(W, cfg, audio, video, prompt, labels)
    assert logits.shape == (2, 256, cfg.llama.vocab)
    assert abs(float(loss) - float(g["train_loss"])) < 1e-5
    assert (logits - T(g["train_logits"])).abs().max() < 5e-4
    assert bool(g["connector_grad_is_none"])          # frozen encoders => connectors get no grad
    for k, v in grads.items():
        ref = T(g["grad." + k])
        assert (v - ref).abs().max() <= 5e-5 * max(1.0, float(ref.abs().max())), k


def test_eval_and_generate_golden(tiny):
    g, cfg, W, audio, video, labels, prompt = tiny
    with torch.no_grad():
        out = O.forward(W, cfg, audio, video, prompt, labels, training=False)
    assert out["logits"].shape == (2, 32 + 512, cfg.llama.vocab)
    assert abs(float(out["loss"]) - float(g["eval_loss"])) < 1e-5
    assert (out["logits"][:, ::4] - T(g["eval_logits_rows"])).abs().max() < 5e-4
    cfg256 = Wt.tiny()
    cfg256.max_seq_len = 256
    ids = O.generate(W, cfg256, audio, video, None, max_new_tokens=12, eos_token_id=2)
    assert torch.equal(ids, T(g["generate_ids"]))


def test_optimizer_golden(golden_dir):
    g = np.load(f"{golden_dir}/g5_optimizer.npz")
    p, m, v = T(g["p0"]).clone(), torch.zeros(1000), torch.zeros(1000)
    for s in range(3):
        grad = T(g[f"g{s}"]).clone()
        n = O.clip_grad_norm_([grad], 0.5)
        assert abs(float(n) - float(g[f"norm{s}"])) < 1e-4 * float(g[f"norm{s}"])
        lr = O.cosine_lr(5e-5, s, 10)
        assert abs(lr - float(g[f"lr{s}"])) < 1e-12
        O.adamw_step(p, grad, m, v, s + 1, lr)
        assert (p - T(g[f"p{s + 1}"])).abs().max() < 1e-7


def test_error_behaviour():
    """Shape guards of encode_audio / encode_video / encode (clip_whisper_model.py:1074-1075, :1115-1116, :445)."""
    cfg = Wt.tiny()
    W = {"whisper": {}, "clip": {}, "llama": {}}
    with pytest.raises(ValueError):
        O.encode(W, cfg, torch.zeros(2, 128, 3000), None, None)
    with pytest.raises(ValueError):
        O.encode(W, cfg, None, torch.zeros(2, 7, 1, 48, 48), None)
    with pytest.raises(ValueError):
        O.encode(W, cfg, None, None, None)


def test_wer_known_answers():
    """jiwer.wer semantics (decode.py:30-37): corpus-level word Levenshtein, no normalisation. Hand-computed."""
    assert O.wer("a b c d", "a x c") == pytest.approx(2 / 4)
    assert O.wer("hello world", "hello world") == 0.0
    assert O.wer(["a b", "c d e"], ["a", "c x e f"]) == pytest.approx((1 + 2) / 5)
    assert O.wer("a", "") == 1.0
    assert O.wer("the cat", "The cat") == pytest.approx(0.5)      # case-sensitive
    assert O.wer("a b c", "b c a d") == pytest.approx(3 / 3)


def test_grouped_query_golden(golden_dir):
    """G7: the reference with num_key_value_heads=2 < num_attention_heads=4 (Llama-3 / Mistral layout)."""
    from oracle.make_golden import gqa_cfg
    g = np.load(f"{golden_dir}/g7_tiny_gqa.npz")
    cfg = gqa_cfg()
    W = Wt.all_weights(cfg, int(g["seed"]), lora_b_std=0.05)
    assert W["llama"]["model.layers.0.self_attn.k_proj.weight"].shape == (128, 256)
    assert W["lora"]["layers.0.v_proj.lora_B"].shape == (128, cfg.lora.r)
    audio, video, labels, _ = Wt.synthetic_batch(cfg, 2, int(g["frames"]), seed=int(g["batch_seed"]))
    prompt = torch.from_numpy(g["prompt"])
    loss, logits, grads = O.train_step_grads(W, cfg, audio, video, prompt, labels)
    assert (logits - torch.from_numpy(g["train_logits"])).abs().max() < 5e-4
    assert abs(float(loss) - float(g["train_loss"])) < 1e-5
    for k, gr in grads.items():
        ref = torch.from_numpy(g["grad." + k])
        assert (gr - ref).abs().max() <= 5e-5 * max(1.0, float(ref.abs().max())), k
    cfg.max_seq_len = 256
    ids = O.generate(W, cfg, audio, video, None, max_new_tokens=10, eos_token_id=2)
    assert torch.equal(ids, torch.from_numpy(g["generate_ids"]))


def test_mxfp8_restatement_matches_the_mx_rule():
    """oracle/mxfp8.py against the OCP MX definition spelled out independently (numpy, element by element on small blocks): shared
    exponent floor(log2(amax)) - 8, e4m3fn grid with round-to-nearest-even, saturation at 448, zero blocks."""
    import numpy as np
    from oracle import mxfp8 as MX
    rng = np.random.default_rng(0)
    x = rng.standard_normal((6, 64)).astype(np.float32)
    x[0, :32] = 0
    x[1, 5] = 1000.0
    x[2, 32:] *= 1e-5
    x[3, 7] = 510.0 * 2.0 ** 5        # scaled value 510 > 448 -> saturates
    codes, e = MX.quantize(torch.from_numpy(x))
    deq = MX.dequantize(codes, e).numpy()
    grid = sorted({(1 + m / 8) * 2.0 ** ex for ex in range(-6, 9) for m in range(8)} | {m / 8 * 2.0 ** -6 for m in range(8)})
    grid = np.array([g for g in grid if g <= 448.0])
    for r in range(6):
        for b in range(2):
            blk = x[r, 32 * b:32 * b + 32]
            amax = np.abs(blk).max()
            ee = -127 if amax == 0 else int(np.floor(np.log2(amax))) - 8
            assert int(e[r, b]) == max(-127, ee)
            for v, d in zip(blk, deq[r, 32 * b:32 * b + 32]):
                t = min(abs(v) / 2.0 ** ee, 448.0) if amax > 0 else 0.0
                j = np.searchsorted(grid, t)
                cands = [grid[max(j - 1, 0)], grid[min(j, len(grid) - 1)]]
                best = min(cands, key=lambda g: abs(g - t))
                if abs(cands[0] - t) == abs(cands[1] - t):                      # tie: even mantissa
                    best = cands[0] if (np.frexp(cands[0])[0] * 16) % 2 == 0 else cands[1]
                assert abs(abs(d) - best * 2.0 ** ee) <= 1e-12 * max(1.0, abs(d)), (r, b, v, d, best * 2.0 ** ee)


def test_deep_connector_golden(golden_dir):
    """oracle connector() for the reference's DeepModalityConnector against tests/golden/g9_deep_connector.npz (oracle/make_golden_connector.py
    ran the reference's own class, 2- and 4-layer, the latter requested through the factory's unknown-name fallback)."""
    g = np.load(f"{golden_dir}/g9_deep_connector.npz")
    for tag in ("a", "b"):
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".sd.")}
        y = O.connector(sd, torch.from_numpy(g[tag + ".x"]))
        assert (y - torch.from_numpy(g[tag + ".y"])).abs().max() < 2e-6


def test_connector_restatements_match_reference_fixture(golden_dir):
    """oracle conv / attention / adaptive connectors == the reference's own modules (fixture g10, eval mode)."""
    g = np.load(f"{golden_dir}/g10_connectors.npz")
    for tag, fn in (("conv", O.connector_conv), ("attn", O.connector_attention), ("adapt_short", O.connector_adaptive), ("adapt_long", O.connector_adaptive)):
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + ".sd.")}
        y = fn(sd, torch.from_numpy(g[tag + ".x"]))
        assert y.shape == g[tag + ".y"].shape and (y - torch.from_numpy(g[tag + ".y"])).abs().max() < 5e-5, tag


def test_llama3_rope_scaling_matches_transformers():
    """The oracle's "llama3" RoPE frequency rule (Llama-3.1 / 3.2: the reference decode.py's default LLM is checkpoints/Llama-3.2-1B) against
    transformers' own ROPE_INIT_FUNCTIONS["llama3"] on Llama-3.2-1B's published rope_scaling values."""
    from transformers import LlamaConfig
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    rs = {"rope_type": "llama3", "factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0, "original_max_position_embeddings": 8192}
    try:
        cfg = LlamaConfig(hidden_size=2048, num_attention_heads=32, num_hidden_layers=1, intermediate_size=64, vocab_size=64, rope_theta=500000.0,
                          rope_scaling=dict(rs), max_position_embeddings=131072)
    except TypeError:
        cfg = LlamaConfig(hidden_size=2048, num_attention_heads=32, num_hidden_layers=1, intermediate_size=64, vocab_size=64,
                          rope_parameters=dict(rs, rope_theta=500000.0), max_position_embeddings=131072)
    inv_hf, att = ROPE_INIT_FUNCTIONS["llama3"](cfg, "cpu")
    assert att == 1.0
    pos = torch.tensor([0, 1, 77, 255, 5000])
    cos, sin = O.rope_cos_sin(pos, 64, 500000.0, (32.0, 1.0, 4.0, 8192))
    fr = pos.float()[:, None] * inv_hf.float()[None, :]
    ref = torch.cat([fr, fr], -1)
    assert (cos - ref.cos()).abs().max() < 1e-5 and (sin - ref.sin()).abs().max() < 1e-5
    plain_cos, _ = O.rope_cos_sin(pos, 64, 500000.0)
    assert (cos - plain_cos).abs().max() > 0.1                    # the rule is not a no-op on these positions
