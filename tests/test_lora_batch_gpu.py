"""The q / k / v adapters batched (csrc/lora_batch.hip) against the per-adapter kernels they replace and against fp32 torch:
forward rank-side products through three dropout masks of one shared input, backward dt (one launch, three inputs), dB over the column ranges of
one dqkv buffer, dA over one shared masked input.  peft semantics: lora_B(lora_A(dropout(x))) per wrapped module (clip_whisper_model.py:961-1005)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from avllm import ops  # noqa: E402
from bars import rel_l2  # noqa: E402
from test_ops_gpu import rnd  # noqa: E402

BF = torch.bfloat16


@pytest.mark.parametrize("M,K,p", [(4096, 4096, 0.05), (300, 512, 0.0), (50, 256, 0.1)])
def test_rank3_shared_input_three_masks(dev, M, K, p):
    x = rnd(M, K, dtype=BF, seed=1)
    As = [rnd(64, K, dtype=BF, seed=10 + j, scale=K ** -0.5) for j in range(3)]
    for A in As:
        A[16:] = 0                                            # rank 16 padded to 64 rows (avllm_lora_pack's image)
    seeds = [1000, 1001, 1002]
    t = torch.full((M, 192), 7.0, device=dev, dtype=BF)       # [t_q | t_k | t_v], as the engine lays it out
    outs = [t[:, 64 * j:64 * j + 64] for j in range(3)]
    ops.lora_rank3([x], As, outs, 16, alpha=2.0, seeds=seeds, p=p, shared=True)
    for j in range(3):
        one = ops.gemm(x, As[j], alpha=2.0, a_drop=(seeds[j], p) if p > 0 else None, n_valid=16)      # the per-adapter kernel
        assert torch.equal(outs[j], one), j                   # same products, same order, same reduction
        xd = ops.dropout(x, seeds[j], p).float() if p > 0 else x.float()
        assert rel_l2(outs[j][:, :16], 2.0 * xd @ As[j][:16].float().t()) < 6e-3
        assert (outs[j][:, 16:] == 0).all()


@pytest.mark.parametrize("M,widths", [(4096, (4096, 4096, 4096)), (300, (512, 256, 256))])
def test_rank3_three_inputs_backward_dt(dev, M, widths):
    """dt_j = s * dy_j . B_j with dy_j the column slices of one dqkv buffer (grouped-query: k / v narrower)."""
    qw = sum(widths)
    dqkv = rnd(M, qw, dtype=BF, seed=2)
    offs = [0, widths[0], widths[0] + widths[1]]
    BTs = []
    for j, wj in enumerate(widths):
        b = rnd(64, wj, dtype=BF, seed=20 + j, scale=wj ** -0.5)
        b[16:] = 0
        BTs.append(b)
    dt = torch.zeros(M, 192, device=dev, dtype=BF)
    outs = [dt[:, 64 * j:64 * j + 64] for j in range(3)]
    dys = [dqkv[:, offs[j]:offs[j] + widths[j]] for j in range(3)]
    ops.lora_rank3(dys, BTs, outs, 16, alpha=0.5)
    for j in range(3):
        assert torch.equal(outs[j], ops.gemm(dys[j], BTs[j], alpha=0.5, n_valid=16)), j
        assert rel_l2(outs[j][:, :16], 0.5 * dys[j].float() @ BTs[j][:16].float().t()) < 6e-3


@pytest.mark.parametrize("M,widths,r", [(4096, (4096, 4096, 4096), 16), (700, (512, 256, 256), 8)])
def test_gemm_tn_multi_dB_over_column_ranges(dev, M, widths, r):
    qw = sum(widths)
    dqkv = rnd(M, qw, dtype=BF, seed=3)
    t = rnd(M, 192, dtype=BF, seed=4)
    cols, off = [], 0
    for wj in widths:
        cols.append((off, wj)); off += wj
    outs = [torch.ones(wj, r, device=dev, dtype=torch.float32) for wj in widths]             # accumulates on top of what is there
    ops.gemm_tn_multi(dqkv, [t[:, 64 * j:64 * j + 64] for j in range(3)], outs, r, cols=cols)
    for j, (c0, wj) in enumerate(cols):
        ref = 1.0 + dqkv[:, c0:c0 + wj].float().t() @ t[:, 64 * j:64 * j + r].float()
        assert rel_l2(outs[j], ref) < 1e-4, j                # bf16 products, fp32 sums: only the order of the atomic adds differs
        one = torch.ones(wj, r, device=dev, dtype=torch.float32)
        ops.gemm_tn(dqkv[:, c0:c0 + wj], t[:, 64 * j:64 * j + 64], one, I=wj, J=r)
        assert rel_l2(outs[j], one) < 1e-5


@pytest.mark.parametrize("M,d,p", [(4096, 4096, 0.05), (333, 512, 0.0)])
def test_gemm_tn_multi_dA_shared_masked_input(dev, M, d, p):
    x = rnd(M, d, dtype=BF, seed=5)
    dt = rnd(M, 192, dtype=BF, seed=6)
    seeds = [77, 78, 79]
    outs = [torch.zeros(16, d, device=dev, dtype=torch.float32) for _ in range(3)]
    ops.gemm_tn_multi(x, [dt[:, 64 * j:64 * j + 64] for j in range(3)], outs, 16, seeds=seeds, p=p, shared=True)
    for j in range(3):
        xd = ops.dropout(x, seeds[j], p).float() if p > 0 else x.float()
        assert rel_l2(outs[j], dt[:, 64 * j:64 * j + 16].float().t() @ xd) < 1e-4, j
        one = torch.zeros(16, d, device=dev, dtype=torch.float32)
        ops.gemm_tn(dt[:, 64 * j:64 * j + 64], x, one, I=16, J=d, drop=(seeds[j], p) if p > 0 else None)
        assert rel_l2(outs[j], one) < 1e-5


def test_batched_and_unbatched_steps_agree(dev, monkeypatch):
    """One bf16 training forward + backward of a 2-layer model with LoRA dropout: loss and every LoRA gradient with the batched adapter
    kernels against AVLLM_LORA_UNBATCHED=1 (the nine-launch form): same masks, same products."""
    from avllm.arch import ClipCfg, LlamaCfg, LoraCfg, ModelCfg, WhisperCfg
    from avllm.model import ClipWhisperModel
    cfg = ModelCfg(WhisperCfg(128, 2, 2, 256), ClipCfg(128, 2, 2, 256, 48, 16), LlamaCfg(256, 2, 2, 512, 256), LoraCfg(16, 32.0))
    g = torch.Generator(device=dev).manual_seed(5)
    audio = torch.randn(2, 80, 3000, generator=g, device=dev)
    video = torch.randn(2, 7, 3, 48, 48, generator=g, device=dev)
    labels = torch.randint(3, 256, (2, 40), generator=g, device=dev)
    res = {}
    from avllm import lib as Lk
    for mode in ("batched", "unbatched"):
        Lk.check(Lk.load().avllm_set_knob(b"LORA_UNBATCHED", 1 if mode == "unbatched" else 0))
        m = ClipWhisperModel(device=dev, max_seq_len=64, config=cfg, precision="bf16", seed=3, synthetic_weights=True, lora_dropout=0.1).train()
        eng = m.llm_engine
        eng.lora_p.normal_(0, 0.02, generator=g.manual_seed(9))
        eng.pack_lora()
        out = m(audio=audio, video=video, labels=labels)
        out["loss"].backward()
        res[mode] = (float(out["loss"].detach()), eng.lora_g.clone())
    Lk.check(Lk.load().avllm_set_knob(b"LORA_UNBATCHED", 0))
    # the loss is a float atomicAdd over ~80 rows (sum ~ 480, fp32 ulp 3e-5): the order of the adds moves the mean by a few 1e-7 from launch to launch
    assert abs(res["batched"][0] - res["unbatched"][0]) < 1e-5
    assert rel_l2(res["batched"][1], res["unbatched"][1]) < 1e-4
    assert float(res["batched"][1].abs().max()) > 0
