import sys, os, torch
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "audio-visual-llm_amd"), os.path.join(os.getcwd(), "tests")]
from avllm import ops
from test_ops_gpu import rnd
B, T, H = 5, 197, 12
d = H * 64
qkv = rnd(B * T, 3 * d, dtype=torch.bfloat16, seed=70 + T)
o, _ = ops.attention_fwd(qkv, B, T, H, 64, causal=False, want_lse=False)
o2, _ = ops.attention_fwd(qkv, B, T, H, 64, causal=False, want_lse=False)
print("deterministic:", torch.equal(o, o2))
q_ref, s_ref = ops.mx_quantize(o, 0)
q, s = ops.attention_fwd_mxq(qkv, B, T, H, 64)
bad = (q != q_ref).nonzero()
print("bad codes", bad.shape[0], "bad scale bytes", int((s != s_ref).sum()))
rows = bad[:, 0].unique()
print("rows", rows[:20].tolist(), "n rows", rows.numel())
r, c = bad[0].tolist()
blk = c // 32 * 32
print("row", r, "col", c, "block", blk)
print("o block", o[r, blk:blk + 32].float().tolist())
print("q    ", q[r, blk:blk + 32].tolist())
print("q_ref", q_ref[r, blk:blk + 32].tolist())
cols = bad[:, 1] % 64
print("col%64 histogram", torch.bincount(cols // 8, minlength=8).tolist())
print("row%16 histogram", torch.bincount(bad[:, 0] % 197 % 16, minlength=16).tolist())
