import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "audio-visual-llm_amd")]
import torch
from avllm import ops
from oracle import mxfp8 as MX
W = torch.eye(128).bfloat16().cuda()
for layout in (0, 1):
    q, s = ops.mx_quantize(W, layout)
    img = s.cpu().view(torch.int32).reshape(1, 4, 4, 16)            # [t][rb][fq][fr]
    word, byte = MX.scale_image_index(layout, 128, 128)
    got = ((s.cpu().view(torch.int32)[word.reshape(-1)].reshape(word.shape) >> (8 * byte)) & 0xFF)
    codes, e = MX.quantize(W.float().cpu())
    print("layout", layout, "image matches oracle:", bool(torch.equal(got - 127, e)), " codes match:", bool(torch.equal(q.cpu(), codes)))
    nzw = (img != 0).nonzero()
    print("  nonzero words [t, rb, fq, fr] count", len(nzw), "first:", nzw[:12].tolist())
    print("  row 16: oracle e", e[16].tolist(), " image bytes via index:", (got[16] - 127).tolist(), " word idx", word[16].tolist(), "byte", byte[16].tolist())
