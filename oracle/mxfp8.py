"""CPU restatement of the block-scaled fp8 arithmetic of BASELINE config 5 (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

The reference has no fp8 mode (src/clip_whisper/models/clip_whisper_model.py:164 offers only `use_fp16`), so there is nothing of the
reference's to restate line by line; what is restated here is the published OCP Microscaling (MX) v1.0 rule the HIP path implements
(csrc/fp8.hip): blocks of 32 consecutive elements along K share one E8M0 scale 2^e with e = floor(log2(amax)) - 8 (8 = emax of
e4m3), elements are round-to-nearest-even e4m3fn values of x * 2^-e, saturated to +-448.  "Parity unpinned" against any reference
output; pinned against torch's own float8_e4m3fn rounding (test_oracle.py) and bit-compared with the kernel's codes on the GPU.

`fake_quant(x)` returns the fp32 values the matrix pipe effectively multiplies, so  fake_quant(x) @ fake_quant(w).T  in fp32 is the
oracle for avllm_gemm_f8 and, wired into avsr_oracle through `linear=`, for the fp8 model mode.
"""
from __future__ import annotations

import torch

BLOCK = 32
E4M3_MAX = 448.0


def block_exponents(x: torch.Tensor) -> torch.Tensor:
    """x [..., K] (K % 32 == 0) -> int32 exponents e [..., K/32]: floor(log2(amax)) - 8, clamped to [-127, 127]; amax == 0 -> -127."""
    xb = x.float().reshape(*x.shape[:-1], x.shape[-1] // BLOCK, BLOCK)
    amax = xb.abs().amax(-1)
    bits = amax.view(torch.int32)
    e = ((bits >> 23) & 0xFF) - 127 - 8            # exponent field: floor(log2) for normal numbers; 0/subnormal -> -135 -> clamped
    return e.clamp(-127, 127).to(torch.int32)


def quantize(x: torch.Tensor):
    """-> (codes uint8 [..., K] e4m3fn bit patterns, exponents int32 [..., K/32])."""
    e = block_exponents(x)
    scale = torch.ldexp(torch.ones((), dtype=torch.float32), -e)              # 2^-e
    xb = x.float().reshape(*x.shape[:-1], x.shape[-1] // BLOCK, BLOCK) * scale.unsqueeze(-1)
    q = xb.clamp(-E4M3_MAX, E4M3_MAX).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(x.shape), e


def dequantize(codes: torch.Tensor, e: torch.Tensor) -> torch.Tensor:
    v = codes.view(torch.float8_e4m3fn).float().reshape(*codes.shape[:-1], codes.shape[-1] // BLOCK, BLOCK)
    return (v * torch.ldexp(torch.ones((), dtype=torch.float32), e).unsqueeze(-1)).reshape(codes.shape)


def fake_quant(x: torch.Tensor) -> torch.Tensor:
    return dequantize(*quantize(x))


def linear_fp8(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """nn.Linear's x W^T as the fp8 path computes it: both operands block-quantised along K, products and sums in fp32.
    The activation is first rounded to bf16 (what the quantiser reads in the HIP path)."""
    return fake_quant(x.to(torch.bfloat16).float()) @ fake_quant(w.to(torch.bfloat16).float()).T


def scale_image_index(layout: int, R: int, K: int):
    """(row, kblock) -> (word index, byte) of the kernel's scale image (csrc/fp8.hip "Formats"); returns a dict-free pair of tensors
    usable to gather the image into a [R, K/32] exponent matrix."""
    RB = (R + 255) // 256 * 4
    rows = torch.arange(R)
    if layout == 0:
        rb, i, fr = rows // 64, (rows // 16) % 4, rows % 16
    else:
        cb, w = rows // 128, rows % 128
        p, e_, fr_hi, fr_lo = w // 32, (w // 4) % 2, (w // 8) % 4, w % 4
        j = 2 * p + e_
        rb, i, fr = 2 * cb + j // 4, j % 4, 4 * fr_hi + fr_lo
    kb = torch.arange(K // 32)
    word = ((kb[None, :] // 4 * RB + rb[:, None]) * 4 + kb[None, :] % 4) * 16 + fr[:, None]
    return word, i[:, None].expand_as(word)
