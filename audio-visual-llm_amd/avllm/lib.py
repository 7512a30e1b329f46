"""ctypes binding of libavllm.so (include/avllm.h).

The product path has NO fallback: if the HIP library is missing or a call fails this module raises.
PyTorch is used only as the owner of device memory and streams; every arithmetic op on the hot path is a
call into the C ABI below.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVLLM_LIB_PATH") or os.path.join(_HERE, "libavllm.so")      # override: A/B runs of two builds (tools/)

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_QUICK_GELU, ACT_SILU = 0, 1, 2, 3
LORA_PAD = 64

i32, i64, f32, vp, sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t
fp = C.POINTER(C.c_float)


class GemmDesc(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("A2", vp), ("B2", vp), ("C", vp), ("bias", vp), ("R", vp),
                ("lda", i64), ("ldb", i64), ("lda2", i64), ("ldb2", i64), ("ldc", i64), ("ldr", i64),
                ("M", i32), ("N", i32), ("K", i32), ("K2", i32), ("dtype", i32), ("out_f32", i32), ("act", i32),
                ("alpha", f32), ("r_mod", i32), ("g_in", i32), ("g_out", i32), ("g_off", i32), ("drop_seed", C.c_uint32), ("drop_p", f32), ("a_drop_seed", C.c_uint32), ("a_drop_p", f32), ("n_valid", i32), ("seed_dev", vp)]


class EncLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2",
                                  "wqkv8", "sqkv8", "wo8", "so8", "w18", "s18", "w28", "s28")]


class Whisper(C.Structure):
    _fields_ = [(n, i32) for n in ("dtype", "d", "heads", "layers", "ffn", "n_mels", "n_ctx", "k1pad")] + \
               [(n, vp) for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "pos")] + \
               [("layer", C.POINTER(EncLayer)), ("lnf_w", vp), ("lnf_b", vp), ("fp8", i32)]


class Clip(C.Structure):
    _fields_ = [(n, i32) for n in ("dtype", "d", "heads", "layers", "ffn", "image", "patch", "tokens")] + \
               [("eps", f32)] + [(n, vp) for n in ("patch_w", "class_emb", "pos", "pre_ln_w", "pre_ln_b")] + \
               [("layer", C.POINTER(EncLayer)), ("fp8", i32), ("frames_bf16", i32)]


class LoraMod(C.Structure):
    _fields_ = [("A_pad", vp), ("AT_pad", vp), ("B_pad", vp), ("BT_pad", vp), ("ld_at", i64), ("gA", vp), ("gB", vp)]


class LlamaLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_w", "ln2_w", "wqkv", "wo", "wgu", "wdown", "wqkv_t", "wo_t", "wgu_t", "wdown_t")] + \
               [("lora", LoraMod * 4)] + [(n, vp) for n in ("wqkv8", "sqkv8", "wo8", "so8", "wgu8", "sgu8", "wdown8", "sdown8")]


class Llama(C.Structure):
    _fields_ = [(n, i32) for n in ("dtype", "d", "heads", "layers", "ffn", "vocab", "lora_r", "kv_heads")] + \
               [(n, f32) for n in ("eps", "theta", "lora_scale", "lora_dropout")] + [("dropout_seed", C.c_uint32)] + \
               [(n, vp) for n in ("dropout_seed_dev", "embed", "norm_w", "lm_head", "lm_head_t")] + [("layer", C.POINTER(LlamaLayer))] + \
               [("fp8", i32), ("lm_head8", vp), ("slm_head8", vp)] + \
               [(n, f32) for n in ("rope_factor", "rope_low_freq_factor", "rope_high_freq_factor")] + [("rope_orig_ctx", i32)]


class GemmF8Desc(C.Structure):
    _fields_ = [("A", vp), ("SA", vp), ("B", vp), ("SB", vp), ("C", vp), ("bias", vp), ("R", vp), ("lda", i64), ("ldb", i64), ("ldc", i64),
                ("ldr", i64), ("M", i32), ("N", i32), ("K", i32), ("act", i32), ("Cq", vp), ("SCq", vp), ("ldcq", i64)]


class DecProjDesc(C.Structure):
    """avllm_dec_proj_desc (include/avllm.h): one projection of a decode token step."""
    _fields_ = [("A", vp), ("lda", i64), ("W", vp), ("ldw", i64), ("norm_w", vp), ("eps", f32), ("M", i32), ("K", i32), ("N", i32), ("mode", i32),
                ("C", vp), ("ldc", i64), ("out_f32", i32), ("R", vp), ("ldr", i64), ("dq", i32), ("dkv", i32), ("hd", i32), ("rope", vp),
                ("kc", vp), ("vc", vp), ("Tmax", i32), ("pos", i32), ("pos_dev", vp),
                ("lora_t", vp), ("ld_lora_t", i64), ("lora_b", vp * 3), ("lora_r", i32), ("lora_scale", f32)]


class StepState(C.Structure):
    """avllm_step_state: per-step scalars in DEVICE memory (this mirror is only used for sizes / field offsets / host reads)."""
    _fields_ = [("step", C.c_uint32), ("dropout_seed", C.c_uint32), ("lr", f32), ("bc1", f32), ("bc2_sqrt", f32), ("skipped", f32),
                ("reserved", C.c_uint32 * 2)]


class Schedule(C.Structure):
    _fields_ = [("base_lr", f32), ("beta1", f32), ("beta2", f32), ("warmup_steps", i32), ("total_steps", i32), ("rank", C.c_uint32)]


LAYER_CB = C.CFUNCTYPE(None, i32, vp)

_SIGS = {
    "avllm_version": ([], i32),
    "avllm_gemm": ([C.POINTER(GemmDesc), vp], i32),
    "avllm_set_gemm_variant": ([i32], i32),
    "avllm_set_knob": ([C.c_char_p, i32], i32),
    "avllm_attention_fwd_mxq": ([vp, vp, vp, vp, i64, vp, i32, i32, i32, i32, i64, i64, i64, f32, vp], i32),
    "avllm_im2col_k3": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "avllm_groupnorm_tokens": ([vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, i32, vp], i32),
    "avllm_gemm_tn": ([vp, i64, i32, vp, i64, i32, i32, vp, i64, f32, i32, vp], i32),
    "avllm_gemm_tn_drop": ([vp, i64, i32, vp, i64, i32, i32, vp, i64, f32, C.c_uint32, f32, i32, vp], i32),
    "avllm_logmel_table_bytes": ([], C.c_size_t),
    "avllm_logmel_table_init": ([vp, i32], i32),
    "avllm_logmel_workspace_bytes": ([i32, i32], C.c_size_t),
    "avllm_logmel": ([vp, vp, i32, i32, i64, i32, i32, vp, vp, C.c_size_t, vp], i32),
    "avllm_clip_preproc_plan_bytes": ([i32, i32, i32], C.c_size_t),
    "avllm_clip_preproc_plan_init": ([vp, i32, i32, i32, vp, vp], i32),
    "avllm_clip_preproc_workspace_bytes": ([i32, i32, i32, i32], C.c_size_t),
    "avllm_clip_preproc": ([vp, vp, i32, i32, i32, i32, vp, i32, vp, C.c_size_t, vp], i32),
    "avllm_lora_pack_batch": ([vp, i32, i32, i32, i32, vp], i32),
    "avllm_layernorm": ([vp, vp, vp, vp, i64, i32, f32, i32, vp], i32),
    "avllm_rmsnorm_fwd": ([vp, vp, vp, vp, i64, i32, f32, i32, vp], i32),
    "avllm_rmsnorm_bwd": ([vp, vp, vp, vp, vp, vp, i64, i32, i32, vp], i32),
    "avllm_rope": ([vp, i64, i64, i32, i32, i32, i32, f32, i32, i32, vp], i32),
    "avllm_swiglu_fwd": ([vp, vp, i64, i32, i32, vp], i32),
    "avllm_swiglu_bwd": ([vp, vp, vp, i64, i32, i32, vp], i32),
    "avllm_attention_fwd": ([vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i64, i64, i64, i64, f32, i32, i32, i32, i32, vp], i32),
    "avllm_attention_bwd": ([vp] * 10 + [i32] * 4 + [i64] * 7 + [f32, i32, i32, i32, i32, vp], i32),
    "avllm_ce_fwd": ([vp, i64, vp, i32, i32, i32, vp, vp, vp, i32, vp], i32),
    "avllm_ce_bwd": ([vp, i64, vp, vp, vp, f32, vp, i32, i32, i32, i32, vp], i32),
    "avllm_argmax_rows": ([vp, i64, i64, i32, vp, i32, vp], i32),
    "avllm_embedding": ([vp, vp, vp, i64, i32, i32, vp], i32),
    "avllm_cast": ([vp, i32, vp, i32, i64, vp], i32),
    "avllm_act_residual": ([vp, vp, vp, i64, i32, i32, vp], i32),
    "avllm_dropout": ([vp, vp, i64, i32, C.c_uint32, f32, i32, vp], i32),
    "avllm_whisper_im2col1": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "avllm_whisper_im2col2": ([vp, vp, i32, i32, i32, i32, vp], i32),
    "avllm_clip_patchify": ([vp, vp, i32, i32, i32, i32, i32, vp], i32),
    "avllm_clip_cls_rows": ([vp, vp, vp, i32, i32, i32, i32, vp], i32),
    "avllm_fuse_pool": ([vp, i32, vp, i32, vp, i32, vp, i32, i32, i32, i32, f32, i32, vp], i32),
    "avllm_grad_sumsq": ([vp, i64, vp, vp], i32),
    "avllm_grad_sumsq_det": ([vp, i64, vp, i32, vp, vp], i32),
    "avllm_adamw_step": ([vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i32, vp, f32, f32, vp, vp, vp, vp], i32),
    "avllm_step_advance": ([vp, C.POINTER(Schedule), vp], i32),
    "avllm_mx_scale_bytes": ([i32, i32], sz),
    "avllm_mx_quantize": ([vp, i64, i32, i32, vp, i64, vp, i32, i32, vp], i32),
    "avllm_gemm_f8": ([C.POINTER(GemmF8Desc), vp], i32),
    "avllm_lora_dx_masked": ([C.POINTER(vp), C.POINTER(i64), C.POINTER(vp), C.POINTER(i64), C.POINTER(C.c_uint32), i32, i32, vp, i64, vp, i64, i32,
                              i32, f32, vp, i32, vp], i32),
    "avllm_llama_lora_bwd_layers": ([C.POINTER(Llama), vp, i32, i32, vp, f32, vp, sz, i32, i32, LAYER_CB, vp, vp], i32),
    "avllm_lora_pack": ([vp, vp, i32, i32, i32, vp, vp, i64, vp, vp, i32, vp], i32),
    "avllm_profile_begin": ([i32], i32),
    "avllm_profile_enable": ([i32], i32),
    "avllm_profile_end": ([C.POINTER(C.c_double)], i32),
    "avllm_whisper_workspace_bytes": ([C.POINTER(Whisper), i32], sz),
    "avllm_whisper_encoder_fwd": ([C.POINTER(Whisper), vp, i32, vp, vp, sz, vp], i32),
    "avllm_clip_workspace_bytes": ([C.POINTER(Clip), i32], sz),
    "avllm_clip_vision_cls_fwd": ([C.POINTER(Clip), vp, i32, vp, vp, sz, vp], i32),
    "avllm_llama_train_workspace_bytes": ([C.POINTER(Llama), i32, i32], sz),
    "avllm_llama_lora_fwd_loss": ([C.POINTER(Llama), vp, vp, i32, i32, vp, vp, vp, vp, sz, vp], i32),
    "avllm_llama_lora_bwd": ([C.POINTER(Llama), vp, i32, i32, vp, f32, vp, sz, LAYER_CB, vp, vp], i32),
    "avllm_llama_infer_workspace_bytes": ([C.POINTER(Llama), i32, i32], sz),
    "avllm_llama_prefill": ([C.POINTER(Llama), vp, i32, i32, vp, vp, i32, vp, vp, vp, sz, vp], i32),
    "avllm_llama_decode_step": ([C.POINTER(Llama), vp, i32, i32, vp, vp, i32, vp, vp, sz, vp], i32),
    "avllm_llama_decode_step_at": ([C.POINTER(Llama), vp, i32, i32, vp, vp, vp, i32, vp, vp, sz, vp], i32),
    "avllm_llama_decode_is_fused": ([C.POINTER(Llama), i32], i32),
    "avllm_pos_advance": ([vp, i32, vp], i32),
    "avllm_dec_proj": ([C.POINTER(DecProjDesc), vp], i32),
    "avllm_gemm_f8_takes_quantised_output": ([C.POINTER(GemmF8Desc)], i32),
    "avllm_norm_mxq": ([vp, vp, vp, vp, vp, vp, i64, vp, i64, i32, f32, vp], i32),
    "avllm_lora_rank3": ([vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, i32, vp], i32),
    "avllm_gemm_tn_multi": ([vp, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, f32, vp, i32, i32, vp], i32),
    "avllm_attention_decode": ([vp, i64, vp, vp, vp, i64, i32, i32, i32, i32, vp, i32, f32, i32, i32, vp], i32),
}

EXPORTS = sorted(list(_SIGS) + ["avllm_last_error"])

_lib = None


def load():
    """Load libavllm.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"libavllm.so not found at {LIB_PATH}: run `python build.py` (hipcc, gfx950). "
                           "There is no CPU or PyTorch fallback for the hot path.")
    lib = C.CDLL(LIB_PATH)
    lib.avllm_last_error.restype = C.c_char_p
    lib.avllm_last_error.argtypes = []
    for name, (args, res) in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    return lib


class AvllmError(RuntimeError):
    pass


class knob:
    """with knob("DECODE_FUSED", 0): ...   -- one of the library's A/B switches for the duration of the block (include/avllm.h
    avllm_set_knob; the table is otherwise filled once per process from AVLLM_<NAME>)."""
    _defaults = {"DECODE_FUSED": 1, "DEC_AL": 0, "LORA_UNBATCHED": 0, "F8_UNFUSED_QUANT": 0, "F8_FAST": 1, "ATTN_SHORT": 1,
                 "NARROW_EPILOGUE": 0, "TN_CHUNK": 0, "GEMM_DBG": 0, "GEMM_GW": 0}

    def __init__(self, name, value):
        self.name, self.value = name, int(value)

    def __enter__(self):
        check(load().avllm_set_knob(self.name.encode(), self.value))
        return self

    def __exit__(self, *a):
        env = os.environ.get("AVLLM_" + self.name)
        check(load().avllm_set_knob(self.name.encode(), int(env) if env is not None else self._defaults[self.name]))


def check(rc: int):
    if rc != 0:
        msg = load().avllm_last_error().decode(errors="replace")
        if rc == 1:
            raise ValueError(msg)
        raise AvllmError(f"libavllm error {rc}: {msg}")


def dt_of(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise TypeError(f"unsupported dtype {t.dtype}")


def torch_dtype(dt: int):
    return torch.float32 if dt == F32 else torch.bfloat16


def ptr(t):
    if t is None:
        return None
    assert t.is_cuda, "device tensor required"
    return t.data_ptr()


def stream_ptr():
    return torch.cuda.current_stream().cuda_stream
