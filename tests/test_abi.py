"""CPU: libavllm.so builds for gfx950, loads, and exports every symbol include/avllm.h declares (no compute)."""
import ctypes
import os
import re

from avllm import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "avllm.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(avllm_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_header():
    import __graft_entry__ as ge
    ge.build()
    assert os.path.exists(L.LIB_PATH)
    so = ctypes.CDLL(L.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(so, s)]
    assert not missing, missing
    # the ctypes table binds exactly the declared functions
    assert set(L.EXPORTS) == set(syms), set(L.EXPORTS) ^ set(syms)
    lib = L.load()
    assert lib.avllm_version() >= 100


def test_struct_layouts_match_header():
    """sizeof() and selected offsetof() of the ctypes mirrors vs the C structs, via a tiny probe compiled with gcc."""
    import subprocess, tempfile
    prog = r'''
#include <stdio.h>
#include <stddef.h>
#include "avllm.h"
int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(avllm_gemm_desc), sizeof(avllm_enc_layer), sizeof(avllm_whisper),
 sizeof(avllm_clip), sizeof(avllm_lora_mod), sizeof(avllm_llama_layer), sizeof(avllm_llama), sizeof(avllm_gemm_f8_desc), sizeof(avllm_dec_proj_desc),
 sizeof(avllm_step_state), sizeof(avllm_schedule));
 printf("%zu %zu %zu %zu %zu %zu\n", offsetof(avllm_gemm_f8_desc, Cq), offsetof(avllm_gemm_f8_desc, ldcq), offsetof(avllm_dec_proj_desc, rope),
 offsetof(avllm_dec_proj_desc, pos_dev), offsetof(avllm_whisper, fp8), offsetof(avllm_step_state, skipped)); return 0;}'''
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "p.c")
        open(c, "w").write(prog)
        exe = os.path.join(td, "p")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        out = subprocess.check_output([exe]).decode().split("\n")
        sizes = [int(x) for x in out[0].split()]
        offs = [int(x) for x in out[1].split()]
    mine = [ctypes.sizeof(t) for t in (L.GemmDesc, L.EncLayer, L.Whisper, L.Clip, L.LoraMod, L.LlamaLayer, L.Llama, L.GemmF8Desc, L.DecProjDesc,
                                       L.StepState, L.Schedule)]
    assert sizes == mine, (sizes, mine)
    mine_off = [L.GemmF8Desc.Cq.offset, L.GemmF8Desc.ldcq.offset, L.DecProjDesc.rope.offset, L.DecProjDesc.pos_dev.offset, L.Whisper.fp8.offset,
                L.StepState.skipped.offset]
    assert offs == mine_off, (offs, mine_off)


def test_missing_library_is_loud(monkeypatch, tmp_path):
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        L.load()
