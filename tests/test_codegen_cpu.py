"""Code-generation invariants of the hand-scheduled kernels (CPU: hipcc cross-compiles gfx950 here).

The persistent GEMM's K loop is inline assembly with its own s_waitcnt arithmetic; the compiler's wait-count pass does not see those loads
but it does add waits of its own when IT believes a load is pending at the loop (a compiler-visible load consumed under another branch, a
spill reload).  Round 2 measured such a wait at 10 % of every GEMM (vmcnt(0) per K-step) and 28 spilled registers at 25 %: the numbers in
profiles/r02_gemm_epilogue_experiments.txt.  The decode projections' load ring must not be copied by the compiler while its loads are in flight."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _asm(src, tmp_path):
    out = tmp_path / (os.path.basename(src) + ".s")
    r = subprocess.run([CLANG, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-x", "hip", src,
                        "--cuda-device-only", "-S", "-o", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return out.read_text()


def _kernel(asm, pattern):
    names = [n for n in re.findall(r"^(_Z\w+):", asm, flags=re.M) if re.search(pattern, n)]
    assert names, pattern
    return {n: asm.split("\n" + n + ":", 1)[1].split("s_endpgm")[0].split("\n") for n in names}


def _inner_loops(lines, depth=None, mfma=None):
    """[(start, end)] of innermost loops: header comment .. the `mfma`-th MFMA after it, or the first scalar conditional branch."""
    out = []
    for i, l in enumerate(lines):
        if "Inner Loop Header" in l and (depth is None or f"Depth={depth}" in l):
            if mfma:
                seen = 0
                for j in range(i, len(lines)):
                    seen += "v_mfma" in lines[j]
                    if seen == mfma:
                        break
            else:
                j = next(k for k in range(i, len(lines)) if re.search(r"s_cbranch_scc[01]", lines[k]))
            out.append((i, j + 1))
    return out


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang")
def test_persistent_gemm_k_loop_has_no_compiler_waits_or_spills(tmp_path):
    asm = _asm(os.path.join(ROOT, "audio-visual-llm_amd", "csrc", "gemm.hip"), tmp_path)
    for name, lines in _kernel(asm, r"gemm_bf16_wp_kernelILb[01]E").items():
        loops = _inner_loops(lines, depth=2, mfma=128)
        assert len(loops) == 1, (name, loops)
        a, b = loops[0]
        body = lines[a:b]
        assert sum("v_mfma" in l for l in body) == 128, name
        compiler_waits = [l.strip() for k, l in enumerate(body) if "s_waitcnt" in l and "vmcnt" in l and "ASMSTART" not in body[k - 1]]
        assert not compiler_waits, (name, compiler_waits)
        assert not [l for l in body if "scratch_" in l], name
    spills = {re.search(r"\.name:\s+(\S+)", blk).group(1): int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
              for blk in asm[asm.index("amdhsa.kernels:"):].split("  - .agpr_count")[1:]}
    for name, n in spills.items():
        if "gemm_bf16_wp_kernel" in name:
            assert n <= 16, (name, n)          # a handful of scalars around the tile loop; 28+ reach the K loop as waits


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm clang")
def test_decode_ring_registers_are_not_copied_in_flight(tmp_path):
    asm = _asm(os.path.join(ROOT, "audio-visual-llm_amd", "csrc", "decode.hip"), tmp_path)
    kernels = _kernel(asm, r"dec_proj_kernel")
    assert len(kernels) == 12                                     # NORM x 3 activation-load forms x adapters in the epilogue or not
    for name, lines in kernels.items():
        loops = _inner_loops(lines)
        assert loops, name
        for a, b in loops:
            body = lines[a:b]
            assert sum("global_load_dwordx4" in l for l in body) >= 10, name
            bad = []
            for k, l in enumerate(body):
                if re.search(r"scratch_|v_accvgpr", l):
                    bad.append(l.strip())
                m = re.search(r"v_mov_b32_e32 (v\d+), v\d+", l)
                if m:       # benign only as the `old` operand of the DPP move that follows (row rotate / broadcast of an operand AFTER its wait)
                    nxt = next((x for x in body[k + 1:k + 80] if re.search(r"\b" + m.group(1) + r"\b", x)), "")
                    if "_dpp" not in nxt or not re.search(r"v_mov_b32_dpp " + m.group(1) + r",", nxt):
                        bad.append(l.strip())
                if re.search(r"v_mov_b64_e32 v\[\d+:\d+\], v\[", l):
                    bad.append(l.strip())
            assert not bad, (name, bad[:4])
