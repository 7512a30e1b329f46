#!/usr/bin/env python3
"""Build libavllm.so (HIP, gfx950 only) and the oracle's C helpers in-tree.  Called by __graft_entry__.build()."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(ROOT, "audio-visual-llm_amd", "csrc")
OUT = os.path.join(ROOT, "audio-visual-llm_amd", "avllm", "libavllm.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-I", os.path.join(ROOT, "include")]
FLAGS += os.environ.get("AVLLM_EXTRA_FLAGS", "").split()        # e.g. -DAVLLM_EXPERIMENT_KNOBS for tools/gemm_epi_experiment.sh (use --force)


def stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "avllm.h")]
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        if force or stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        r = subprocess.run([HIPCC] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    failed = False
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for src, rc, out in ex.map(cc, jobs):
            if verbose and (rc or out.strip()):
                print(f"[{os.path.basename(src)}] rc={rc}\n{out}")
            failed |= rc != 0
    if failed:
        raise RuntimeError("hipcc failed")
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in srcs]
    if force or jobs or not os.path.exists(OUT):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs, capture_output=True, text=True)
        if r.returncode:
            print(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    if verbose:
        print("built", OUT, f"({len(jobs)} objects recompiled)")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
