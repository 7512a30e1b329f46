#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: per-kernel HBM-side bytes per launch, corrected as
/opt/skills/guides/MI355X_MICROARCH.md §HBM prescribes (counter unit = KiB; on gfx950 FETCH_SIZE reports half of a wide
coalesced read stream -> x2; WRITE_SIZE exact).  Usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import collections, csv, re, sys
csv.field_size_limit(1 << 30)

def short(n):
    m = re.search(r"(attn_\w+|gemm_\w+|norm_fwd_kernel|rmsnorm_bwd_kernel|rope\w*kernel|swiglu_\w+kernel|ce_\w+kernel|lora_pack_kernel|"
                  r"fuse_pool_kernel|im2col\d_kernel|patchify\w*kernel|adamw_kernel|sumsq_kernel)", n)
    return m.group(1) if m else None

def load(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    return acc

f, w = load(sys.argv[1]), load(sys.argv[2])
print(f"{'kernel':34s} {'launches':>8s} {'fetch MiB/launch (x2 corrected)':>32s} {'write MiB/launch':>18s} {'total GiB (all launches)':>26s}")
for k in sorted(f, key=lambda k: -(2 * f[k][1] + w.get(k, [0, 0])[1])):
    n = f[k][0]
    fb = 2 * f[k][1] * 1024 / n
    wb = w.get(k, [1, 0.0])[1] * 1024 / max(1, w.get(k, [1, 0])[0])
    print(f"{k:34s} {n:8d} {fb / 2**20:32.2f} {wb / 2**20:18.2f} {(fb + wb) * n / 2**30:26.2f}")
