#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace CSV, attributed to the (previous kernel -> next kernel) pair.
Usage: gap_summary.py <kernel_trace.csv> [skip_frac]   (skip_frac drops the warm-up part of the trace)"""
import collections, csv, re, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(attn_\w+|gemm_\w+|norm_fwd_kernel|layernorm768_kernel|rmsnorm_bwd_kernel|rope_\w*kernel|swiglu_\w+|ce_\w+|lora_pack\w*kernel|fuse_pool_kernel|"
                  r"im2col\d_kernel|patchify\w*|cls_rows_kernel|embedding_kernel|adamw_kernel|sumsq_kernel|cast_kernel|dropout\w*)", n)
    return m.group(1) if m else ("torch:" + re.sub(r"[^A-Za-z_]", "", n)[:40] if "at::" in n else n[:40])

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * (float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)):]
gaps = collections.defaultdict(lambda: [0, 0])
end, prev, tot, busy = None, None, 0, 0
for r in rows:
    s, e, k = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])
    if end is not None and s > end:
        gaps[(prev, k)][0] += 1; gaps[(prev, k)][1] += s - end; tot += s - end
    busy += e - s
    if end is None or e > end:
        end, prev = e, k
span = end - int(rows[0]["Start_Timestamp"])
print(f"span {span/1e6:.2f} ms, kernels {busy/1e6:.2f} ms, idle {tot/1e6:.2f} ms ({100*tot/span:.1f} %)")
for (a, b), (n, t) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"{t/1e6:8.3f} ms  n={n:5d}  avg {t/n/1e3:7.1f} us   {a} -> {b}")
