#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals (short names) restricted to the last N dispatches of
the library's kernels, plus GEMM launches grouped by grid size.  Usage: prof_summary.py <kernel_trace.csv> [skip_frac]"""
import csv, re, sys, collections

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.search(r"(attn_\w+|gemm_\w+|norm_fwd_kernel|rmsnorm_bwd_kernel|rope_kernel|swiglu_\w+|ce_\w+|lora_pack_kernel|fuse_pool_kernel|"
                  r"im2col\d_kernel|patchify\w*|cls_rows_kernel|embedding_kernel|adamw_kernel|sumsq_kernel|cast_kernel|kv_append_kernel|argmax_kernel)", n)
    if m:
        t = re.search(r"<([^>]{0,40})>", n)
        return m.group(1) + (f"<{t.group(1)}>" if t else "")
    return ("torch:" + re.sub(r"[^A-Za-z_:]", "", n)[:60]) if "at::" in n else n[:60]

rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * skip):]
tot = collections.defaultdict(lambda: [0, 0])
gemm = collections.defaultdict(lambda: [0, 0])
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    k = short(r["Kernel_Name"])
    tot[k][0] += 1; tot[k][1] += d
    if k.startswith("gemm_bf16"):
        g = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
        gemm[g][0] += 1; gemm[g][1] += d
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
allk = sum(v[1] for v in tot.values())
print(f"dispatches {len(rows)}  span {span/1e6:.2f} ms  kernel-sum {allk/1e6:.2f} ms")
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{d/1e6:10.3f} ms {100*d/allk:6.2f}%  n={c:6d} avg={d/c/1e3:9.1f} us  {k}")
print("-- gemm_bf16 by number of tiles (grid):")
for g, (c, d) in sorted(gemm.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{d/1e6:10.3f} ms  n={c:5d} avg={d/c/1e3:9.1f} us  tiles={g}")
