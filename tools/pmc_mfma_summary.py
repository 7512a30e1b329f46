#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY pass: per kernel, MFMA-pipe busy cycles against (a) GRBM_GUI_ACTIVE/8 x 1024 SIMDs -- rocprof's MfmaUtil
definition -- and (b) wall time, plus the wave-cycle split.  Usage: pmc_mfma_summary.py <counter_collection.csv>"""
import collections, csv, re, sys
csv.field_size_limit(1 << 30)
PAT = re.compile(r"(gemm_bf16_wp_kernel<\w+>|gemm_bf16_h_kernel|gemm_bf16_w_kernel|gemm_bf16_kernel|gemm_bf16_l_kernel|gemm_skinny64_kernel|gemm_tn_mfma_kernel|gemm_smallm_kernel|"
                 r"attn_fwd_mfma<\d+, \d|attn_bwd_dkv_mfma|attn_bwd_dq_mfma)")
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen, dur, cnt = set(), collections.Counter(), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    m = PAT.search(r["Kernel_Name"])
    if not m:
        continue
    k = m.group(1)
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (k, r["Dispatch_Id"]) not in seen:
        seen.add((k, r["Dispatch_Id"]))
        cnt[k] += 1
        dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
print(f"{'kernel':24s} {'launches':>8s} {'avg us':>8s} {'MFMA busy % of GUI_ACTIVE':>26s} {'busy Gcycles/s/SIMD':>20s} {'GUI MHz':>8s} {'wait_any':>9s} {'wait_inst':>10s} {'active':>7s}")
for k in sorted(acc, key=lambda k: -dur[k]):
    c = acc[k]
    gui = c["GRBM_GUI_ACTIVE"] / 8
    wc = c["SQ_WAVE_CYCLES"] or 1
    print(f"{k:24s} {cnt[k]:8d} {1e6 * dur[k] / cnt[k]:8.1f} {100 * c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024):25.1f}% "
          f"{c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / dur[k] / 1e9:20.3f} {gui / dur[k] / 1e6:8.0f} {100 * c['SQ_WAIT_ANY'] / wc:8.1f}% "
          f"{100 * c['SQ_WAIT_INST_ANY'] / wc:9.1f}% {100 * c['SQ_ACTIVE_INST_ANY'] / wc:6.1f}%")
