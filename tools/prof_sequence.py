#!/usr/bin/env python3
"""Kernel sequence of the LAST train step in a rocprofv3 kernel trace (CSV), runs of one kernel collapsed: what is launched besides our own kernels?
Usage: prof_sequence.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# a step starts with the log-mel / conv front end: find the last launch of the first kernel name that appears exactly once per step
first = next((i for i in range(len(names) - 1, -1, -1) if "patchify" in names[i]), 0)
start = max((i for i in range(first, -1, -1) if "logmel" in names[i] or "im2col" in names[i] or i == first), default=first)
seq, tot = [], {}
for r in rows[min(start, first):]:
    n = r["Kernel_Name"][:90]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0
    if seq and seq[-1][0] == n:
        seq[-1][1] += 1; seq[-1][2] += d
    else:
        seq.append([n, 1, d])
    tot[n] = tot.get(n, 0.0) + d
for n, c, d in seq:
    print(f"{c:4d} x {d / c:9.1f} us  {n}")
print("---- totals from that point")
for n, d in sorted(tot.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{d / 1000:9.3f} ms  {n}")
