#!/usr/bin/env python3
"""Gaps between the kernels of an HJB solve, from a rocprofv3 --kernel-trace CSV (argv[1] = *_kernel_trace.csv).
Prints, per kernel class of the attempt sequence: launches, mean duration, mean gap to the previous kernel's end."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
keys = ("k_stageV", "k_control", "k_out", "k_reduce_final", "k_halo_stage", "k_forces", "k_finish", "k_stage<", "k_scale", "k_density", "k_norm_scaled", "k_rhs", "k_fill", "k_lincomb")
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    name = r["Kernel_Name"]
    k = next((q for q in keys if q in name), None)
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if k:
        if "k_stageV" in name and ", 3, " in name:
            k = "k_stageV 3B" if "true" in name else "k_stageV 3A"
        dur[k].append(en - st)
        if prev_end is not None:
            gap[k].append(st - prev_end)
    prev_end = en
for k in dur:
    d, g = dur[k], gap[k]
    print("%-14s n=%5d  dur mean %8.2f us  median %8.2f | gap before: mean %7.2f us median %7.2f" % (
        k, len(d), sum(d) / len(d) / 1e3, sorted(d)[len(d) // 2] / 1e3, sum(g) / max(len(g), 1) / 1e3,
        sorted(g)[len(g) // 2] / 1e3 if g else 0))
