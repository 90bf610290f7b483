#!/usr/bin/env python3
"""Idle gaps between the kernels of one MCMI_min round (subsample 1000, k = 4) out of a rocprofv3 --kernel-trace CSV of
tools/mcmi_bench.py:   python tools/mcmi_gaps.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "cov_block_kernel" in r["Kernel_Name"]]
a, b = idx[3], idx[4]            # one round of the timed loop of the first (subsample 1000) configuration
prev = int(rows[a - 1]["End_Timestamp"])
gaps = busy = 0.0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%7.1f us idle  %8.1f us  %s" % ((s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    gaps += max(s - prev, 0) / 1e3
    busy += (e - s) / 1e3
    prev = max(prev, e)
print("round: busy %.1f us, idle %.1f us, %d kernels / copies" % (busy, gaps, b - a))
