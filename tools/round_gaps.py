#!/usr/bin/env python3
"""Idle gaps between the kernels of one benchmark round, out of a rocprofv3 --kernel-trace CSV of bench.py.
    python tools/round_gaps.py gpurun_out/.../stats_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "qmc_main_kernel<4>" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
prev = int(rows[a]["End_Timestamp"])
gaps = busy = 0.0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%7.1f us idle  %8.1f us  %s" % ((s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    gaps += max(s - prev, 0) / 1e3
    busy += (e - s) / 1e3
    prev = e
print("round: busy %.1f us, idle %.1f us" % (busy, gaps))
