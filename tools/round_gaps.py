#!/usr/bin/env python3
"""Idle gaps between the kernels of one benchmark round, out of a rocprofv3 --kernel-trace CSV of bench.py.
    python tools/round_gaps.py gpurun_out/.../stats_kernel_trace.csv [out.json]
The round is one of the last of the timed loop of bench.py (the last three rounds of a run are its untimed
sampling of the short kernels).  With a second argument the totals are also written as JSON (bench.py quotes them)."""
import csv
import json
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "qmc_main_kernel<4>" in r["Kernel_Name"]]
mid = max(len(idx) - 5, 1)      # bench.py ends with 3 untimed sampling rounds: this one lies in the timed loop
a, b = idx[mid - 1], idx[mid]
prev = int(rows[a]["End_Timestamp"])
gaps = busy = 0.0
launches = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%7.1f us idle  %8.1f us  %s" % ((s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
    gaps += max(s - prev, 0) / 1e3
    busy += (e - s) / 1e3
    prev = max(prev, e)
    launches += 1
print("round: busy %.1f us, idle %.1f us, %d kernels / copies" % (busy, gaps, launches))
if len(sys.argv) > 2:
    json.dump({"busy_us": round(busy, 1), "idle_us": round(gaps, 1), "gpu_busy_frac": round(busy / (busy + gaps), 4),
               "kernels_and_copies_per_round": launches,
               "note": "one fetch + update round of the headline workload under rocprofv3 --kernel-trace (tools/round_gaps.py); "
                       "kernel times under the profiler are a few percent above the un-profiled ones"},
              open(sys.argv[2], "w"))
