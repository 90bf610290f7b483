#!/usr/bin/env python3
"""Where the time of the general scorer's steps goes, out of a rocprofv3 --kernel-trace CSV of a fetch_unlabelled(k) run
(tools/scale_probe.py n d k mc): the trace is cut into runs of consecutive gen_* kernels (one ital_score_generic step
each; any other kernel -- cross-covariance column, selection -- ends a run); per step: the wall span from its first kernel's
start to its last kernel's end, the time some kernel of it was running (union over both streams), the sums per kernel and
the idle time inside the span (host pattern uploads, launch gaps).

    python tools/step_shares.py gpurun_out/.../stats_kernel_trace.csv
"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
steps, cur = [], []
for r in rows:
    name = r["Kernel_Name"]
    if "ital::gen_" in name:
        cur.append(r)
    elif cur:
        steps.append(cur)
        cur = []
if cur:
    steps.append(cur)
print("%3s %9s %9s %9s %7s | %s" % ("T", "span_ms", "busy_ms", "idle_ms", "launches", "ms per kernel (launches)"))
tot_span = tot_main = 0.0
for st in steps:
    T = 0
    by = {}
    iv = []
    for r in st:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        iv.append((s, e))
        short = re.sub(r"^void ital::", "", r["Kernel_Name"].split("(")[0])
        m = re.match(r"gen_main_kernel<(\d+)", short)
        if m and int(m.group(1)) > 0:
            T = max(T, int(m.group(1)))
        d = by.setdefault(short, [0.0, 0])
        d[0] += (e - s) / 1e6
        d[1] += 1
    iv.sort()
    busy, hi = 0.0, iv[0][0]
    for s, e in iv:
        if e > hi:
            busy += (e - max(s, hi)) / 1e6
            hi = e
    span = (max(e for _, e in iv) - iv[0][0]) / 1e6
    tot_span += span
    tot_main += sum(v[0] for k, v in by.items() if k.startswith("gen_main_kernel") and not k.endswith("<0>"))
    parts = ", ".join("%s %.2f (%d)" % (k, v[0], v[1]) for k, v in sorted(by.items(), key=lambda kv: -kv[1][0]))
    print("%3d %9.2f %9.2f %9.2f %7d | %s" % (T, span, busy, span - busy, len(st), parts))
print("all steps: span %.1f ms, lattice sums (gen_main_kernel<T>, T > 0) %.1f ms = %.3f of it" % (tot_span, tot_main, tot_main / max(tot_span, 1e-9)))
