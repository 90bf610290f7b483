#!/usr/bin/env python3
"""Kernel-stats summary (CSV, same columns as `rocprofv3 --stats` kernel_stats.csv) out of a rocprofv3 rocpd database.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.csv
"""
import csv
import sqlite3
import sys

import numpy as np


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, end - start from kernels").fetchall()
    by = {}
    for name, dur in rows:
        by.setdefault(name, []).append(dur)
    total = float(sum(sum(v) for v in by.values()))
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        a = np.asarray(v, dtype=np.float64)
        w.writerow([name, len(v), int(a.sum()), round(float(a.mean()), 3), round(100 * a.sum() / total, 4), int(a.min()),
                    int(a.max()), round(float(a.std(ddof=1)) if len(v) > 1 else 0.0, 3)])


if __name__ == "__main__":
    main(sys.argv[1])
