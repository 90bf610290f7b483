#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters (counter_collection.csv files, one pass each), one row per kernel.

    python tools/pmc_summary.py gpurun_out/prof_x/*/*_counter_collection.csv > profiles/x_pmc.csv

FETCH_SIZE / WRITE_SIZE come in units of 1024 bytes.  Per /opt/skills/guides/MI355X_MICROARCH.md (HBM section) gfx950's
FETCH_SIZE tallies the 128-B requests of wide coalesced reads at 64 B, so it is doubled here
(fetch_bytes_corrected_avg); WRITE_SIZE is taken as is.  SQ_* counters are quad-cycle counts summed over all SIMDs.
"""
import csv
import sys
from collections import defaultdict


def main(paths):
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for path in paths:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"]
                if "ital::" not in name:
                    continue
                acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur[name].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    counters = sorted({c for k in acc.values() for c in k})
    w = csv.writer(sys.stdout, quoting=csv.QUOTE_NONNUMERIC)
    head = ["kernel", "launches_per_pass"] + [c + "_avg" for c in counters]
    if "FETCH_SIZE" in counters:
        head.append("fetch_bytes_corrected_avg")
    if "WRITE_SIZE" in counters:
        head.append("write_bytes_avg")
    if "SQ_ACTIVE_INST_VALU" in counters and "SQ_WAVE_CYCLES" in counters:
        head += ["valu_active_per_wave_cycle", "valu_active_per_busy_cycle_per_simd"]
    w.writerow(head)
    for name in sorted(acc, key=lambda k: -sum(dur[k])):
        cs = acc[name]
        n = max(len(v) for v in cs.values())
        avg = {c: (sum(cs[c]) / len(cs[c]) if cs.get(c) else float("nan")) for c in counters}
        row = [name.split("(")[0], n] + [round(avg[c], 3) for c in counters]
        if "FETCH_SIZE" in counters:
            row.append(round(avg["FETCH_SIZE"] * 1024 * 2, 1))
        if "WRITE_SIZE" in counters:
            row.append(round(avg["WRITE_SIZE"] * 1024, 1))
        if "SQ_ACTIVE_INST_VALU" in counters and "SQ_WAVE_CYCLES" in counters:
            row.append(round(avg["SQ_ACTIVE_INST_VALU"] / avg["SQ_WAVE_CYCLES"], 4))
            busy = avg.get("SQ_BUSY_CYCLES", float("nan"))
            row.append(round(avg["SQ_ACTIVE_INST_VALU"] / busy, 4) if busy else float("nan"))
        w.writerow(row)


if __name__ == "__main__":
    main(sys.argv[1:])
