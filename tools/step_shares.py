#!/usr/bin/env python3
"""Where the time of the general scorer's greedy steps goes, out of a rocprofv3 --kernel-trace CSV of a fetch_unlabelled(k)
run (tools/scale_probe.py n d k mc).

A greedy step = the gen_* kernels between two kernels of the library that are not the general scorer's (selection, the new
member's covariance column); memory fills and copies (hipMemsetAsync of the pipeline's counters, torch's uploads) do NOT end
a step.  Per step:
    T         largest gen_main_kernel<T> of the step (variables of its lattice sums)
    gap_ms    from the end of the previous step's last kernel (its selection / covariance column) to this step's first
              kernel: the host's share between two steps (download of the pick, the first range's pattern sampling)
    span_ms   first kernel start .. last kernel end of the step
    busy_ms   time at least one kernel (of ANY name, both streams) was running inside the span -- union of the intervals
    idle_ms   span - busy: the GPU had nothing resident (host-side pattern sampling of a later range, launch gaps)
    main_ms   sum of gen_main_kernel<T > 0> (the lattice sums)
Totals at the end: idle inside the steps + the gaps between them = GPU idle of the round's scoring.

(Round 5's version cut the trace at every fill kernel and summed per-run spans: kernels that had started in an earlier run
were not counted as busy, so a run that only launched the next slab's preparation under a running lattice sum showed up
as ~140 ms of "idle" -- the 1.9 s the round-5 verdict read out of it was mostly that artefact.)

    python tools/step_shares.py gpurun_out/.../stats_kernel_trace.csv
"""
import csv
import re
import sys


def union_ms(iv):
    iv = sorted(iv)
    busy, hi = 0.0, None
    for s, e in iv:
        if hi is None or s > hi:
            busy += (e - s) / 1e6
            hi = e
        elif e > hi:
            busy += (e - hi) / 1e6
            hi = e
    return busy


def main(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], None
    other_end = None           # end of the last non-scorer library kernel seen (selection / covariance column)
    for r in rows:
        name = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "ital::gen_" in name:
            if cur is None:
                cur = {"rows": [], "all": [], "prev_end": other_end}
            cur["rows"].append(r)
            cur["all"].append((s, e))
        elif "ital::" in name:
            if cur is not None:
                steps.append(cur)
                cur = None
            other_end = e if other_end is None else max(other_end, e)
        elif cur is not None:
            cur["all"].append((s, e))        # fills / copies inside a step count as busy time, and do not end it
    if cur is not None:
        steps.append(cur)
    print("%3s %8s %9s %9s %8s %9s %5s | %s" % ("T", "gap_ms", "span_ms", "busy_ms", "idle_ms", "main_ms", "kern", "ms per kernel (launches)"))
    tot = dict(gap=0.0, span=0.0, busy=0.0, main=0.0)
    for st in steps:
        T, by = 0, {}
        for r in st["rows"]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            short = re.sub(r"^void ital::", "", r["Kernel_Name"].split("(")[0])
            m = re.match(r"gen_main_kernel<(\d+)", short)
            if m and int(m.group(1)) > 0:
                T = max(T, int(m.group(1)))
            d = by.setdefault(short, [0.0, 0])
            d[0] += (e - s) / 1e6
            d[1] += 1
        first = min(s for s, _ in st["all"])
        last = max(e for _, e in st["all"])
        span = (last - first) / 1e6
        busy = union_ms(st["all"])
        gap = (first - st["prev_end"]) / 1e6 if st["prev_end"] is not None else 0.0
        main_ms = sum(v[0] for k, v in by.items() if k.startswith("gen_main_kernel") and not k.startswith("gen_main_kernel<0"))
        tot["gap"] += max(gap, 0.0)
        tot["span"] += span
        tot["busy"] += busy
        tot["main"] += main_ms
        parts = ", ".join("%s %.2f (%d)" % (k, v[0], v[1]) for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:4])
        print("%3d %8.2f %9.2f %9.2f %8.2f %9.2f %5d | %s" % (T, gap, span, busy, span - busy, main_ms, len(st["rows"]), parts))
    idle = tot["span"] - tot["busy"]
    whole = tot["span"] + tot["gap"]
    print("all steps: span %.1f ms + gaps between steps %.1f ms = %.1f ms; busy %.1f ms; idle inside steps %.1f ms (%.1f %% of the "
          "spans), idle incl. gaps %.1f ms = %.1f %% of the scoring time; lattice sums %.1f ms = %.3f of it"
          % (tot["span"], tot["gap"], whole, tot["busy"], idle, 100 * idle / max(tot["span"], 1e-9), idle + tot["gap"],
             100 * (idle + tot["gap"]) / max(whole, 1e-9), tot["main"], tot["main"] / max(whole, 1e-9)))


if __name__ == "__main__":
    main(sys.argv[1])
