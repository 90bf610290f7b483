#!/usr/bin/env python3
"""Per dimension T of the general scorer's lattice sums (gen_main_kernel<T>) in BASELINE config 5's share
(tools/scale_probe.py 125000 512 16 1 under rocprofv3, tools/profile_r6.sh c5): launches, kernel time, algorithmic (Phi, Phi^-1)
pairs, fraction of the FP64 vector peak, vector instructions per pair, share of the vector issue slots, scalar per vector
instruction, scalar-unit activity, bytes fetched per launch.
    python tools/c5_fractions.py profiles/r6_c5_kernel_stats.csv profiles/r6_c5_pmc_summary.csv [rows=125000]"""
import csv
import re
import sys

PRIMES = (31, 47, 73, 113, 173, 263, 397, 593, 907, 1361)
FLOP_PER_PAIR = 53 + 0.85 * 48 + 0.15 * 126
PEAK = 78.6


def main():
    stats, pmc = sys.argv[1], sys.argv[2]
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 125000
    st = {}
    for r in csv.DictReader(open(stats)):
        m = re.search(r"gen_main_kernel<(\d+), false>", r["Name"])
        if m and int(m.group(1)) > 0:
            st[int(m.group(1))] = (int(r["Calls"]), float(r["TotalDurationNs"]) * 1e-9)
    pm = {}
    for r in csv.DictReader(open(pmc)):
        m = re.search(r"gen_main_kernel<(\d+), false>", r["kernel"])
        if m and int(m.group(1)) > 0:
            pm[int(m.group(1))] = r

    def f(row, key):
        try:
            return float(row[key])
        except (KeyError, ValueError, TypeError):
            return float("nan")
    print("BASELINE config 5's share of one of 8 ranks: tools/scale_probe.py %d 512 16 1 under rocprofv3 (tools/profile_r6.sh c5)" % n)
    print("%2s %9s %9s %11s %8s %12s %10s %16s %10s %14s %16s" % ("T", "launches", "total_ms", "pairs(1e9)", "TFLOP/s", "frac_of_78.6",
          "VALU/pair", "valu_issue_frac", "SALU/VALU", "sca_active/busy", "fetch_MB/launch"))
    tot_t = tot_p = 0.0
    for T in sorted(st):
        calls, sec = st[T]
        p = PRIMES[min(T - 1, 10) - 1]
        pairs = float(n) * T * 16 * p * (T - 1)
        tf = pairs * FLOP_PER_PAIR / sec / 1e12
        row = pm.get(T, {})
        valu, salu = f(row, "SQ_INSTS_VALU_avg"), f(row, "SQ_INSTS_SALU_avg")
        sca, busy = f(row, "SQ_ACTIVE_INST_SCA_avg"), f(row, "SQ_BUSY_CYCLES_avg")
        print("%2d %9d %9.1f %11.2f %8.1f %12.3f %10.1f %16.3f %10.2f %14.3f %16.1f" % (
            T, calls, sec * 1e3, pairs / 1e9, tf, tf / PEAK, valu * calls * 64 / pairs, valu * calls * 4 / (1024 * 2.4e9 * sec),
            salu / valu, sca / busy if busy == busy and busy else float("nan"), f(row, "fetch_bytes_corrected_avg") / 1e6))
        if T >= 7:
            tot_t += sec
            tot_p += pairs
    print("T >= 7 together: %.1f ms, %.3f of the FP64 vector peak (kernel time only)" % (tot_t * 1e3, tot_p * FLOP_PER_PAIR / tot_t / 1e12 / PEAK))


if __name__ == "__main__":
    main()
