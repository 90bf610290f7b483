#!/usr/bin/env python3
"""Calibration of bench.py's CPU baseline: the oracle (oracle/, kind "port") timed against the REAL
reference (/root/reference, importable in the build container only) on identical inputs, serial mode, one process each.

    python tools/calibrate_oracle.py > profiles/r2_oracle_calibration.json

The ratio oracle/reference is what bench.py quotes next to `cpu_baseline` (the reference cannot travel to the GPU box)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, D, K = 400, 64, 4


def run(which):
    os.environ.update(OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    import numpy as np
    sys.path.insert(0, ROOT)
    X = np.random.default_rng(0).random((N, D))
    ls = float(np.sqrt(D / 12.0))
    if which == "reference":
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_golden
        make_golden.install_shims()
        from ital.ital import ITAL
        L = ITAL(X, length_scale=ls, parallelized=False)
    else:
        from oracle.ital import OracleITAL
        L = OracleITAL(X, length_scale=ls)
    L.update({0: 1})
    t0 = time.perf_counter()
    ret = L.fetch_unlabelled(K)
    dt = time.perf_counter() - t0
    print(json.dumps({"which": which, "seconds": dt, "picks": [int(i) for i in ret]}))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        res = {}
        for which in ("reference", "oracle"):
            best = None
            for _ in range(2):
                out = subprocess.check_output([sys.executable, os.path.abspath(__file__), which], text=True)
                j = json.loads(out.strip().splitlines()[-1])
                if best is None or j["seconds"] < best["seconds"]:
                    best = j
            res[which] = best
        assert res["reference"]["picks"] == res["oracle"]["picks"], res
        print(json.dumps({"workload": "fetch_unlabelled(%d) on %d x %d synthetic, serial, one core" % (K, N, D),
                          "reference_s": res["reference"]["seconds"], "oracle_s": res["oracle"]["seconds"],
                          "oracle_over_reference": res["oracle"]["seconds"] / res["reference"]["seconds"],
                          "picks": res["oracle"]["picks"], "host": "build container (8 cores), best of 2"}, indent=1))
