#!/usr/bin/env python3
"""End-to-end golden tables: runs the REAL reference harness (utils.load_config + run_experiment.run_retrieval_experiment,
learner in serial mode) on the small configs under tests/golden/conf/ and stores the printed AP / NDCG table plus every
fetched batch and simulated feedback.  This container only (imports /root/reference through the shims of
make_golden.py); the product never imports it.

    python tests/golden/make_golden_harness.py harness_iris      (one fresh process per fixture)
"""
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402

NAMES = ["harness_iris", "harness_noisy", "harness_mcmi", "harness_topscoring", "harness_border", "harness_unc", "harness_random",
         "harness_var", "harness_emoc", "harness_entropy", "harness_border_div"]


def run(name):
    make_golden.install_shims()
    import matplotlib
    matplotlib.use("Agg")
    os.chdir(make_golden.REF)
    import utils as ref_utils
    import run_experiment as ref_run
    conf = os.path.join(HERE, "conf", name + ".conf")
    config, dataset, learner = ref_utils.load_config(conf, "EXPERIMENT", {})
    learner.parallelized = False
    trace = []
    fetch, update = learner.fetch_unlabelled, learner.update

    def logged_fetch(k, *a, **kw):
        ret = fetch(k, *a, **kw)
        trace.append(dict(ret=[int(i) for i in ret]))
        return ret

    def logged_update(fb):
        if trace and "fb" not in trace[-1] and len(fb) == len(trace[-1]["ret"]) and list(fb.keys()) == trace[-1]["ret"]:
            trace[-1]["fb"] = [int(v) for v in fb.values()]
        return update(fb)

    learner.fetch_unlabelled, learner.update = logged_fetch, logged_update
    buf = io.StringIO()
    stdout = sys.stdout
    sys.stdout = buf
    try:
        ref_run.run_retrieval_experiment(config, dataset, learner)
    finally:
        sys.stdout = stdout
    out = dict(table=buf.getvalue(), trace=trace, n_train=int(len(dataset.X_train_norm)))
    with open(os.path.join(HERE, name + ".json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(name, "ok:", len(trace), "fetches")
    print(buf.getvalue())


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        import subprocess
        for n in NAMES:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), n])
