import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import torch
from ital_amd import ITAL, mvn_stream
import make_golden
z = np.load("tests/golden/iris_ce5.npz")
spec = make_golden.FIXTURES["iris_ce5"]
res = {}
for pipe in (True, False):
    mvn_stream.GLOBAL.reset()
    np.random.seed(0)
    L = ITAL(z["X"], length_scale=float(z["length_scale"]), device="cuda:0", **spec["kw"])
    L.keep_scores = True
    L.generic_pipeline = pipe
    L.update({int(z["query"]): 1})
    out = []
    for r in range(int(z["rounds"])):
        ret = L.fetch_unlabelled(int(z["k"]))
        cand0 = z[f"r{r}_s0_cand"].tolist()
        pos = {c: i for i, c in enumerate(cand0)}
        for t in range(len(ret)):
            cand = z[f"r{r}_s{t}_cand"].tolist()
            mine = L.last_scores[t].cpu().numpy()[[pos[c] for c in cand]]
            want = z[f"r{r}_s{t}_mi"]
            bad = np.flatnonzero(np.abs(mine - want) > 1e-8 * np.abs(want) + 1e-10)
            out.append((r, t, mine))
            for b in bad:
                print("pipeline" if pipe else "single  ", "round", r, "step", t, "cand", cand[b], "in subset", cand[b] in L._ce_subset,
                      "picks so far", ret[:t], "device %.12e golden %.12e rel %.2e" % (mine[b], want[b], abs(mine[b] - want[b]) / abs(want[b])), "subset", L._ce_subset)
        L.update({int(i): float(z["rel"][int(i)]) for i in ret})
    res[pipe] = out
for (r, t, a), (_, _, b) in zip(res[True], res[False]):
    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-12)
    print("round", r, "step", t, "pipeline vs single kernel: max rel", float(np.nanmax(d)), "at", int(np.nanargmax(d)))
