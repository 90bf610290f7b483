#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the REAL reference (/root/reference).

Runs only in the build container (the reference does not exist on the GPU box and never travels).
One fresh process per fixture: SciPy's mvndst keeps a process-global, un-seedable random state
(SURVEY.md section 8c), so every fixture starts from the Fortran DATA seeds.

    python tests/golden/make_golden.py            # regenerate all fixtures (spawns subprocesses)
    python tests/golden/make_golden.py usps500    # one fixture, in this process

Shims: numexpr -> numpy eval; scipy.stats.mvn -> scipy.stats._mvn; the
learners run with parallelized=False (the serial path is the only reproducible one).
What is stored: inputs (feature rows, hyper-parameters, feedback) and the reference's outputs
(GP state, predictive mean/variance, per-step MI vectors, picks, mvndst call log).
"""
import os
import subprocess
import sys
import types

os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

FIXTURES = {
    # name: dataset, rows, length_scale, k per round, rounds, learner, learner kwargs
    "usps500": dict(data="usps", rows=500, ls=3.0, k=4, rounds=2, learner="ITAL", kw={}),
    "usps2007": dict(data="usps", rows=2007, ls=3.0, k=4, rounds=1, learner="ITAL", kw={}),
    "butterflies": dict(data="butterflies", rows=1000, ls=2.5, k=3, rounds=2, learner="ITAL", kw={}),
    "synth300": dict(data="synth", rows=300, d=16, ls=None, k=5, rounds=2, learner="ITAL", kw={}),
    "synth96_k6": dict(data="synth", rows=96, d=8, ls=None, k=6, rounds=1, learner="ITAL", kw={}),
    "synth200_noisy": dict(data="synth", rows=200, d=12, ls=None, k=3, rounds=2, learner="ITAL",
                           kw=dict(label_prob=0.8, mistake_prob=0.1)),
    "synth200_motivated": dict(data="synth", rows=200, d=12, ls=None, k=3, rounds=1, learner="ITAL",
                               kw=dict(label_prob=1.0, mistake_prob=0.15)),
    "synth200_optimistic": dict(data="synth", rows=200, d=12, ls=None, k=3, rounds=1, learner="ITAL",
                                kw=dict(label_estimation="optimistic")),
    "synth200_topcand": dict(data="synth", rows=200, d=12, ls=None, k=3, rounds=2, learner="ITAL",
                             kw=dict(top_candidates=40)),
    # float top_candidates = multiple of the number of labelled samples (reference ital.py:111-114; what the shipped
    # *-topscoring.conf use): 12, 50, 87 candidates in rounds 1 .. 3
    "synth200_topcand_float": dict(data="synth", rows=200, d=12, ls=None, k=3, rounds=3, learner="ITAL",
                                   kw=dict(top_candidates=12.5)),
    "iris_ce5": dict(data="iris", rows=120, ls=0.1, k=4, rounds=2, learner="ITAL",
                     kw=dict(change_estimation_subset=5)),
    "synth80_mcrel": dict(data="synth", rows=80, d=6, ls=None, k=5, rounds=1, learner="ITAL",
                          kw=dict(monte_carlo_num_rel=2)),
    "synth60_mcfb": dict(data="synth", rows=60, d=5, ls=None, k=3, rounds=2, learner="ITAL",
                         kw=dict(label_prob=0.7, mistake_prob=0.1, monte_carlo_num_fb=2)),
    "synth50_mcboth": dict(data="synth", rows=50, d=5, ls=None, k=4, rounds=1, learner="ITAL",
                           kw=dict(label_prob=1.0, mistake_prob=0.2, monte_carlo_num_rel=2, monte_carlo_num_fb=1)),
    "synth50_clip": dict(data="synth", rows=50, d=4, ls=None, k=3, rounds=2, learner="ITAL",
                         kw=dict(change_estimation_subset=4, clip_cov=0.35)),
    "usps500_mcmi": dict(data="usps", rows=500, ls=3.0, k=3, rounds=2, learner="MCMI_min",
                         kw=dict(subsample=150)),
    "synth300_mcmi": dict(data="synth", rows=300, d=16, ls=None, k=3, rounds=2, learner="MCMI_min", kw={}),
}


def install_shims():
    ne = types.ModuleType("numexpr")

    def evaluate(expr, local_dict=None, **_):
        env = {"exp": np.exp}
        env.update(local_dict or {})
        return eval(expr, {"__builtins__": {}}, env)

    ne.evaluate = evaluate
    sys.modules["numexpr"] = ne
    import scipy.stats
    import scipy.stats._mvn as _mvn
    sys.modules["scipy.stats.mvn"] = _mvn
    scipy.stats.mvn = _mvn
    for name in ("skimage", "skimage.io", "skimage.transform"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.path.insert(0, REF)
    return _mvn


def load_data(spec):
    if spec["data"] == "usps":
        X, y = [], []
        with open(os.path.join(REF, "data/usps_test.jf")) as f:
            f.readline()
            for line in f:
                s = line.strip()
                if s == "" or s == "-1":
                    break
                v = s.split()
                y.append(int(v[0]))
                X.append([float(x) for x in v[1:]])
        X = np.array(X)[: spec["rows"]]
        y = np.array(y)[: spec["rows"]]
        X = (X - X.min()) / (X.max() - X.min())  # datasets.py:110-112 min-max normalisation
        q = int(np.nonzero(y == 3)[0][0])
        rel = np.where(y == 3, 1.0, -1.0)
    elif spec["data"] == "butterflies":
        z = np.load(os.path.join(REF, "data/butterflies_pca50.npz"))
        X = np.asarray(z["X_train"], dtype=np.float64)[: spec["rows"]]
        y = np.asarray(z["y_train"])[: spec["rows"]]
        X = (X - X.min()) / (X.max() - X.min())
        cls = y[7]
        q = 7
        rel = np.where(y == cls, 1.0, -1.0)
    elif spec["data"] == "iris":
        from sklearn.datasets import load_iris
        from sklearn.model_selection import train_test_split
        d = load_iris()
        Xtr, _, ytr, _ = train_test_split(d.data, d.target, test_size=0.2, random_state=0)  # datasets.py:92
        X = (Xtr - Xtr.min()) / (Xtr.max() - Xtr.min())
        q = int(np.nonzero(ytr == 1)[0][0])
        rel = np.where(ytr == 1, 1.0, -1.0)
    else:
        rng = np.random.default_rng(1234 + spec["rows"] + spec["d"])
        X = rng.random((spec["rows"], spec["d"]))
        q = 5
        rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
        rel[q] = 1.0
    return np.ascontiguousarray(X, dtype=np.float64), q, rel


def run_fixture(name):
    spec = FIXTURES[name]
    _mvn = install_shims()
    X, q, rel = load_data(spec)
    ls = spec["ls"] if spec["ls"] is not None else float(np.sqrt(X.shape[1] / 12.0))

    # ---- instrumentation (monkey-patches, the reference code itself is untouched)
    calls = dict(n=[], err=[], val=[], inform=[], lower=[], infin=[], correl=[])
    orig = _mvn.mvndst

    def logged(lower, upper, infin, correl, maxpts=2000, abseps=1e-6, releps=1e-6):
        e, v, i = orig(lower, upper, infin, correl, maxpts=maxpts, abseps=abseps, releps=releps)
        n = len(lower)
        calls["n"].append(n)
        calls["err"].append(e)
        calls["val"].append(v)
        calls["inform"].append(i)
        if len(calls["lower"]) < 400 and n >= 2:
            calls["lower"].append(np.array(lower, dtype=np.float64))
            calls["infin"].append(np.array(infin, dtype=np.int32))
            calls["correl"].append(np.array(correl, dtype=np.float64))
        return e, v, i

    class _Proxy(types.ModuleType):
        pass

    proxy = _Proxy("mvnproxy")
    proxy.mvndst = logged
    import scipy.stats
    scipy.stats.mvn = proxy

    import ital.ital as ref_ital
    import ital.mcmi as ref_mcmi
    steps = []  # one dict per greedy step

    if spec["learner"] == "ITAL":
        cls = ref_ital.AppendedMutualInformation
        helper_mod = ref_ital
    else:
        cls = ref_mcmi.AppendedConditionalEntropy
        helper_mod = ref_mcmi
    orig_call, orig_append, orig_init = cls.__call__, cls.append, cls.__init__
    cur = dict(cand=[], val=[])

    def call(self, i):
        v = orig_call(self, i)
        cur["cand"].append(int(i))
        cur["val"].append(float(v))
        return v

    def append(self, i):
        steps.append(dict(cand=np.array(cur["cand"], dtype=np.int64), val=np.array(cur["val"], dtype=np.float64),
                          pick=int(i), ncalls_end=len(calls["n"])))
        cur["cand"], cur["val"] = [], []
        return orig_append(self, i)

    cls.__call__, cls.append = call, append

    Learner = getattr(helper_mod, spec["learner"])
    np.random.seed(0)
    learner = Learner(X, length_scale=ls, parallelized=False, **spec["kw"])
    out = dict(X=X, length_scale=ls, var=1.0, noise=1e-6, query=q, rel=rel, k=spec["k"], rounds=spec["rounds"])
    learner.update({q: 1})
    for r in range(spec["rounds"]):
        mean, var = learner.gp.predict_stored(cov_mode="diag")
        out[f"r{r}_ind"] = np.array(learner.gp.ind, dtype=np.int64)
        out[f"r{r}_y"] = np.array(learner.gp.y, dtype=np.float64)
        out[f"r{r}_rel_mean"] = np.array(learner.rel_mean, dtype=np.float64)
        out[f"r{r}_var"] = np.array(var, dtype=np.float64)
        out[f"r{r}_ncalls_begin"] = len(calls["n"])
        s0 = len(steps)
        ret = learner.fetch_unlabelled(spec["k"])
        out[f"r{r}_ret"] = np.array(ret, dtype=np.int64)
        if getattr(learner, "_ce_subset", None) is not None:
            out[f"r{r}_ce_subset"] = np.array(learner._ce_subset, dtype=np.int64)
        for t, st in enumerate(steps[s0:]):
            out[f"r{r}_s{t}_cand"] = st["cand"]
            out[f"r{r}_s{t}_mi"] = st["val"]
            out[f"r{r}_s{t}_pick"] = st["pick"]
            out[f"r{r}_s{t}_ncalls_end"] = st["ncalls_end"]
        learner.update({int(i): float(rel[i]) for i in ret})
    out["final_rel_mean"] = np.array(learner.rel_mean, dtype=np.float64)
    out["top_results_10"] = np.array(learner.top_results(10), dtype=np.int64)
    rng = np.random.default_rng(7)
    Xt = rng.random((16, X.shape[1]))
    pm, pv = learner.gp.predict(Xt, cov_mode="diag")
    out["predict_X"], out["predict_mean"], out["predict_var"] = Xt, pm, pv
    out["mvn_n"] = np.array(calls["n"], dtype=np.int16)
    out["mvn_val"] = np.array(calls["val"], dtype=np.float64)
    out["mvn_err"] = np.array(calls["err"], dtype=np.float64)
    out["mvn_inform"] = np.array(calls["inform"], dtype=np.int8)
    nlog = len(calls["lower"])
    out["mvnlog_count"] = nlog
    if nlog:
        nmax = max(len(a) for a in calls["lower"])
        L = np.zeros((nlog, nmax))
        I = np.zeros((nlog, nmax), dtype=np.int32)
        Cc = np.zeros((nlog, nmax * (nmax - 1) // 2))
        for j in range(nlog):
            L[j, : len(calls["lower"][j])] = calls["lower"][j]
            I[j, : len(calls["infin"][j])] = calls["infin"][j]
            Cc[j, : len(calls["correl"][j])] = calls["correl"][j]
        out["mvnlog_lower"], out["mvnlog_infin"], out["mvnlog_correl"] = L, I, Cc
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "ok: picks", [out[f"r{r}_ret"].tolist() for r in range(spec["rounds"])], "mvndst calls", len(calls["n"]))


def mvndst_stream_fixture():
    """A fresh-process sequence of mvndst calls (n = 1..12) pinning the oracle's MVNDST restatement,
    its random stream included."""
    _mvn = install_shims()
    rng = np.random.default_rng(2024)
    recs = []
    for it in range(240):
        n = [3, 4, 2, 5, 3, 6, 1, 4, 7, 8, 3, 9, 10, 11, 12, 5][it % 16]
        Z = rng.random((n, 6))
        D = ((Z[:, None] - Z[None]) ** 2).sum(-1)
        S = np.exp(-D / (2 * 0.8 ** 2)) + 0.02 * np.eye(n)
        s = np.sqrt(np.diag(S))
        Cn = S / np.outer(s, s)
        i_, j_ = np.tril_indices(n, -1)
        cor = Cn[i_, j_] if n > 1 else np.zeros(1)
        a = rng.normal(size=n) * 0.8
        inf = rng.integers(0, 2, size=n).astype(np.int32)
        e, v, i = _mvn.mvndst(a, a, inf, cor, maxpts=100 * n, abseps=1e-4, releps=1e-4)
        recs.append((n, a, inf, cor if n > 1 else np.zeros(0), e, v, i))
    nmax = 12
    out = dict(n=np.array([r[0] for r in recs], dtype=np.int32),
               lower=np.zeros((len(recs), nmax)), infin=np.zeros((len(recs), nmax), dtype=np.int32),
               correl=np.zeros((len(recs), nmax * (nmax - 1) // 2)),
               err=np.array([r[4] for r in recs]), val=np.array([r[5] for r in recs]),
               inform=np.array([r[6] for r in recs], dtype=np.int32))
    for j, r in enumerate(recs):
        out["lower"][j, : r[0]] = r[1]
        out["infin"][j, : r[0]] = r[2]
        out["correl"][j, : len(r[3])] = r[3]
    # known-answer tables for Phi (n=1) and the bivariate closed form (n=2)
    zs = np.concatenate([np.linspace(-39, 39, 157), rng.normal(size=100) * 2])
    out["phi_z"] = zs
    out["phi_val"] = np.array([_mvn.mvndst(np.array([z]), np.array([z]), np.array([0]), np.zeros(1))[1] for z in zs])
    bv = []
    for it in range(400):
        a = rng.normal(size=2) * 1.5
        r = float(np.tanh(rng.normal() * 1.5))
        inf = rng.integers(0, 2, size=2).astype(np.int32)
        e, v, i = _mvn.mvndst(a, a, inf, np.array([r]), maxpts=200, abseps=1e-4, releps=1e-4)
        bv.append((a[0], a[1], r, inf[0], inf[1], v))
    out["bvn"] = np.array(bv)
    np.savez_compressed(os.path.join(HERE, "mvndst_stream.npz"), **out)
    print("mvndst_stream ok")


def mvndst_stream_hi_fixture():
    """Fresh-process mvndst calls of dimension 13..20 (the Monte-Carlo pattern mode of ITAL reaches batch sizes above
    12): pins the Korobov generators and the stream consumption beyond the dimensions of mvndst_stream.npz.
    Includes singular cases (duplicated variables), which exercise COVSRT's zero-diagonal branch."""
    _mvn = install_shims()
    rng = np.random.default_rng(4711)
    recs = []
    for it in range(40):
        n = [13, 14, 16, 15, 17, 18, 19, 20][it % 8]
        Z = rng.random((n, 5))
        if it % 9 == 4:
            Z[3] = Z[1]            # two identical variables -> singular correlation matrix
        D = ((Z[:, None] - Z[None]) ** 2).sum(-1)
        S = np.exp(-D / (2 * 0.7 ** 2)) + (0.0 if it % 9 == 4 else 0.05) * np.eye(n)
        s = np.sqrt(np.diag(S))
        Cn = S / np.outer(s, s)
        i_, j_ = np.tril_indices(n, -1)
        cor = Cn[i_, j_]
        a = rng.normal(size=n) * 0.6
        if it % 9 == 4:
            a[3] = a[1]
        inf = rng.integers(0, 2, size=n).astype(np.int32)
        if it % 9 == 4:
            inf[3] = inf[1]
        e, v, i = _mvn.mvndst(a, a, inf, cor, maxpts=100 * n, abseps=1e-4, releps=1e-4)
        recs.append((n, a, inf, cor, e, v, i))
    nmax = 20
    out = dict(n=np.array([r[0] for r in recs], dtype=np.int32),
               lower=np.zeros((len(recs), nmax)), infin=np.zeros((len(recs), nmax), dtype=np.int32),
               correl=np.zeros((len(recs), nmax * (nmax - 1) // 2)),
               err=np.array([r[4] for r in recs]), val=np.array([r[5] for r in recs]),
               inform=np.array([r[6] for r in recs], dtype=np.int32))
    for j, r in enumerate(recs):
        out["lower"][j, : r[0]] = r[1]
        out["infin"][j, : r[0]] = r[2]
        out["correl"][j, : len(r[3])] = r[3]
    np.savez_compressed(os.path.join(HERE, "mvndst_stream_hi.npz"), **out)
    print("mvndst_stream_hi ok", out["val"][:6])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        if sys.argv[1] == "mvndst_stream":
            mvndst_stream_fixture()
        elif sys.argv[1] == "mvndst_stream_hi":
            mvndst_stream_hi_fixture()
        else:
            run_fixture(sys.argv[1])
    else:
        for name in ["mvndst_stream", "mvndst_stream_hi"] + list(FIXTURES):
            subprocess.check_call([sys.executable, os.path.abspath(__file__), name])
