"""Experiment harness for the device learners: counterpart of the reference's `utils.py` (config reader, learner
registry, NDCG / AUC), `datasets.py` (the loaders whose data ship with the reference or with scikit-learn) and
`run_experiment.py` (simulated relevance-feedback loop, AP / NDCG table), so that the reference's `configs/*.conf`
run unchanged against `ital_amd.ITAL` / `ital_amd.MCMI_min`:

    python -m ital_amd.harness configs/iris.conf [--rounds=3 --repetitions=2 ...]

SURVEY.md section 8(f) row f2.  Semantics kept: value casting of every config entry (`utils.py:144-166`), the
`import` directive (`:70-76`), `[EXPERIMENT]` overrides (`:79-80`), learner kwargs = `[METHOD_DEFAULTS]` + `[<method>]`
(`:110-119`), min-max scaling with the training split's extrema (`datasets.py:107-112`), class relevance +-1
(`:141-146`), `np.random.seed(0)` per class and the order in which numpy's global generator is consumed by the
query draw and the simulated user (`run_experiment.py:16-47, 138-142`), the result table (`:176-195`).
Learners other than ITAL / MCMI (the comparison baselines of the paper) are outside the MI355X hot path.
"""
import configparser
import math
import os
import sys
from collections import OrderedDict

import numpy as np


# ----------------------------------------------------------------------------------------------- configuration
class ConversionInterpolation(configparser.BasicInterpolation):
    """Every value is cast to int, float or bool when it parses as one (reference utils.py:144-166)."""

    def before_get(self, parser, section, option, value, defaults):
        val = super().before_get(parser, section, option, value, defaults)
        for cast in (int, float):
            try:
                return cast(val)
            except ValueError:
                pass
        low = val.lower()
        if low in ("yes", "on", "true"):
            return True
        if low in ("no", "off", "false"):
            return False
        return val


def read_config_file(config_file, section, overrides):
    """reference utils.py:43-83"""
    config = configparser.ConfigParser(interpolation=ConversionInterpolation())
    with open(config_file) as fh:
        config.read_file(fh)
    imports = config.get(section, "import", fallback=None)
    if imports:
        base = os.path.dirname(config_file) or "."
        files = [q if os.path.isabs(q) else os.path.join(base, q) for q in (w.strip() for w in imports.split())]
        config.read(files + [config_file])
    for key, value in overrides.items():
        config[section][key] = value
    return config


def _learners():
    from . import ITAL, MCMI_min
    from .baselines import LEARNERS as ranking
    table = {"ITAL": ITAL, "MCMI": MCMI_min}
    table.update(ranking)
    return table


BASELINES = ("SUD", "RBMAL", "TCAL", "USDM", "AdaptAL")


def make_learner(method, data, learner_config, **placement):
    table = _learners()
    if method not in table:
        if method in BASELINES:
            raise NotImplementedError("learner %r is one of the reference's comparison baselines that are not part of the "
                                      "MI355X path (ITAL, MCMI and the GP-sharing baselines random / topscoring / border / "
                                      "border_div / var / unc / entropy / EMOC are)" % method)
        raise KeyError("unknown learner %r" % method)
    return table[method](data, **learner_config, **placement)


def load_config(config_file, section="EXPERIMENT", overrides={}, **placement):
    """(config, dataset, learner) as reference utils.py:86-121; `placement` = device / rank / world / group."""
    config = read_config_file(config_file, section, overrides)
    name = config[section]["dataset"]
    dataset = load_dataset(name, **(config[name] if name in config else {}))
    method = config[section]["method"]
    learner_config = dict(config["METHOD_DEFAULTS"]) if "METHOD_DEFAULTS" in config else {}
    if method in config:
        learner_config.update(config[method])
    learner = make_learner(method, dataset.X_train_norm, learner_config, **placement)
    return config, dataset, learner


# ----------------------------------------------------------------------------------------------- datasets
class RetrievalDataset(object):
    """Train / test split, global min-max scaling, one-vs-rest relevance (reference datasets.py:40-146)."""

    queries = None

    def __init__(self, X, y, X_test=None, y_test=None, test_size=0.2):
        if X_test is None or y_test is None:
            from sklearn.model_selection import train_test_split
            self.X, self.y = np.array(X), np.array(y)
            (self.X_train, self.X_test, self.y_train, self.y_test, self.ind_train,
             self.ind_test) = train_test_split(self.X, self.y, np.arange(len(self.X)), test_size=test_size, random_state=0)
        else:
            self.X_train, self.y_train = np.array(X), np.array(y)
            self.X_test, self.y_test = np.array(X_test), np.array(y_test)
            self.X = np.concatenate([self.X_train, self.X_test])
            self.y = np.concatenate([self.y_train, self.y_test])
        self._preprocess()

    def _preprocess(self):
        self.X_max, self.X_min = self.X_train.max(), self.X_train.min()
        scale = self.X_max - self.X_min
        self.X_train_norm = (self.X_train - self.X_min) / scale
        self.X_test_norm = (self.X_test - self.X_min) / scale
        self.labels = np.unique(self.y)
        self.class_relevance = {lbl: (2 * (self.y_train == lbl) - 1, 2 * (self.y_test == lbl) - 1) for lbl in self.labels}


class IrisDataset(RetrievalDataset):
    def __init__(self, **kwargs):
        import sklearn.datasets
        X, y = sklearn.datasets.load_iris(return_X_y=True)
        super().__init__(X, y, **kwargs)


class StoredDataset(RetrievalDataset):
    """.npz with X_train, y_train, X_test, y_test (reference datasets.py:249-270)."""

    def __init__(self, data_file, **kwargs):
        z = np.load(data_file)
        super().__init__(z["X_train"], z["y_train"], z["X_test"], z["y_test"])


class USPSDataset(RetrievalDataset):
    """USPS in the `.jf` text format (reference datasets.py:347-389)."""

    def __init__(self, train_data_file, test_data_file, **kwargs):
        Xa, ya = self._read(train_data_file)
        Xb, yb = self._read(test_data_file)
        super().__init__(Xa, ya, Xb, yb)

    @staticmethod
    def _read(path):
        X, y = [], []
        with open(path) as fh:
            fh.readline()
            for line in fh:
                tok = line.split()
                if not tok or tok == ["-1"]:
                    break
                y.append(int(tok[0]))
                X.append([float(v) for v in tok[1:]])
        return np.array(X), np.array(y)


class WineDataset(RetrievalDataset):
    def __init__(self, data_file, **kwargs):
        raw = np.loadtxt(data_file, delimiter=",", dtype=float)
        super().__init__(raw[:, 1:], raw[:, 0].astype(int), **kwargs)


class LeafDataset(RetrievalDataset):
    def __init__(self, data_file, test_size=0.5, **kwargs):
        raw = np.loadtxt(data_file, delimiter=",", dtype=float)
        super().__init__(raw[:, 2:], raw[:, 0].astype(int), test_size=test_size, **kwargs)


DATASETS = {"Iris": IrisDataset, "Stored": StoredDataset, "USPS": USPSDataset, "Wine": WineDataset, "Leaf": LeafDataset}


def load_dataset(name, **kwargs):
    if name not in DATASETS:
        raise ValueError("Unknown dataset: {}".format(name))
    return DATASETS[name](**kwargs)


# ----------------------------------------------------------------------------------------------- metrics
def ndcg(y_true, y_score):
    """reference utils.py:174-200"""
    y_true = np.asarray(y_true)
    n_rel = int(np.sum(y_true > 0))
    rank, gain_sum, best = 0, 0.0, 0.0
    for j in np.argsort(y_score)[::-1]:
        if y_true[j] != 0:
            rank += 1
            g = 1.0 / math.log2(rank + 1)
            if y_true[j] > 0:
                gain_sum += g
            if rank <= n_rel:
                best += g
    return gain_sum / best


def area_under_curve(perf, normalized=True):
    """reference utils.py:203-226"""
    perf = np.asarray(perf)
    single = perf.ndim == 1
    if single:
        perf = perf[None, :]
    auc = (perf[:, 1:-1].sum(axis=-1) + (perf[:, 0] + perf[:, -1]) / 2) / perf.shape[1]
    return auc[0] if single else auc


def average_precision(y_true, y_score):
    from sklearn.metrics import average_precision_score
    return average_precision_score(y_true, y_score)


# ----------------------------------------------------------------------------------------------- experiment loop
def simulate_retrieval_feedback(labels, ret, label_prob=0.8, mistake_prob=0.05):
    """Simulated user (reference run_experiment.py:16-47); consumes numpy's global generator sample by sample."""
    fb = []
    for i in ret:
        if np.random.rand() >= label_prob:
            fb.append(0)
        elif np.random.rand() >= mistake_prob:
            fb.append(labels[i])
        elif labels[i] == 0:
            fb.append(np.random.choice([-1, 1]))
        else:
            fb.append(-1 if labels[i] > 0 else 1)
    return fb


def run_retrieval_experiment(config, dataset, learner, out=None, trace=None):
    """Active retrieval rounds for every class and query (reference run_experiment.py:79-195).  `trace`, if a list,
    receives (class, query, round, fetched batch, feedback)."""
    exp = "EXPERIMENT"
    out = sys.stdout if out is None else out
    classes = str(config.get(exp, "query_classes", fallback="")).split()
    if not classes:
        classes = list(dataset.class_relevance.keys())
    else:
        classes = [int(c) if c.lstrip("-").isdigit() else c for c in classes]
    n_neg = config.getint(exp, "initial_negatives", fallback=0)
    reps = config.getint(exp, "repetitions", fallback=10)
    n_init = config.getint(exp, "num_init", fallback=1)
    rounds = config.getint(exp, "rounds", fallback=10)
    batch = config.getint(exp, "batch_size")
    lp = config.getfloat(exp, "label_prob", fallback=1.0)
    mp = config.getfloat(exp, "mistake_prob", fallback=0.0)
    aps, ndcgs = OrderedDict(), OrderedDict()
    for lbl in classes:
        relevance, test_relevance = dataset.class_relevance[lbl]
        test_relevance = np.asarray(test_relevance)
        known = test_relevance != 0
        aps[lbl], ndcgs[lbl] = [], []
        np.random.seed(0)
        if dataset.queries is not None:
            queries = dataset.queries[lbl]
        else:
            queries = np.random.choice(np.nonzero(np.asarray(relevance) > 0)[0], (reps, n_init), replace=False)

        def evaluate():
            scores = np.asarray(learner.gp.predict(dataset.X_test_norm))
            return average_precision(test_relevance[known], scores[known]), ndcg(test_relevance, scores)

        for query in queries:
            learner.reset()
            learner.update({int(q): 1 for q in query})
            if n_neg > 0:
                neg = np.argpartition(learner.rel_mean, n_neg - 1)[:n_neg]
                learner.update({int(j): -1 for j in neg})
            it = [evaluate()]
            for r in range(rounds):
                ret = learner.fetch_unlabelled(batch)
                fb = simulate_retrieval_feedback(relevance, ret, label_prob=lp, mistake_prob=mp)
                learner.update(dict(zip(ret, fb)))
                if trace is not None:
                    trace.append((lbl, [int(q) for q in query], r, list(ret), [int(f) for f in fb]))
                it.append(evaluate())
            aps[lbl].append([a for a, _ in it])
            ndcgs[lbl].append([g for _, g in it])
    if config.get(exp, "avg_class_perf", fallback=True):
        tables = OrderedDict([("Overall Performance", (np.concatenate(list(aps.values())), np.concatenate(list(ndcgs.values()))))])
    else:
        tables = OrderedDict((lbl, (np.asarray(aps[lbl]), np.asarray(ndcgs[lbl]))) for lbl in aps)
    for title, (ap, nd) in tables.items():
        if len(tables) > 1:
            title = title if isinstance(title, str) else "Class {}".format(title)
            print("\n{}\n{:-<{}}\n".format(title, "", len(title)), file=out)
        print("Round;Median_AP;Mean_AP;AP_SD;Median_NDCG;Mean_NDCG;NDCG_SD", file=out)
        for i in range(ap.shape[1]):
            print("{};{:.4f};{:.4f};{:.4f};{:.4f};{:.4f};{:.4f}".format(
                i, np.median(ap[:, i]), np.mean(ap[:, i]), np.std(ap[:, i]), np.median(nd[:, i]), np.mean(nd[:, i]),
                np.std(nd[:, i])), file=out)
    return aps, ndcgs


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if not argv or argv[0].startswith("--"):
        raise SystemExit("usage: python -m ital_amd.harness <config file> [--key=value ...]")
    overrides = {}
    for arg in argv[1:]:
        if arg.startswith("--") and "=" in arg:
            key, value = arg[2:].split("=", 1)
            overrides[key] = value
    config, dataset, learner = load_config(argv[0], "EXPERIMENT", overrides)
    run_retrieval_experiment(config, dataset, learner)


if __name__ == "__main__":
    main()
