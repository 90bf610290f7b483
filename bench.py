#!/usr/bin/env python3
"""Benchmark of the ITAL hot path on MI355X: MI-scored candidates/sec per fetch_unlabelled(k) round.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one round of the reference's retrieval loop (run_experiment.py:160-164): fetch_unlabelled(k) on the
current relevance model followed by update() with the simulated feedback for the fetched batch (the update is
inside the timed region: nothing is skipped).  Workload (BASELINE.json configs[1], SURVEY.md section 8d C2'):
synthetic USPS-shaped features, 9298 x 256 fp64 in [0,1], length_scale 3.0, k = 4, perfect user; with N > 1 every
rank holds 9298 rows (weak scaling) and the greedy steps exchange one record per rank over RCCL.
A scored candidate = one (candidate, greedy step) MI evaluation with full 2^t sign-pattern enumeration.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS_PER_GPU = 9298
DIM = 256
BATCH = 4
LENGTH_SCALE = 3.0
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (= half the 157.3 TF FP32 vector rate of MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0
PRIMES = (31, 47, 73, 113, 173, 263, 397, 593, 907, 1361)
# FP64 work of one (Phi, Phi^-1) pair of the lattice integrand in flops (FMA = 2, multiply / add = 1), counted in the ISA
# of the isolated chain (tools/ubench/phi_ubench.hip): Hart's Phi 53 flops (22 FMA + 9 mul/add among 47 vector
# instructions), AS241's central Phi^-1 48 (21 + 6 among 45), its log/sqrt tail branch 126 (56 + 14 among 121) for the
# 15 % of the arguments that need it.  The chain, not the kernel: bookkeeping the kernel adds is not achieved work.
FLOP_PER_PAIR = 53 + 0.85 * 48 + 0.15 * 126
VALU_PER_PAIR_CHAIN = 47 + 0.85 * 45 + 0.15 * 121      # vector instructions of the isolated chain per pair


def make_data(n, d, seed):
    rng = np.random.default_rng(seed)
    return rng.random((n, d))


def qmc_pairs(t, n_cand):
    """(Phi, Phi^-1) pairs the scorer of greedy step t must evaluate: 2^t prior orthant probabilities per candidate,
    16*P lattice evaluations each, t - 1 pairs per evaluation (the first variable's Phi is the same for every point
    and the last variable needs no Phi^-1; the 2^t post-update probabilities are provably 1 and cost none)."""
    p = PRIMES[min(t - 1, 10) - 1]
    return n_cand * (2 ** t) * 16 * p * (t - 1)


def hbm_stream_probe(device, rows=1_000_000, d=DIM, m=21, reps=10):
    """The HBM-bound streaming kernel of the path (one cross-covariance column: reads X and V once, writes one column)
    at a size that leaves the caches: algorithmic bytes 8*(d + m + 1 + 1) per row, timed with HIP events
    on the launch stream.  The benchmark workload itself (9298 rows, 19 MB per launch) is launch-latency bound."""
    import torch
    from ital_amd import _lib
    from ital_amd.gp import _ptr, _stream
    lib = _lib.lib()
    ldv = (rows + 15) // 16 * 16
    cap = 32
    with torch.cuda.device(device):
        X = torch.rand((rows, d), dtype=torch.float64, device=device)
        xn = torch.empty(rows, dtype=torch.float64, device=device)
        V = torch.rand((cap, ldv), dtype=torch.float64, device=device) * 0.01
        out = torch.empty(ldv, dtype=torch.float64, device=device)
        W = torch.rand(cap, dtype=torch.float64, device=device) * 0.01
        sn = torch.empty(1, dtype=torch.float64, device=device)
        st = _stream()
        lib.ital_row_norms(_ptr(X), rows, d, _ptr(xn), st)
        lib.ital_row_norms(_ptr(X), 1, d, _ptr(sn), st)

        def launch():
            _lib.check(lib.ital_cross_cov_cols(_ptr(X), _ptr(xn), rows, d, _ptr(X), _ptr(sn), 1, _ptr(W), cap, _ptr(V), ldv, m,
                                               1.0, LENGTH_SCALE, _ptr(out), ldv, st))
        launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            launch()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
    nbytes = 8.0 * rows * (d + m + 2)
    ach = nbytes / sec / 1e9
    return {"bound": "hbm", "kernel": "kcols_kernel (ital_cross_cov_cols, c=1)", "achieved": ach, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": sec * 1e3,
            "rows": rows, "d": d, "m": m, "algorithmic_bytes_per_launch": nbytes,
            "note": "same kernel as the greedy steps' cross-covariance column, at 1M rows (2.2 GB per launch)"}


def other_workloads(X, rel, device):
    """Secondary timings on the same synthetic data (rank 0, N = 1 only; not part of `value`): the general scorer with a
    noisy user (reference configs usps-mistakes / mirflickr-mistakes style) and MCMI_min with the reference's subsample."""
    import torch
    from ital_amd import ITAL, MCMI_min, mvn_stream
    out = {}

    def timed(learner, rounds, k):
        learner.update({0: 1})
        ret = learner.fetch_unlabelled(k)                 # warm-up round
        learner.update({int(i): float(rel[i]) for i in ret})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scored = 0
        for _ in range(rounds):
            n_c = len(learner.get_unseen())
            ret = learner.fetch_unlabelled(k)
            learner.update({int(i): float(rel[i]) for i in ret})
            scored += sum(n_c - t for t in range(k))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return {"ms_per_round": dt / rounds * 1e3, "candidates_per_s": scored / dt}

    mvn_stream.GLOBAL.reset()
    out["ital_general_user_k4"] = dict(timed(ITAL(X, length_scale=LENGTH_SCALE, label_prob=0.5, mistake_prob=0.25,
                                                  device=device), 2, BATCH),
                                       config="label_prob 0.5, mistake_prob 0.25: 3^t - 1 feedback configurations per pattern")
    np.random.seed(0)
    m = MCMI_min(X, length_scale=LENGTH_SCALE, subsample=1000, device=device)
    r = timed(m, 5, BATCH)
    r["candidates_per_s"] = BATCH * 1000 / (r["ms_per_round"] * 1e-3)
    out["mcmi_min_subsample1000_k4"] = dict(r, config="MCMI_min, subsample 1000 (reference configs/usps.conf)")
    return out


def pmc_row(kernel_prefix):
    """Row of a kernel in the committed PMC summary (profiles/, collected with tools/profile_gpu.sh in separate passes and
    corrected as MI355X_MICROARCH.md prescribes), or None."""
    import csv
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*pmc_summary*.csv")) if "mcmi" not in f)
    if not files:
        return None
    with open(files[-1], newline="") as f:
        for row in csv.DictReader(f):
            if row["kernel"].startswith(kernel_prefix):
                return row
    return None


def pmc_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel (fetch + write), or None."""
    row = pmc_row(kernel_prefix)
    try:
        return float(row["fetch_bytes_corrected_avg"]) + float(row["write_bytes_avg"])
    except (TypeError, KeyError, ValueError):
        return None


def pmc_valu_issue_frac(kernel_prefix, launch_s):
    """Share of the chip's vector-issue cycles the kernel's VALU instructions fill: a wave64 instruction holds its SIMD's
    16 lanes for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz.  Instruction count from the committed PMC pass, launch time live."""
    row = pmc_row(kernel_prefix)
    try:
        return float(row["SQ_INSTS_VALU_avg"]) * 4.0 / (1024 * 2.4e9 * launch_s)
    except (TypeError, KeyError, ValueError):
        return None


def cpu_baseline(X, cores):
    """The oracle (CPU restatement of the reference, oracle/) in the reference's parallel mode on a bounded sample."""
    from oracle.ital import OracleITAL
    from oracle.parallel import fetch_unlabelled_parallel
    n = int(min(len(X), max(512, 600 * cores)))
    learner = OracleITAL(X[:n], length_scale=LENGTH_SCALE)
    learner.update({0: 1})
    t0 = time.time()
    _, scored = fetch_unlabelled_parallel(learner, BATCH, processes=cores)
    dt = time.time() - t0
    return {"value": scored / dt, "unit": "candidates/s", "cores": cores, "kind": "port",
            "sample": "one fetch_unlabelled(%d) round on the first %d rows of the workload (%d scored candidates, "
                      "%.1f s), fork pool per greedy step as reference ital/ital.py:124-126" % (BATCH, n, scored, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rows", type=int, default=9298, help="rows per GPU (experiments; the default is the metric's workload)")
    ap.add_argument("--batch", type=int, default=4, help="batch size k (experiments)")
    ap.add_argument("--label-prob", type=float, default=1.0, help="user model (experiments; != 1 selects the general scorer)")
    ap.add_argument("--mistake-prob", type=float, default=0.0)
    ap.add_argument("--force-collectives", action="store_true",
                    help="one rank, but through the exchange path of N > 1 (1-rank RCCL group): prices the per-step collective")
    args = ap.parse_args()
    globals().update(ROWS_PER_GPU=args.rows, BATCH=args.batch)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_base = None
    if world == 1 and not args.no_cpu_baseline:
        # before anything touches the GPU: the baseline forks worker pools
        cpu_base = cpu_baseline(make_data(ROWS_PER_GPU, DIM, seed=0), os.cpu_count() or 1)
    import torch
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1 or args.force_collectives:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.force_collectives:
            os.environ["ITAL_FORCE_COLLECTIVES"] = "1"
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        group = dist.group.WORLD

    from ital_amd import ITAL, mvn_stream

    n_total = ROWS_PER_GPU * world
    X = make_data(n_total, DIM, seed=0)
    rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
    learner = ITAL(X, length_scale=LENGTH_SCALE, label_prob=args.label_prob, mistake_prob=args.mistake_prob, device=device,
                   rank=rank, world=world, group=group)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def one_round():
        ret = learner.fetch_unlabelled(BATCH)
        learner.update({int(i): float(rel[i]) for i in ret})
        return ret

    def restart():
        learner.reset()
        mvn_stream.GLOBAL.reset()
        learner.update({0: 1})

    restart()
    for _ in range(args.warmup):
        one_round()
    restart()
    learner.profile = None if os.environ.get("ITAL_BENCH_NO_EVENTS") else []
    # timing events are created before the timed region (only recorded inside it)
    learner.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(4 * (2 * BATCH) * args.steps)]
    scored = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_cand = n_total - len(learner.relevant_ids) - len(learner.irrelevant_ids)
        one_round()
        scored += sum(n_cand - t for t in range(BATCH))
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # per-kernel durations from the HIP events recorded on the launch stream during the timed region
    prof = {}
    for name, t, n_c, e0, e1 in (learner.profile or []):
        prof.setdefault((name, t), []).append((e0.elapsed_time(e1) * 1e-3, n_c))
    out = None
    if rank == 0:
        qm = prof.get(("score", BATCH), [])
        roof = None
        if qm:
            avg_s = float(np.mean([d for d, _ in qm]))
            avg_c = float(np.mean([c for _, c in qm]))
            flops = qmc_pairs(BATCH, avg_c) * FLOP_PER_PAIR
            ach = flops / avg_s / 1e12
            roof = {"bound": "fp64-valu", "kernel": "score_qmc_kernel<%d>" % BATCH, "achieved": ach,
                    "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
                    "traffic": pmc_traffic("void ital::score_qmc_kernel<%d>" % BATCH), "avg_launch_ms": avg_s * 1e3,
                    "pairs_per_s": qmc_pairs(BATCH, avg_c) / avg_s,
                    "flop_per_pair": FLOP_PER_PAIR, "valu_per_pair_isolated_chain": VALU_PER_PAIR_CHAIN,
                    "valu_issue_frac": pmc_valu_issue_frac("void ital::score_qmc_kernel<%d>" % BATCH, avg_s),
                    "note": "transcendental FP64 chains (Phi, Phi^-1): neither HBM nor MFMA bounds this kernel "
                            "(SURVEY.md 8d S-qmc), so the peak is the FP64 vector rate; achieved = algorithmic pairs x "
                            "flops of the isolated chain / launch time.  Only ~55 % of the chain's instructions are "
                            "FMAs, so the flop fraction understates how busy the vector unit is: valu_issue_frac is "
                            "the share of its issue slots the kernel fills.  HBM-bound streaming kernel in roofline_hbm"}
        cc = prof.get(("cross_cov", 1), []) + prof.get(("cross_cov", 2), []) + prof.get(("cross_cov", 3), [])
        roof_hbm = None
        if cc:
            avg_s = float(np.mean([d for d, _ in cc]))
            m_avg = float(np.mean([c for _, c in cc]))
            bytes_alg = learner.gp.n * 8.0 * (DIM + m_avg + 1)
            ach = bytes_alg / avg_s / 1e9
            roof_hbm = {"bound": "hbm", "kernel": "kcols_kernel (cross-covariance column)", "achieved": ach,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_ms": avg_s * 1e3,
                        "note": "19 MB per launch at this size: launch-latency bound, see DESIGN.md for the large-N figure"}
        out = {"metric": "MI-scored candidates/sec per fetch_unlabelled(k) round", "value": scored / dt,
               "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "USPS-shaped synthetic %dx%d per GPU, k=%d, perfect user, full 2^t enumeration, "
                                      "fetch_unlabelled + update per step" % (ROWS_PER_GPU, DIM, BATCH),
                          "n": n_total, "d": DIM, "k": BATCH, "length_scale": LENGTH_SCALE,
                          "parallelism": "candidate rows sharded over %d GPU(s), 1 record all-gather per greedy step" % world},
               "roofline": roof, "roofline_hbm": hbm_stream_probe(device) if world == 1 else roof_hbm,
               "roofline_hbm_at_workload_size": roof_hbm,
               "kernel_ms": {"%s_t%d" % k: float(np.mean([d for d, _ in v])) * 1e3 for k, v in sorted(prof.items())}}
        if world == 1 and not os.environ.get("ITAL_BENCH_NO_EXTRAS"):
            out["other_workloads"] = other_workloads(X, rel, device)
        out["cpu_baseline"] = cpu_base
        if cpu_base:
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu_base["value"]
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
