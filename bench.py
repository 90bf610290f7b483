#!/usr/bin/env python3
"""Benchmark of the ITAL hot path on MI355X: MI-scored candidates/sec per fetch_unlabelled(k) round.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one round of the reference's retrieval loop (run_experiment.py:160-164): fetch_unlabelled(k) on the
current relevance model followed by update() with the simulated feedback for the fetched batch (the update is
inside the timed region: nothing is skipped).

Headline (`value`): BASELINE.json configs[1], SURVEY.md section 8d C2': synthetic USPS-shaped features, 9298 x 256 fp64 in
[0,1], length_scale 3.0, k = 4, perfect user; with N > 1 every rank holds 9298 rows (weak scaling) and the greedy steps
exchange one record per rank over RCCL.  A scored candidate = one (candidate, greedy step) MI evaluation with full 2^t
sign-pattern enumeration.

Every line also carries `scaling_workload`: the curve north_star asks for -- 1 000 000 x 512, k = 4, full enumeration,
the rows STRONG-scaled over the N ranks (each rank generates and holds only its own row block), timed the same way
(barrier + synchronize on both sides, maximum over ranks), with the per-step record exchange priced separately.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the pool's host driver only supports dmabuf IPC: RCCL between processes needs this before the HIP runtime starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROWS_PER_GPU = 9298
DIM = 256
BATCH = 4
LENGTH_SCALE = 3.0
SCALE_ROWS, SCALE_DIM, SCALE_BATCH = 1_000_000, 512, 4
FP64_VALU_PEAK_TFLOPS = 78.6   # MI355X vector FP64 (= half the 157.3 TF FP32 vector rate of MI355X_MICROARCH.md)
FP64_MFMA_PEAK_TFLOPS = 78.6   # MI355X matrix FP64 (MI355X_MICROARCH.md: same rate as the vector unit on gfx950)
HBM_PEAK_GBS = 8000.0
PRIMES = (31, 47, 73, 113, 173, 263, 397, 593, 907, 1361)
# FP64 work of one (Phi, Phi^-1) pair of the lattice integrand in flops (FMA = 2, multiply / add = 1), counted in the ISA
# of the isolated chain (tools/ubench/phi_ubench.hip): Hart's Phi 53 flops (22 FMA + 9 mul/add among 47 vector
# instructions), AS241's central Phi^-1 48 (21 + 6 among 45), its log/sqrt tail branch 126 (56 + 14 among 121) for the
# 15 % of the arguments that need it.  The chain, not the kernel: bookkeeping the kernel adds is not achieved work.
FLOP_PER_PAIR = 53 + 0.85 * 48 + 0.15 * 126
VALU_PER_PAIR_CHAIN = 47 + 0.85 * 45 + 0.15 * 121      # vector instructions of the isolated chain per pair
# flops of one term of the MCMI objective (Phi: 53, two logs: 2 x 44, products / sums: 6), counted the same way
FLOP_PER_MCMI_TERM = 53 + 2 * 44 + 6
# committed counter summaries (rocprofv3 --pmc passes of this very command, tools/profile_r6.sh + tools/pmc_summary.py):
# HBM traffic and instruction counts per launch are read from these files and the file is named in the output
PMC_FILES = {"headline": "profiles/r6_headline_pmc_summary.csv", "general": "profiles/r6_general_pmc_summary.csv",
             "k8": "profiles/r6_k8_pmc_summary.csv", "mcmi": "profiles/r6_mcmi_pmc_summary.csv",
             "kcols": "profiles/r6_kcols_pmc_summary.csv", "c5": "profiles/r6_c5_pmc_summary.csv"}
ROUND_GAPS_FILE = "profiles/r6_round_gaps.json"   # launches / busy fraction of a round out of a committed kernel trace
CALIBRATION_FILE = "profiles/r6_oracle_calibration.json"
STAMP_FILE = "profiles/r6_stamp.json"             # kernel sources each committed profile was taken with (tools/stamp.py)
PICKS_N1_FILE = "profiles/scaling_picks_n1.json"   # the batches the N = 1 run of scaling_workload picks (bench.py wrote it on one GPU)


def make_data(n, d, seed):
    rng = np.random.default_rng(seed)
    return rng.random((n, d))


def block_rows(row0, row1, d, seed, block=65536):
    """Rows [row0, row1) of the synthetic n x d matrix whose row block b (65536 rows) is default_rng([seed, b]).random():
    every rank can generate exactly its own rows, whatever the number of ranks."""
    out = np.empty((row1 - row0, d))
    b = row0 // block
    while b * block < row1:
        lo, hi = max(row0, b * block), min(row1, (b + 1) * block)
        blk = np.random.default_rng([seed, b]).random((block, d))
        out[lo - row0:hi - row0] = blk[lo - b * block:hi - b * block]
        b += 1
    return out


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
    pm = pmc_fields("kcols", "ital::kcols_kernel", sec)      # counters of this very probe (tools/kcols_probe.py under rocprofv3)
    pm.pop("valu_issue_frac", None)
    return dict({"bound": "hbm", "kernel": "kcols_kernel (ital_cross_cov_cols, c=1)", "achieved": ach, "peak": HBM_PEAK_GBS,
                 "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": sec * 1e3,
                 "rows": rows, "d": d, "m": m, "algorithmic_bytes_per_launch": nbytes,
                 "note": "same kernel as the greedy steps' cross-covariance column, at 1M rows (2.2 GB per launch); traffic = "
                         "FETCH_SIZE (corrected) + WRITE_SIZE per launch from the committed counter pass of this probe"}, **pm)


def profile_is_current(rel_path):
    """(True, stamp) when the committed profile `rel_path` was taken with the kernel sources of this tree (tools/stamp.py:
    sha256 over ital_amd/csrc/* and the header); (False, why) otherwise -- its numbers are then not quoted."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import stamp
        path = os.path.join(ROOT, STAMP_FILE)
        if not os.path.exists(path):
            return False, "no stamp file %s" % STAMP_FILE
        with open(path) as f:
            entry = json.load(f).get(rel_path)
        if entry is None:
            return False, "%s has no stamp in %s" % (rel_path, STAMP_FILE)
        now = stamp.csrc_sha(entry.get("units"))      # the translation units the profiled workload runs (all of them: r4 stamps)
        if entry.get("csrc_sha") != now:
            return False, "taken with other kernel sources (csrc_sha %s, this tree %s): re-run tools/profile_r6.sh" % (entry.get("csrc_sha"), now)
        return True, entry
    finally:
        sys.path.pop(0)


def pmc_row(which, kernel_prefix):
    """Row of a kernel in the committed PMC summary named by PMC_FILES[which] (collected with tools/profile_r6.sh in
    separate passes and corrected as MI355X_MICROARCH.md prescribes), or None."""
    import csv
    path = os.path.join(ROOT, PMC_FILES[which])
    if not os.path.exists(path):
        return None
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row["kernel"].startswith(kernel_prefix):
                return row
    return None


def pmc_fields(which, kernel_prefix, launch_s):
    """traffic (HBM bytes per launch, fetch + write), share of the chip's vector-issue cycles the kernel's VALU instructions
    fill (a wave64 instruction holds its SIMD's 16 lanes for 4 cycles; 256 CUs x 4 SIMDs at 2.4 GHz; count from the
    committed pass, launch time live), and the name of the file both came from."""
    row = pmc_row(which, kernel_prefix)
    out = {"traffic": None, "valu_issue_frac": None, "pmc_file": PMC_FILES[which] if row else None}
    if row:
        ok, info = profile_is_current(PMC_FILES[which])
        out["pmc_stamp"] = info
        if not ok:
            return dict(out, pmc_file=None, pmc_note="counters not quoted: " + str(info))
        try:
            out["traffic"] = float(row["fetch_bytes_corrected_avg"]) + float(row["write_bytes_avg"])
        except (KeyError, ValueError):
            pass
        try:
            out["valu_issue_frac"] = float(row["SQ_INSTS_VALU_avg"]) * 4.0 / (1024 * 2.4e9 * launch_s)
        except (KeyError, ValueError):
            pass
    return out


def round_gaps():
    """GPU-busy fraction and launches of one headline round out of the committed kernel trace summary (tools/round_gaps.py
    on a rocprofv3 --kernel-trace run of this command), or None."""
    path = os.path.join(ROOT, ROUND_GAPS_FILE)
    if not os.path.exists(path):
        return None
    ok, info = profile_is_current(ROUND_GAPS_FILE)
    if not ok:
        return {"file": None, "note": "not quoted: " + str(info)}
    with open(path) as f:
        return dict(json.load(f), file=ROUND_GAPS_FILE, stamp=info)


def picks_digest(picks):
    """sha256 (first 16 hex digits) of the rounds' picks in order, and the four 64-bit words of it that ranks compare."""
    import hashlib
    h = hashlib.sha256(np.asarray(picks, dtype=np.int64).tobytes()).digest()
    return h.hex()[:16], np.frombuffer(h, dtype=np.int64).copy()


def ranks_agree(words, device, world):
    """True when every rank holds the same digest words (all-gather of 4 x int64 through torch.distributed)."""
    if world == 1:
        return True
    import torch
    import torch.distributed as dist
    mine = torch.from_numpy(words).to(device)
    got = [torch.empty_like(mine) for _ in range(world)]
    if dist.get_backend() == "gloo":
        got = [g.cpu() for g in got]
        dist.all_gather(got, mine.cpu())
        return all(bool(torch.equal(g, mine.cpu())) for g in got)
    dist.all_gather(got, mine)
    return all(bool(torch.equal(g, mine)) for g in got)


def gather_floats(value, device, world):
    """The ranks' values of one float, in rank order (on every rank)."""
    if world == 1:
        return [float(value)]
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if dist.get_backend() == "gloo":
        got = [torch.empty(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(got, t.cpu())
    else:
        got = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(got, t)
    return [float(g.item()) for g in got]


def split_rounds(learner, label, k, rounds, barrier):
    """fetch_unlabelled(k) and update() timed SEPARATELY ("fit / update reported separately"): `rounds` more
    rounds, untimed for the headline, with the device drained around each call -- so the two do not overlap as they do in
    the timed loop (the update's launches normally run under the host's bookkeeping).  Collective on several ranks."""
    import torch
    tf = tu = 0.0
    for _ in range(rounds):
        barrier()
        t0 = time.perf_counter()
        ret = learner.fetch_unlabelled(k)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        learner.update({int(i): label(int(i)) for i in ret})
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        tf += t1 - t0
        tu += t2 - t1
    return {"fetch_ms_per_round": tf / rounds * 1e3, "update_ms_per_round": tu / rounds * 1e3, "rounds": rounds,
            "note": "drained device around each call (not the timed loop: there update() overlaps the host's bookkeeping)"}


def transport_report(learner, group, device):
    """Which way the records of a greedy step travel on this rank, and what RCCL says about the communicator."""
    import ctypes
    from ital_amd import _lib, sharding
    if not learner.gp.collective:
        return {"transport": None}
    import torch.distributed as dist
    kind = learner._round_transport()
    out = {"transport": {"nccl": "raw_nccl", "host": "host"}[kind[0]] if kind else "torch_dist",
           "round_as_one_call": bool(kind) and learner.round_call, "group_world_size": dist.get_world_size(group)}
    if kind and kind[0] == "nccl":
        w, r = ctypes.c_int(-1), ctypes.c_int(-1)
        how = ctypes.create_string_buffer(600)
        if _lib.lib().ital_exchange_info(kind[1], ctypes.byref(w), ctypes.byref(r), how, 600) == 0:
            out.update(rccl_world_size=w.value, rccl_rank=r.value, rccl_found=how.value.decode())
    else:
        out["why_not_raw_nccl"] = sharding.raw_comm_reason(group, device)
    return out


def other_workloads(X, rel, device):
    """Secondary timings on the same synthetic data (rank 0, N = 1 only; not part of `value`): the general scorer with a
    noisy user (reference configs usps-mistakes / mirflickr-mistakes style) and MCMI_min with the reference's subsample,
    each with the roofline of its dominant kernel."""
    import torch
    from ital_amd import ITAL, MCMI_min, mvn_stream
    out = {}

    def timed(learner, rounds, k, warm=2, sample_rounds=0):
        """sample_rounds > 0: the timed rounds run without kernel events (a learner that enqueues its whole round in one
        call below the C ABI only does so unprofiled); the per-kernel times come from that many extra rounds afterwards."""
        learner.update({0: 1})
        for _ in range(warm):                             # warm-up rounds (the second one still loads code: lazily
            ret = learner.fetch_unlabelled(k)             # initialised torch kernels of the update path, 12-50 ms once)
            learner.update({int(i): float(rel[i]) for i in ret})
        if getattr(learner, "pair_counter", None) is not None:
            learner.pair_counter.zero_()                  # count the timed rounds only
        learner.profile = None if sample_rounds else []
        learner.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(6 * k * (sample_rounds or rounds) + 16)]   # (a large pool of recorded events slows the first rounds down, see main())
        for ev in learner.event_pool:
            ev.record()
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
        if sample_rounds:
            learner.profile = []
            for _ in range(sample_rounds):
                ret = learner.fetch_unlabelled(k)
                learner.update({int(i): float(rel[i]) for i in ret})
            torch.cuda.synchronize()
        prof = {}
        for name, t, n_c, e0, e1 in learner.profile:
            prof.setdefault((name, t), []).append((e0.elapsed_time(e1) * 1e-3, n_c))
        return {"ms_per_round": dt / rounds * 1e3, "candidates_per_s": scored / dt}, prof

    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=LENGTH_SCALE, label_prob=0.5, mistake_prob=0.25, device=device)
    L.pair_counter = torch.zeros(1, dtype=torch.int64, device=device)
    res, prof = timed(L, 6, BATCH, warm=2)
    top = prof.get(("score_generic", BATCH), [])
    roof = None
    if top:
        # pairs of the t = 4 launches: the counter runs over all steps; the closed forms of t <= 2 add none, t = 3 is
        # priced by its own launches' share of the scorer time (same kernel, same rate)
        sec4 = float(np.mean([d for d, _ in top]))
        sec_all = sum(d for key, v in prof.items() if key[0] == "score_generic" and key[1] >= 3 for d, _ in v)
        pairs_all = float(L.pair_counter.item())
        rate = pairs_all / sec_all if sec_all > 0 else 0.0
        ach = rate * FLOP_PER_PAIR / 1e12
        roof = dict({"bound": "fp64-valu", "kernel": "gen_main_kernel<%d> (lattice sums of ital_score_generic)" % BATCH,
                     "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS,
                     "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS, "avg_launch_ms": sec4 * 1e3,
                     "pairs_per_s": rate, "pairs_counted_on_device": pairs_all,
                     "note": "pairs = lattice points x (n - 1) of the calls that are really integrated (counted by the kernel, "
                             "ital_gscore_desc.pair_count: 240 of the 1296 calls per candidate at t = 4); time = the whole "
                             "ital_score_generic step (preparation of all 1296 calls, lattice sums, combine; the preparation "
                             "runs under the lattice sums on a second stream); avg_launch_ms = one t = 4 step"},
                    **dict(pmc_fields("general", "void ital::gen_main_kernel<%d," % BATCH, sec4), valu_issue_frac=None,
                           traffic_note="traffic: per launch of gen_main_kernel, one of the ~12 slab launches of a step"))
    out["ital_general_user_k4"] = dict(res, roofline=roof,
                                       config="label_prob 0.5, mistake_prob 0.25: 3^t - 1 feedback configurations per pattern")
    del L
    out["ital_k8_25000x512"] = k8_workload(device)
    out["ital_ce_subset5_iris_shaped_150x4"] = cesub_workload(device, 150, 4, rounds=5, length_scale=0.5)
    out["ital_ce_subset5_9298x256"] = cesub_workload(device, ROWS_PER_GPU, DIM, length_scale=LENGTH_SCALE)
    np.random.seed(0)
    m = MCMI_min(X, length_scale=LENGTH_SCALE, subsample=1000, device=device)
    r, prof = timed(m, 20, BATCH, sample_rounds=5)
    r["candidates_per_s"] = BATCH * 1000 / (r["ms_per_round"] * 1e-3)
    roofs = {}
    cb = prof.get(("cov_block", 0), [])
    if cb:
        sec = float(np.mean([d for d, _ in cb]))
        nc = float(np.mean([c for _, c in cb]))
        flops = 2.0 * nc * nc * (DIM + m.gp.m)
        ach = flops / sec / 1e12
        roofs["cov_block_kernel"] = dict({"bound": "mfma", "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                          "frac": ach / FP64_MFMA_PEAK_TFLOPS, "avg_launch_ms": sec * 1e3,
                                          "note": "1000 x 1000 block: 0.55 GFLOP per launch, launch-latency bound at the "
                                                  "reference's subsample; blocks of >= 1024 tiles of 128^2 take the LDS-staged "
                                                  "kernel: 45 TFLOP/s at 9273^2 x 256, 60 at 20000^2 x 512 "
                                                  "(tools/cov_bench.py; other_workloads.cov_block_20000x512; DESIGN.md section 4)",
                                          "traffic": None})
    ms = prof.get(("mcmi_score", BATCH), [])
    if ms:
        sec = float(np.mean([d for d, _ in ms]))
        nc = float(np.mean([c for _, c in ms]))
        terms = nc * nc * (2 ** BATCH)
        ach = terms * FLOP_PER_MCMI_TERM / sec / 1e12
        roofs["mcmi_score_kernel<%d>" % BATCH] = dict({"bound": "fp64-valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS,
                                                       "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
                                                       "avg_launch_ms": sec * 1e3, "terms_per_s": terms / sec},
                                                      **pmc_fields("mcmi", "void ital::mcmi_score_kernel<%d>" % BATCH, sec),
                                                      **{"pmc_note": "counters per launch of tools/mcmi_bench.py 1000 (this problem size "
                                                                     "alone: tools/profile_r6.sh mcmi), quoted only while their stamp matches"})
    out["mcmi_min_subsample1000_k4"] = dict(r, roofline=roofs, config="MCMI_min, subsample 1000 (reference configs/usps.conf)")
    # batches of 6 (reference configs/toy*.conf use batch_size = 6): the split scorer (preparation + workgroup per
    # candidate and group of 8 label patterns)
    k6 = 6
    np.random.seed(0)
    m6 = MCMI_min(X, length_scale=LENGTH_SCALE, subsample=1000, device=device)
    r6, prof6 = timed(m6, 5, k6, sample_rounds=3)
    r6["candidates_per_s"] = k6 * 1000 / (r6["ms_per_round"] * 1e-3)
    ms6 = prof6.get(("mcmi_score", k6), [])
    roof6 = None
    if ms6:
        sec = float(np.mean([d for d, _ in ms6]))
        nc = float(np.mean([c for _, c in ms6]))
        terms = nc * nc * (2 ** k6)
        ach = terms * FLOP_PER_MCMI_TERM / sec / 1e12
        roof6 = {"kernel": "mcmi_prep_kernel<6> + mcmi_split_kernel<6>", "bound": "fp64-valu", "achieved": ach,
                 "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
                 "avg_launch_ms": sec * 1e3, "terms_per_s": terms / sec, "traffic": None}
    out["mcmi_min_subsample1000_k6"] = dict(r6, roofline=roof6, kernel_ms={"%s_t%d" % key: float(np.mean([d for d, _ in v])) * 1e3
                                                                           for key, v in sorted(prof6.items())},
                                            config="MCMI_min, subsample 1000, batch of 6")
    return out


def cesub_workload(device, n, d, k=4, subset=5, rounds=3, length_scale=None, seed=5):
    """The change-estimation subset (reference ital.py:103-108, :227-275 `_call_iter_sub`; BASELINE.json configs[0] is the
    Iris-shaped case): every orthant spans the subset + the batch + the candidate (up to 5 + 4 + 1 variables here), one wave
    per candidate in the monolithic score_generic_kernel -- the path no other workload of this file times."""
    import torch
    from ital_amd import ITAL, mvn_stream
    X = make_data(n, d, seed=seed)
    rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
    mvn_stream.GLOBAL.reset()
    np.random.seed(0)
    L = ITAL(X, length_scale=length_scale or float(np.sqrt(d / 12.0)), change_estimation_subset=subset, device=device)
    L.update({0: 1, 1: -1})
    ret = L.fetch_unlabelled(k)                                  # warm-up round (code load)
    L.update({int(i): float(rel[i]) for i in ret})
    L.profile = []
    L.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(8 * k * rounds + 16)]
    for ev in L.event_pool:
        ev.record()
    torch.cuda.synchronize()
    scored = 0
    t0 = time.perf_counter()
    for _ in range(rounds):
        n_c = len(L.get_unseen())
        ret = L.fetch_unlabelled(k)
        L.update({int(i): float(rel[i]) for i in ret})
        scored += sum(n_c - t for t in range(k))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = {}
    for name, t, n_c, e0, e1 in L.profile:
        prof.setdefault((name, t), []).append(e0.elapsed_time(e1))
    res = {"ms_per_round": dt / rounds * 1e3, "candidates_per_s": scored / dt, "rounds": rounds,
           "config": "synthetic %d x %d, k=%d, change_estimation_subset=%d, perfect user (reference ital.py:227-275)" % (n, d, k, subset),
           "kernel_ms": {"%s_t%d" % key: float(np.mean(v)) for key, v in sorted(prof.items())}}
    del L
    gc.collect()
    torch.cuda.empty_cache()
    return res


def k8_workload(device, n=25000, d=512, k=8):
    """BASELINE.json configs[2] / SURVEY.md 8d C3': 25 000 x 512, batches of 8, perfect user, full 2^t enumeration -- one
    fetch + update round after a warm-up round, with the roofline of its dominant kernel (the t = 8 lattice sums)."""
    import torch
    from ital_amd import ITAL, mvn_stream
    X = make_data(n, d, seed=3)
    rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=float(np.sqrt(d / 12.0)), device=device)
    L.update({0: 1})
    ret = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in ret})
    L.profile = []
    L.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(8 * k)]
    for ev in L.event_pool:
        ev.record()
    torch.cuda.synchronize()
    n_c = len(L.get_unseen())
    t0 = time.perf_counter()
    ret = L.fetch_unlabelled(k)
    L.update({int(i): float(rel[i]) for i in ret})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scored = sum(n_c - t for t in range(k))
    prof = {(name, t): (e0.elapsed_time(e1) * 1e-3, c) for name, t, c, e0, e1 in L.profile}
    res = {"ms_per_round": dt * 1e3, "candidates_per_s": scored / dt,
           "config": "synthetic %d x %d, k=%d, perfect user, full 2^t enumeration (BASELINE.json configs[2], SURVEY 8d C3')" % (n, d, k),
           "kernel_ms": {"%s_t%d" % key: v[0] * 1e3 for key, v in sorted(prof.items())}}
    roofs = {}
    for (name, t), (sec, n_cand) in sorted(prof.items()):
        if t < 7 or not name.startswith("qmc_"):
            continue
        # the candidates of a step are walked in slabs of the 1 GiB workspace (150 KB of prepared calls per candidate at
        # t = 8): the event pair spans first .. last lattice-sum launch of the step
        slabs = int(name[len("qmc_slabs"):]) if name.startswith("qmc_slabs") else 1
        pairs = qmc_pairs(t, n_cand)
        ach = pairs * FLOP_PER_PAIR / sec / 1e12
        pm = pmc_fields("k8", "void ital::qmc_main_kernel<%d>" % t, sec / slabs)
        roofs["qmc_main_kernel<%d>" % t] = dict(
            {"bound": "fp64-valu", "achieved": ach, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
             "frac": ach / FP64_VALU_PEAK_TFLOPS, "launches_per_step": slabs, "avg_launch_ms": sec * 1e3 / slabs,
             "step_ms": sec * 1e3, "pairs_per_s": pairs / sec, "algorithmic_pairs_per_step": pairs,
             "note": "time = first to last lattice-sum launch of the step (HIP events recorded by the library), incl. the "
                     "preparation / combine launches between the slabs (~0.5 %); counters per launch from the committed pass"},
            **pm)
    res["roofline"] = roofs
    del L
    gc.collect()
    torch.cuda.empty_cache()
    return res


def topcand_workload(device, n=1_000_000, d=512, k=4, top=4096, rounds=3):
    """`top_candidates` at scale (reference ital.py:111-117, the shipped *-topscoring.conf): every fetch restricts the candidate
    list to the `top` samples of largest predictive mean -- np.argpartition over ALL unlabelled samples on the host, whose
    ORDER is the list's order (tie-breaks, stream offsets) and therefore stays numpy's -- and scores only those.  Timed: the
    whole fetch + update round and the host's share of it (means download + argpartition), the number the round-4 verdict
    found missing at 1M rows."""
    import torch
    from ital_amd import ITAL, mvn_stream
    X = block_rows(0, n, d, seed=1)
    mvn_stream.GLOBAL.reset()
    L = ITAL(X, length_scale=float(np.sqrt(d / 12.0)), top_candidates=top, device=device)
    L.update({0: 1})
    inner = L._candidate_list
    host = [0.0]

    def timed_list(*a, **kw):
        t_ = time.perf_counter()
        out_ = inner(*a, **kw)
        host[0] += time.perf_counter() - t_
        return out_
    L._candidate_list = timed_list

    def label(i):
        return 1.0 if (i * 2654435761) % (1 << 32) < (1 << 31) else -1.0
    ret = L.fetch_unlabelled(k)                       # warm-up
    L.update({int(i): label(int(i)) for i in ret})
    host[0] = 0.0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rounds):
        ret = L.fetch_unlabelled(k)
        L.update({int(i): label(int(i)) for i in ret})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"ms_per_round": dt / rounds * 1e3, "host_candidate_list_ms_per_round": host[0] / rounds * 1e3,
           "candidates_per_s": rounds * sum(top - t for t in range(k)) / dt,
           "config": "synthetic %d x %d, k=%d, top_candidates=%d, perfect user: the list of a fetch = the %d samples of largest "
                     "predictive mean in np.argpartition's order" % (n, d, k, top, top)}
    del L, X
    gc.collect()
    torch.cuda.empty_cache()
    return res


def k16_workload(device, n=1_000_000, d=512, k=16, mc=1):
    """BASELINE.json configs[4] / SURVEY.md 8d C5' on ONE GPU: n x 512 (1 000 000: the configuration in one piece; 125 000: the
    share one of 8 ranks holds), batches of 16, monte_carlo_num_rel = 1 (2^16 sign patterns are infeasible anywhere: the
    reference's own switch, ital.py:293-297) -- one fetch + update round, with the roofline of its lattice sums (the general
    scorer's pipeline, 3 .. 16 variables)."""
    import torch
    from ital_amd import ITAL, mvn_stream
    torch.cuda.reset_peak_memory_stats(device)
    X = block_rows(0, n, d, seed=1)
    mvn_stream.GLOBAL.reset()
    np.random.seed(0)
    L = ITAL(X, length_scale=float(np.sqrt(d / 12.0)), monte_carlo_num_rel=mc, device=device)
    L.pair_counter = torch.zeros(1, dtype=torch.int64, device=device)
    L.update({0: 1})
    L.profile = []
    L.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(8 * k)]
    for ev in L.event_pool:
        ev.record()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ret = L.fetch_unlabelled(k)
    t1 = time.perf_counter()
    L.update({int(i): (1.0 if (int(i) * 2654435761) % (1 << 32) < (1 << 31) else -1.0) for i in ret})
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    scored = sum(n - 1 - t for t in range(k))
    steps = {t: e0.elapsed_time(e1) * 1e-3 for name, t, c, e0, e1 in L.profile if name == "score_generic"}
    sec = sum(v for t, v in steps.items() if t >= 3)
    pairs = float(L.pair_counter.item())
    ach = pairs * FLOP_PER_PAIR / sec / 1e12 if sec > 0 else 0.0
    made, skipped, walk_s = L.mc_walk
    res = {"ms_per_round": dt * 1e3, "fetch_s": t1 - t0, "candidates_per_s": scored / dt,
           "config": "synthetic %d x %d, k=%d, perfect user, monte_carlo_num_rel=%d (BASELINE.json configs[4], SURVEY 8d C5'%s), "
                     "one GPU" % (n, d, k, mc, "" if n >= 1_000_000 else ": the rows one of 8 ranks holds"),
           "step_ms": {"t%d" % t: v * 1e3 for t, v in sorted(steps.items())},
           "pattern_sampling": {"standard_normals_computed": made, "skipped": skipped, "host_s": walk_s,
                                "note": "numpy's legacy generator walked in the reference's order (ital_np_legacy_normals), under "
                                        "the scorer of the step before"},
           "peak_device_memory_gib": torch.cuda.max_memory_allocated(device) / 2 ** 30,
           "roofline": {"bound": "fp64-valu", "kernel": "gen_main_kernel<3..16> (lattice sums of ital_score_generic)", "achieved": ach,
                        "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
                        "pairs_counted_on_device": pairs, "seconds_steps_3_to_16": sec,
                        "note": "pairs = lattice points x (n - 1) of the calls that are integrated, counted by the kernels; time = "
                                "the ital_score_generic steps of 3 .. 16 variables whole (preparation, lattice sums, combine, the "
                                "host's pattern uploads between the ranges of a step)",
                        **c5_counters()}}
    del L, X
    gc.collect()
    torch.cuda.empty_cache()
    return res


def cov_block_workload(device, n=20000, d=512, m=17, reps=5):
    """The dense posterior-covariance block of the MCMI / EMOC paths at a size that fills the chip (reference ital/mcmi.py:101-124
    via gp.py:128: K_all[S, S] - k^T K^-1 k): ONE ital_cov_block of n x n x d on the FP64 matrix cores
    (v_mfma_f64_16x16x4_f64, LDS-staged 128 x 128 tiles: cov_block_lds_kernel), timed with HIP events on the launch stream;
    2 n^2 (d + m) flops per launch against the FP64 MFMA peak.  (At the reference's MCMI subsample of 1000 the same call is
    launch-latency bound: other_workloads.mcmi_min_subsample1000_k4.)"""
    import torch
    from ital_amd import _lib
    from ital_amd.gp import _ptr, _stream
    lib = _lib.lib()
    ldx = (d + 15) // 16 * 16
    with torch.cuda.device(device):
        g = torch.Generator(device="cpu").manual_seed(n)
        X = torch.zeros(n, ldx, dtype=torch.float64)
        X[:, :d] = torch.rand(n, d, generator=g, dtype=torch.float64)
        X = X.to(device)
        xn = (X * X).sum(1)
        V = (0.05 * torch.randn(m, n, generator=g, dtype=torch.float64)).to(device)
        out = torch.empty((n, n), dtype=torch.float64, device=device)
        ls, var = float((d / 12.0) ** 0.5), 1.0
        st = _stream()

        def call():
            _lib.check(lib.ital_cov_block(_ptr(X), _ptr(xn), n, _ptr(X), _ptr(xn), n, ldx, _ptr(V), n, _ptr(V), n, m, var, ls,
                                          _ptr(out), n, st))
        call()
        torch.cuda.synchronize()
        # spot check against the dense formula (256 x 256 corner)
        a = slice(0, 256)
        want = var * torch.exp(-(xn[a, None] + xn[None, a] - 2 * X[a] @ X[a].T) / (2 * ls * ls)) - V[:, a].T @ V[:, a]
        err = float((out[a, a] - want).abs().max().item())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        sec = e0.elapsed_time(e1) * 1e-3 / reps
        del out, X, V
    gc.collect()
    torch.cuda.empty_cache()
    flops = 2.0 * n * n * (ldx + m)
    ach = flops / sec / 1e12
    return {"ms_per_launch": sec * 1e3, "max_abs_err_vs_dense_formula": err,
            "config": "one ital_cov_block of %d x %d x %d (m = %d whitened rows), fp64" % (n, n, d, m),
            "roofline": {"bound": "mfma", "kernel": "cov_block_lds_kernel (v_mfma_f64_16x16x4_f64)", "achieved": ach,
                         "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                         "avg_launch_ms": sec * 1e3, "algorithmic_flops_per_launch": flops, "traffic": None,
                         "note": "2 n^2 (d + m) flops / launch time (HIP events around %d launches on the launch stream); the "
                                 "block's output alone is %.1f GB written per launch" % (reps, n * n * 8 / 1e9)}}


def c5_counters():
    """HBM traffic and vector-issue share of the widest lattice-sum kernel out of the committed counter pass of BASELINE config 5's
    125 000-row share (tools/profile_r6.sh c5: what one of 8 ranks runs), per launch; null when the stamp does not match."""
    import csv
    name = "void ital::gen_main_kernel<16"
    row = pmc_row("c5", name)
    out = {"traffic": None, "valu_issue_frac": None}
    if not row:
        return out
    ok, info = profile_is_current(PMC_FILES["c5"])
    if not ok:
        return dict(out, pmc_note="counters not quoted: " + str(info))
    stats = os.path.join(ROOT, "profiles", "r6_c5_kernel_stats.csv")
    avg_ns = None
    if os.path.exists(stats):
        with open(stats, newline="") as f:
            for r_ in csv.DictReader(f):
                if r_.get("Name", "").startswith(name):
                    avg_ns = float(r_["AverageNs"])
    out["traffic"] = float(row["fetch_bytes_corrected_avg"]) + float(row["write_bytes_avg"])
    if avg_ns:
        out["valu_issue_frac"] = float(row["SQ_INSTS_VALU_avg"]) * 4.0 / (1024 * 2.4e9 * avg_ns * 1e-9)
        out["avg_launch_ms_in_profile"] = avg_ns * 1e-6
    out["pmc_file"] = PMC_FILES["c5"]
    out["traffic_note"] = ("gen_main_kernel<16> per launch in the profile of the 125 000-row share (a range of a step: 8 - 67 k "
                           "candidates x 16 calls); the records of a launch are read once (1.8 KB per call)")
    return out


def cpu_cores():
    """Host cores the CPU baselines run on: what this process may really use (affinity mask, cgroup quota), not the host's
    core count -- a 16-CPU share of a 256-core host runs 256 workers no faster than 16."""
    from oracle.parallel import effective_cores
    return effective_cores()


def _calibration():
    path = os.path.join(ROOT, CALIBRATION_FILE)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        cal = json.load(f)
    return {"oracle_over_reference_time": cal["oracle_over_reference"], "file": CALIBRATION_FILE,
            "note": "oracle vs the real reference on identical inputs in the build container (%s): "
                    "the port is a fair stand-in for the reference, which cannot run on the GPU box" % cal["workload"]}


def cpu_baseline(X, cores):
    """The oracle (CPU restatement of the reference, oracle/) in the reference's parallel mode on the headline workload."""
    from oracle.ital import OracleITAL
    from oracle.parallel import fetch_unlabelled_parallel
    n = int(min(len(X), max(512, 600 * cores)))
    times = []
    for _ in range(3):                      # three samples of the same round (fresh learner each): the median is `value`
        learner = OracleITAL(X[:n], length_scale=LENGTH_SCALE)
        learner.update({0: 1})
        t0 = time.time()
        _, scored = fetch_unlabelled_parallel(learner, BATCH, processes=cores)
        times.append(time.time() - t0)
    dt = sorted(times)[1]
    out = {"value": scored / dt, "unit": "candidates/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
           "samples_candidates_per_s": [scored / t_ for t_ in times],
           "sample": "fetch_unlabelled(%d) on the first %d rows of the workload (%d scored candidates), three times: %s s, "
                     "value = the median; fork pool of %d workers per greedy step as reference ital/ital.py:124-126"
                     % (BATCH, n, scored, " / ".join("%.1f" % t_ for t_ in times), cores)}
    cal = _calibration()
    if cal:
        out["calibration"] = cal
    return out


def cpu_baseline_sampled(n_workload, d, k, cores, kw=None, length_scale=None, seed=3, budget_s=15.0, n_sub=2048):
    """CPU baseline of a configuration whose full round would take the host hours (C3' - C5' "timed on a 2048-
    candidate subsample ... flagged extrapolated"; the per-candidate cost of the reference's scorer does not depend on N,
    ital/ital.py:183-224, and its dense N x N kernel matrix could not even be formed at these sizes, gp.py:128): the oracle in
    the reference's parallel scheme on the first n_sub rows of a matrix of the workload's shape -- same d, k, user model,
    m = 1 labelled sample -- with per greedy step as many candidates as the step's share of `budget_s` allows
    (oracle.parallel.fetch_unlabelled_sampled: from t = 6 on that is fewer than n_sub), extrapolated to the workload's N:
    round time = sum_t [pool start-up + (N - t + 1) x measured seconds per candidate of step t]."""
    from oracle.ital import OracleITAL
    from oracle.parallel import fetch_unlabelled_sampled
    kw = kw or {}
    n_sub = int(min(n_sub, n_workload))
    X = make_data(n_sub, d, seed=seed)
    learner = OracleITAL(X, length_scale=length_scale or float(np.sqrt(d / 12.0)), **kw)
    learner.update({0: 1})
    np.random.seed(0)
    t0 = time.time()
    steps = fetch_unlabelled_sampled(learner, k, processes=cores, budget_s=budget_s, n_max=n_sub - 1)
    dt = time.time() - t0
    round_s = sum(s_["fork_s"] + (n_workload - s_["t"] + 1) * s_["per_cand_s"] for s_ in steps)
    scored = sum(n_workload - s_["t"] + 1 for s_ in steps)
    return {"value": scored / round_s, "unit": "candidates/s", "cores": cores, "host_cpu_count": os.cpu_count(), "kind": "port",
            "extrapolated": True, "extrapolated_round_s": round_s,
            "sample": "oracle in the reference's fork-pool scheme on the first %d rows of a %d x %d matrix of the workload's kind "
                      "(k=%d%s), %.1f s of CPU work: candidates scored per greedy step %s (fewer where a step's share of the %d s "
                      "budget ends); per-candidate cost does not depend on N (reference ital/ital.py:183-224), extrapolated to N = %d"
                      % (n_sub, n_workload, d, k, "".join(", %s=%s" % kv for kv in sorted(kw.items())), dt,
                         [s_["scored"] for s_ in steps], int(budget_s), n_workload),
            "seconds_per_candidate_by_step": [round(s_["per_cand_s"], 6) for s_ in steps]}


def cpu_baselines_other(cores, whole_c5=True):
    """The extrapolated CPU baselines of the other BASELINE.json configurations the default line times (before anything
    touches the GPU: the baselines fork)."""
    out = {}
    out["ital_general_user_k4"] = cpu_baseline_sampled(ROWS_PER_GPU, DIM, BATCH, cores, dict(label_prob=0.5, mistake_prob=0.25),
                                                       length_scale=LENGTH_SCALE, seed=0, budget_s=12.0)
    out["ital_k8_25000x512"] = cpu_baseline_sampled(25000, 512, 8, cores, budget_s=15.0)
    out["ital_k8_50000x2048"] = cpu_baseline_sampled(50000, 2048, 8, cores, budget_s=15.0)
    c5 = cpu_baseline_sampled(125000, 512, 16, cores, dict(monte_carlo_num_rel=1), seed=1, budget_s=20.0)
    out["ital_k16_mc1_125000x512"] = c5
    if whole_c5:
        # the same measured seconds per candidate, N = 1M (nothing new is timed)
        steps = c5["seconds_per_candidate_by_step"]
        n = 1_000_000
        round_s = sum((n - t) * v for t, v in enumerate(steps))
        out["ital_k16_mc1_1Mx512"] = dict(c5, value=sum(n - t for t in range(len(steps))) / round_s, extrapolated_round_s=round_s,
                                          sample=c5["sample"].replace("extrapolated to N = 125000", "extrapolated to N = 1000000 (same sample)"))
    return out


def scaling_workload(device, rank, world, group, rounds=3):
    """north_star's scaling curve: n = 1M synthetic, d = 512, k = 4 with full enumeration, rows split over the ranks."""
    import torch
    from ital_amd import ITAL, mvn_stream, sharding
    n, d, k = SCALE_ROWS, SCALE_DIM, SCALE_BATCH
    row0, row1 = sharding.row_range(n, world, rank)
    local = block_rows(row0, row1, d, seed=1)
    data = sharding.ShardedRows(local, n, row0)
    mvn_stream.GLOBAL.reset()
    torch.cuda.synchronize()
    t_fit0 = time.perf_counter()
    L = ITAL(data, length_scale=float(np.sqrt(d / 12.0)), device=device, rank=rank, world=world, group=group)
    torch.cuda.synchronize()
    t_fit1 = time.perf_counter()
    del local

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def label(i):
        # the simulated perfect user's answer for sample i: any fixed function of the sample will do (every rank must
        # give the same one without holding the sample's features); a multiplicative hash of the index
        return 1.0 if (i * 2654435761) % (1 << 32) < (1 << 31) else -1.0

    def one_round():
        ret = L.fetch_unlabelled(k)
        L.update({int(i): label(int(i)) for i in ret})
        return ret

    picks = []
    L.update({0: 1})
    torch.cuda.synchronize()
    t_fit2 = time.perf_counter()
    picks.append(one_round())                              # warm-up
    L.host_clock = L.gp.host_clock = dict(gap_s=0.0, gaps=0, enqueue_s=0.0, t_download=None)
    L.profile = []
    L.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(6 * k * rounds + 16)]
    for ev in L.event_pool:
        ev.record()
    scored = 0
    from ital_amd import _lib
    barrier()
    launches0 = _lib.lib().ital_launch_count()
    t0 = time.perf_counter()
    for _ in range(rounds):
        n_cand = n - len(L.relevant_ids) - len(L.irrelevant_ids)
        picks.append(one_round())
        scored += sum(n_cand - t for t in range(k))
    barrier()
    dt = time.perf_counter() - t0
    launches = (_lib.lib().ital_launch_count() - launches0) / rounds
    hc, L.host_clock, L.gp.host_clock = L.host_clock, None, None
    parts = split_rounds(L, label, k, 2, barrier)          # (after the timed rounds; the picks of these are not compared)
    # parity evidence of the run itself: every rank must have picked the same batches (digest compared across the ranks),
    # and the batches of the N = 1 run of this workload (committed: PICKS_N1_FILE) -- the picks do not depend on the sharding
    sha, words = picks_digest(picks)
    agree = ranks_agree(words, device, world)
    key = "%dx%d_k%d_rounds%d" % (n, d, k, rounds + 1)
    ref_path = os.path.join(ROOT, PICKS_N1_FILE)
    ref = None
    if os.path.exists(ref_path):
        with open(ref_path) as f:
            ref = json.load(f).get(key)
    if world == 1 and rank == 0 and os.environ.get("ITAL_BENCH_WRITE_PICKS"):
        with open(os.environ["ITAL_BENCH_WRITE_PICKS"], "w") as f:
            json.dump({key: {"picks_sha": sha, "picks": [[int(i) for i in r_] for r_ in picks],
                             "ms_per_round_n1": dt / rounds * 1e3,
                             "ms_per_round_n1_note": "fetch + update round of this workload on ONE MI355X, the run that wrote this file"}}, f)
    # the strong-scaling point north_star asks for, readable from ONE line: this run's round time against the committed N = 1
    # time of the same workload (same rounds, same picks)
    n1_ms = ref.get("ms_per_round_n1") if ref else None
    host_ms = gather_floats(hc["gap_s"] / max(hc["gaps"], 1) * 1e3 if hc["gaps"] else float("nan"), device, world)
    enq_ms = gather_floats(hc["enqueue_s"] / max(hc["gaps"] + 1, 1) * 1e3, device, world)
    # ... and what that host time is made of (per rank, per round; rounds - 1 gaps, rounds updates)
    split = {name: gather_floats(hc.get(key, 0.0) / max(cnt, 1) * 1e3, device, world)
             for name, key, cnt in (("update_append_ms", "append_s", rounds), ("means_allgather_ms", "means_s", rounds),
                                    ("fetch_prologue_ms", "prologue_s", hc["gaps"]))}
    transport = transport_report(L, group, device)
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if world > 1 and not any(p_[0] == "exchange" for p_ in L.profile):
        # the rounds above ran as single calls (the exchange is issued from C): one more round step by step, untimed, for
        # the duration of the per-step collective (every rank takes this branch: same path on all of them)
        L.round_call = False
        one_round()
        torch.cuda.synchronize()
    prof = {}
    for name, t, n_c, e0, e1 in L.profile:
        prof.setdefault((name, t), []).append(e0.elapsed_time(e1))
    ex = [v for key, vs in prof.items() if key[0] == "exchange" for v in vs]
    backend = None
    if group is not None:
        import torch.distributed as dist
        backend = dist.get_backend(group)
    mem = torch.cuda.max_memory_allocated(device) / 2 ** 30
    del L
    gc.collect()
    torch.cuda.empty_cache()
    return {"workload": "synthetic %d x %d, k=%d, perfect user, full 2^t enumeration, fetch_unlabelled + update per round"
                        % (n, d, k), "scaling": "strong", "rows_per_rank": row1 - row0, "world_size": world,
            "backend": backend, "rounds": rounds, "ms_per_round": dt / rounds * 1e3, "candidates_per_s": scored / dt,
            "n1_ms_per_round": n1_ms, "speedup_vs_n1": (n1_ms / (dt / rounds * 1e3)) if n1_ms else None,
            "efficiency": (n1_ms / (dt / rounds * 1e3) / world) if n1_ms else None,
            "speedup_note": "ms_per_round of the committed one-GPU run of this workload (%s) / this run's; efficiency = speed-up / "
                            "world_size; north_star's target: >= 6x at 8 GPUs" % PICKS_N1_FILE,
            "fit_ms": {"construct_ms": (t_fit1 - t_fit0) * 1e3, "first_update_ms": (t_fit2 - t_fit1) * 1e3,
                       "note": "construct = upload of this rank's rows (host -> device over PCIe: 4 GB at 1M x 512 on one rank) + "
                               "row norms + buffers; the reference's fit forms the dense N x N kernel matrix here (gp.py:128), "
                               "which at N = 1M does not exist on any machine"},
            **{k_: parts[k_] for k_ in ("fetch_ms_per_round", "update_ms_per_round")},
            "picks_sha": sha, "picks_agree_across_ranks": agree,
            "picks_match_n1": (sha == ref["picks_sha"]) if ref else None, "picks_n1_file": PICKS_N1_FILE if ref else None,
            "host_ms_per_round": host_ms, "host_enqueue_ms_per_round": enq_ms, "host_ms_split": split,
            "host_ms_split_note": "per rank, host-side time of one round's update(): the Cholesky append + whitening sweep calls "
                                  "(update_append_ms), the replication of the refreshed means (means_allgather_ms: asynchronous on "
                                  "RCCL; a gloo rehearsal stages 8 B per sample through the host and blocks), and of the next "
                                  "fetch's prologue up to the call that enqueues the round (fetch_prologue_ms); the rest of "
                                  "host_ms_per_round is the caller's feedback dictionary and the candidate bookkeeping of update()",
            "host_ms_note": "per rank: host time between the download of a round's picks and the call that enqueues the next "
                            "round (feedback, update(), fetch prologue) / time inside that call + the next round's descriptor "
                            "(GPU busy meanwhile); NaN: the rounds did not run as single calls",
            **transport,
            "exchange_ms_per_greedy_step": float(np.mean(ex)) if ex else None,
            "kernel_ms": {"%s_t%d" % key: float(np.mean(v)) for key, v in sorted(prof.items())},
            "kernel_ms_note": "qmc_main_t*: the lattice-sum kernel alone (one slab of the workspace); qmc_slabsN_t*: first to "
                              "last lattice sum of a step that needed N slabs, incl. the preparation / combine launches between",
            "library_launches_per_round": launches, "peak_device_memory_gib": mem,
            "note": "per-step exchange = ncclAllGather of one record per rank on the process group's communicator (null on one "
                    "rank: no collective)"}


def headline_workload_name(world):
    """config.workload of the line: at N > 1 the headline is the WEAK-scaled metric workload (9298 rows on every GPU: a step is
    launch-latency bound there and the line prices the collective); north_star's strong-scaling curve is scaling_workload."""
    base = ("USPS-shaped synthetic %dx%d, k=%d, perfect user, full 2^t enumeration, fetch_unlabelled + update per step"
            % (ROWS_PER_GPU, DIM, BATCH))
    if world == 1:
        return base
    return ("weak: %d rows PER GPU (%d in all) -- " % (ROWS_PER_GPU, ROWS_PER_GPU * world)) + base + \
        "; the STRONG-scaling point (1M x 512, k=4) is scaling_workload / config.strong_scaling_1M"


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): this process -- which has made no GPU call and never
    will -- starts the N ranks as FRESH child processes through torch.distributed.run (one per GPU, rendezvous on
    127.0.0.1 and a free port), passes everything they print on stdout/stderr through to its own stderr, prints rank 0's
    JSON line as ITS last stdout line and returns the launcher's exit code.  The counterpart of the reference spawning its
    own workers (ital/ital.py:124-126); never exec: see the pool's rule about replacing a process."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("bench.py: starting %d ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    line_json = None
    for line in proc.stdout:
        s_ = line.strip()
        if s_.startswith("{") and s_.endswith("}") and '"metric"' in s_:
            line_json = s_
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    sys.stderr.flush()
    if line_json is not None:
        print(line_json, flush=True)
    if rc == 0 and line_json is None:
        print("bench.py: the ranks exited 0 but printed no JSON line", file=sys.stderr)
        rc = 1
    return rc


def dry_run(rank, world, args):
    """The launch path without a GPU: rendezvous (gloo), the digest comparison of the ranks, one JSON line from rank 0."""
    import torch
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if os.environ.get("ITAL_BENCH_DRY_FAIL_RANK") == str(rank):
        raise SystemExit("bench.py --dry-run: rank %d fails on request" % rank)      # the launch test's failing rank
    sha, words = picks_digest([[1, 2, 3, 4]])
    agree = ranks_agree(words, torch.device("cpu"), world)
    ranks = gather_floats(rank, torch.device("cpu"), world)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # (the keys a reader of an N > 1 line looks for are present in the dry run too: tests/test_bench_launch.py)
        print(json.dumps({"metric": "MI-scored candidates/sec per fetch_unlabelled(k) round", "value": None, "dry_run": True,
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ranks_seen": ranks,
                          "picks_agree_across_ranks": agree, "scaling": "weak",
                          "config": {"workload": headline_workload_name(world)},
                          "scaling_workload": {"scaling": "strong", "world_size": world, "ms_per_round": None, "n1_ms_per_round": None,
                                               "speedup_vs_n1": None, "efficiency": None}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scaling-workload", action="store_true", help="skip the 1M x 512 strong-scaling record")
    ap.add_argument("--rows", type=int, default=9298, help="rows per GPU (experiments; the default is the metric's workload)")
    ap.add_argument("--batch", type=int, default=4, help="batch size k (experiments)")
    ap.add_argument("--label-prob", type=float, default=1.0, help="user model (experiments; != 1 selects the general scorer)")
    ap.add_argument("--mistake-prob", type=float, default=0.0)
    ap.add_argument("--extra", default="", help="comma list of opt-in workloads (N = 1): topcand = 1 000 000 x 512 with "
                    "top_candidates = 4096 (the host's argpartition share).  (c4 / c5k16 -- BASELINE configs[3] and [4] -- are part "
                    "of the default run since round 6.)")
    ap.add_argument("--quick", action="store_true",
                    help="headline + scaling workload only: skips the other BASELINE configurations (configs[2], [3], [4] share "
                         "and whole, the noisy user, MCMI, the 20 000^2 covariance block) and their CPU baselines, which the "
                         "default run times in ~4 minutes (ITAL_BENCH_NO_EXTRAS=1 does the same)")
    ap.add_argument("--no-c5-whole", action="store_true",
                    help="skip BASELINE configs[4] in one piece (1 000 000 x 512, k = 16: ~90 s of GPU); its 125 000-row share "
                         "(what one of 8 ranks runs) stays")
    ap.add_argument("--force-collectives", action="store_true",
                    help="one rank, but through the exchange path of N > 1 (1-rank RCCL group): prices the per-step collective")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch path only (no GPU): the ranks rendezvous over gloo, agree on a digest and rank 0 prints a "
                         "line with dry_run: true -- what tests/test_bench_launch.py runs on CPU")
    args = ap.parse_args()
    globals().update(ROWS_PER_GPU=args.rows, BATCH=args.batch)
    t_process = time.perf_counter()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    if args.dry_run:
        return dry_run(rank, world, args)
    cpu_base = None
    cpu_other = {}
    extras_on = world == 1 and not args.quick and not os.environ.get("ITAL_BENCH_NO_EXTRAS")
    if world == 1 and not args.no_cpu_baseline:
        # before anything touches the GPU: the baselines fork worker pools
        cores = cpu_cores()
        cpu_base = cpu_baseline(make_data(ROWS_PER_GPU, DIM, seed=0), cores)
        if extras_on:
            cpu_other = cpu_baselines_other(cores, whole_c5=not args.no_c5_whole)
    import torch
    if os.environ.get("ITAL_BENCH_ONE_DEVICE"):
        local_rank = 0      # debugging aid for a 1-GPU box: all ranks on cuda:0 (use with ITAL_BENCH_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    group = None
    if world > 1 or args.force_collectives:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.force_collectives:
            os.environ["ITAL_FORCE_COLLECTIVES"] = "1"
        backend = os.environ.get("ITAL_BENCH_BACKEND", "nccl")      # "nccl" is RCCL; gloo only to rehearse N > 1 on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        group = dist.group.WORLD

    from ital_amd import ITAL, mvn_stream

    n_total = ROWS_PER_GPU * world
    X = make_data(n_total, DIM, seed=0)
    rel = np.where(X[:, 0] > 0.5, 1.0, -1.0)
    torch.cuda.synchronize()
    t_fit0 = time.perf_counter()
    learner = ITAL(X, length_scale=LENGTH_SCALE, label_prob=args.label_prob, mistake_prob=args.mistake_prob, device=device,
                   rank=rank, world=world, group=group)
    torch.cuda.synchronize()
    t_fit1 = time.perf_counter()
    learner.update({0: 1})
    torch.cuda.synchronize()
    fit_ms = {"construct_ms": (t_fit1 - t_fit0) * 1e3, "first_update_ms": (time.perf_counter() - t_fit1) * 1e3,
              "note": "first use in the process (library load, lazily initialised torch kernels, allocations): construct = upload of "
                      "the rows + row norms + buffers, first update = Cholesky append + whitening sweep + means"}
    learner.reset()

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

    # a serving process freezes its start-up heap (ital_amd.serving_mode()): without this CPython's generation-2 collector
    # walks torch's ~10^5 objects once every few rounds (a 40 ms pause, measured: ten 3.4 ms rounds cost 73 ms), which has
    # nothing to do with the path under test; the figure without it is measured after the timed region
    # (ms_per_step_unfrozen_heap).  Done BEFORE the warm-up: the collection itself takes ~0.1 s during which the GPU idles
    # and drops its clocks -- between warm-up and timed loop it cost the first eight timed rounds 0.1 - 0.9 ms each
    # (3.6, 3.2, 3.1, 3.0 ... 2.7 ms)
    import ital_amd
    ital_amd.serving_mode()
    # timing events are created before the warm-up too (creating one costs ~0.3 ms on a loaded host: GPU idle time again) and
    # only recorded inside the timed region: one pair per round, around the dominant kernel -- not more than needed (round 2's pool
    # of 1280 events took 0.4 s to create)
    use_events = not os.environ.get("ITAL_BENCH_NO_EVENTS")
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(2 * (args.steps + 2) if use_events else 0)]
    for ev in pool:
        ev.record()          # creates the handle (the library records them itself, around single kernels)
    # the GPU has idled through the CPU baseline, the imports and the collection above and sits at low clocks: rounds of the
    # workload for ~0.1 s bring them up (the ramp was measured to take ~8 rounds = 25 ms), then the W warm-up rounds
    restart()
    preheat = 36              # a fixed count (~0.1 s): with several ranks every round is collective, all must run the same number
    for _ in range(preheat):
        one_round()
    restart()
    for _ in range(args.warmup):
        one_round()
    restart()
    learner.profile = [] if use_events else None
    learner.profile_steps = {BATCH}      # events around the dominant kernel only (every record is a barrier packet in the queue)
    learner.event_pool = pool
    from ital_amd import _lib
    rows_local = learner.gp.n
    scored = 0
    barrier()
    launches0 = _lib.lib().ital_launch_count()
    t0 = time.perf_counter()
    marks = []
    timed_picks = []
    for _ in range(args.steps):
        n_cand = n_total - len(learner.relevant_ids) - len(learner.irrelevant_ids)
        timed_picks.append(one_round())
        scored += sum(n_cand - t for t in range(BATCH))
        marks.append(time.perf_counter())
    barrier()
    dt = time.perf_counter() - t0
    launches_end = _lib.lib().ital_launch_count()
    head_sha, head_words = picks_digest(timed_picks)
    head_agree = ranks_agree(head_words, device, world)
    head_transport = transport_report(learner, group, device)
    # the same K steps with the interpreter's heap as a process that never froze it has it (reported next to the headline)
    prof_timed, learner.profile = learner.profile, None
    gc.unfreeze()
    restart()
    barrier()
    tu = time.perf_counter()
    marks_u = []
    for _ in range(args.steps):
        one_round()
        marks_u.append(time.perf_counter())
    barrier()
    dt_unfrozen = time.perf_counter() - tu
    parts = split_rounds(learner, lambda i: float(rel[i]), BATCH, min(args.steps, 5), barrier)
    learner.profile = prof_timed
    if os.environ.get("ITAL_BENCH_STEP_TIMES"):
        print("step ms (timed):    " + " ".join("%.2f" % ((b_ - a_) * 1e3) for a_, b_ in zip([t0] + marks[:-1], marks)), file=sys.stderr)
        print("step ms (unfrozen): " + " ".join("%.2f" % ((b_ - a_) * 1e3) for a_, b_ in zip([tu] + marks_u[:-1], marks_u)), file=sys.stderr)
    launches = (launches_end - launches0) / args.steps
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt, dt_unfrozen], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_unfrozen = float(tt[0].item()), float(tt[1].item())

    # per-kernel durations from the HIP events recorded on the launch stream during the timed region (the lattice sums: the
    # library brackets them itself); the short kernels of the first greedy steps are sampled in three more rounds, untimed,
    # launched step by step from Python (inside the timed region the whole round is one C call without events around them)
    prof = {}
    for name, t, n_c, e0, e1 in (learner.profile or []):
        prof.setdefault((name, t), []).append((e0.elapsed_time(e1) * 1e-3, n_c))
    if learner.profile is not None:
        learner.profile = []
        learner.profile_steps = None
        learner.round_call = False
        learner.event_pool = [torch.cuda.Event(enable_timing=True) for _ in range(16 * BATCH * 3)]
        for ev in learner.event_pool:
            ev.record()
        for _ in range(3):
            one_round()
        torch.cuda.synchronize()
        for name, t, n_c, e0, e1 in learner.profile:
            if not (name == "qmc_main" and t == BATCH):
                prof.setdefault((name, t), []).append((e0.elapsed_time(e1) * 1e-3, n_c))
        learner.round_call = True
    learner.profile = None
    # SURVEY.md 8d "Seeds 0, 1, 2": the same K timed steps on the matrices of seeds 1 and 2 (seed 0 is `value`); same protocol
    # -- frozen heap, pre-heat, W warm-up rounds, barrier + synchronize on both sides, maximum over ranks
    seed_ms = {0: dt / args.steps * 1e3}
    for sd in (1, 2):
        Xs = make_data(n_total, DIM, seed=sd)
        rels = np.where(Xs[:, 0] > 0.5, 1.0, -1.0)
        Ls = ITAL(Xs, length_scale=LENGTH_SCALE, label_prob=args.label_prob, mistake_prob=args.mistake_prob, device=device,
                  rank=rank, world=world, group=group)

        def round_s():
            ret_ = Ls.fetch_unlabelled(BATCH)
            Ls.update({int(i): float(rels[i]) for i in ret_})

        def restart_s():
            Ls.reset()
            mvn_stream.GLOBAL.reset()
            Ls.update({sd % n_total: 1})          # first label q = seed mod n
        ital_amd.serving_mode()
        restart_s()
        for _ in range(preheat):
            round_s()
        restart_s()
        for _ in range(args.warmup):
            round_s()
        restart_s()
        barrier()
        ts = time.perf_counter()
        for _ in range(args.steps):
            round_s()
        barrier()
        dts = time.perf_counter() - ts
        if world > 1:
            import torch.distributed as dist
            tt = torch.tensor([dts], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dts = float(tt.item())
        seed_ms[sd] = dts / args.steps * 1e3
        gc.unfreeze()
        del Ls, Xs
    gc.collect()
    scale = None
    if not args.no_scaling_workload:
        del learner
        gc.collect()
        torch.cuda.empty_cache()
        scale = scaling_workload(device, rank, world, group)
    out = None
    if rank == 0:
        qm = prof.get(("qmc_main", BATCH), [])
        roof = None
        if qm:
            avg_s = float(np.mean([d for d, _ in qm]))
            avg_c = float(np.mean([c for _, c in qm]))
            flops = qmc_pairs(BATCH, avg_c) * FLOP_PER_PAIR
            ach = flops / avg_s / 1e12
            kname = "void ital::qmc_main_kernel<%d>" % BATCH
            roof = dict({"bound": "fp64-valu", "kernel": "qmc_main_kernel<%d>" % BATCH, "achieved": ach,
                         "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_VALU_PEAK_TFLOPS,
                         "avg_launch_ms": avg_s * 1e3, "pairs_per_s": qmc_pairs(BATCH, avg_c) / avg_s,
                         "flop_per_pair": FLOP_PER_PAIR, "valu_per_pair_isolated_chain": VALU_PER_PAIR_CHAIN,
                         "algorithmic_pairs_per_launch": qmc_pairs(BATCH, avg_c),
                         "note": "transcendental FP64 chains (Phi, Phi^-1): neither HBM nor MFMA bounds this kernel "
                                 "(SURVEY.md 8d S-qmc), so the peak is the FP64 vector rate; achieved = algorithmic pairs x "
                                 "flops of the isolated chain / launch time (HIP events the library records around this "
                                 "kernel alone).  Only ~55 % of the chain's instructions are FMAs, so the flop fraction "
                                 "understates how busy the vector unit is: valu_issue_frac is the share of its issue slots "
                                 "the kernel fills.  HBM-bound streaming kernel in roofline_hbm"},
                        **pmc_fields("headline", kname, avg_s))
        cc = prof.get(("cross_cov", 1), [])
        roof_hbm = None
        if cc:
            avg_s = float(np.mean([d for d, _ in cc]))
            m_avg = float(np.mean([c for _, c in cc]))
            bytes_alg = rows_local * 8.0 * (DIM + m_avg + 1)
            ach = bytes_alg / avg_s / 1e9
            roof_hbm = {"bound": "hbm", "kernel": "kcols_kernel (cross-covariance column)", "achieved": ach,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_ms": avg_s * 1e3,
                        "note": "19 MB per launch at this size: launch-latency bound, see DESIGN.md for the large-N figure"}
        sv = sorted(seed_ms.values())
        seeds_obj = {"ms_per_step_by_seed": {str(k_): round(v_, 4) for k_, v_ in sorted(seed_ms.items())}, "min": sv[0],
                     "median": sv[len(sv) // 2], "max": sv[-1],
                     "note": "`value` / `ms_per_step` are seed 0; seeds 1, 2: the same protocol on make_data(seed), first label seed mod n"}
        strong = None
        if scale is not None:
            strong = {k_: scale[k_] for k_ in ("ms_per_round", "candidates_per_s", "n1_ms_per_round", "speedup_vs_n1", "efficiency",
                                               "world_size", "picks_match_n1")}
        out = {"metric": "MI-scored candidates/sec per fetch_unlabelled(k) round", "value": scored / dt,
               "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": headline_workload_name(world),
                          "n": n_total, "d": DIM, "k": BATCH, "length_scale": LENGTH_SCALE,
                          "parallelism": "candidate rows sharded over %d GPU(s), 1 record all-gather per greedy step" % world,
                          "seeds": seeds_obj, "strong_scaling_1M": strong},
               "roofline_hbm_at_workload_size": roof_hbm,
               "kernel_ms": {"%s_t%d" % k: float(np.mean([d for d, _ in v])) * 1e3 for k, v in sorted(prof.items())},
               "fit_ms": fit_ms, "fetch_ms_per_round": parts["fetch_ms_per_round"], "update_ms_per_round": parts["update_ms_per_round"],
               "fetch_update_split_note": parts["note"],
               "ms_per_step_unfrozen_heap": dt_unfrozen / args.steps * 1e3, "preheat_rounds_before_warmup": preheat,
               "library_launches_per_step": launches, "round_gaps": round_gaps(),
               "picks_sha": head_sha, "picks_agree_across_ranks": head_agree, **head_transport,
               "scaling_workload": scale}
        if scale is not None:
            # the strong-scaling figure north_star asks for (1M x 512, k = 4, rows split over the ranks), also at top level
            out["value_strong_1M"] = scale["candidates_per_s"]
            out["ms_per_round_strong_1M"] = scale["ms_per_round"]
        ow = {}
        if extras_on:
            ow = other_workloads(X, rel, device)
            ow["ital_k8_50000x2048"] = dict(k8_workload(device, 50000, 2048, 8), config="synthetic 50000 x 2048, k=8, perfect user, "
                                            "full 2^t enumeration (BASELINE.json configs[3], SURVEY 8d C4'), one GPU")
            ow["ital_k16_mc1_125000x512"] = k16_workload(device, n=125_000)
            if not args.no_c5_whole:
                ow["ital_k16_mc1_1Mx512"] = k16_workload(device)
            ow["cov_block_20000x512"] = cov_block_workload(device)
        extra = [e for e in args.extra.split(",") if e]
        if world == 1 and "topcand" in extra:
            ow["ital_topcand4096_1Mx512"] = topcand_workload(device)
        for name, cb in cpu_other.items():      # each configuration's own CPU baseline (extrapolated from a bounded sample)
            if name in ow:
                ow[name]["cpu_baseline"] = cb
                ow[name]["speedup_vs_cpu_baseline"] = ow[name]["candidates_per_s"] / cb["value"]
        if ow:
            out["other_workloads"] = ow
        # ---- what a reader of the line's tail / of its `roofline` and `cpu_baseline` objects needs: every configuration's
        # dominant kernel against its bound, and every configuration's CPU baseline, compact
        hbm = hbm_stream_probe(device) if world == 1 else roof_hbm
        out["roofline_hbm"] = hbm

        def brief(r_):
            return None if not r_ else {k_: r_.get(k_) for k_ in ("bound", "achieved", "peak", "unit", "frac", "avg_launch_ms") if k_ in r_}
        others = {}
        hbm_name = "kcols 1Mx256 (hbm)" if world == 1 else "kcols at the workload's %d rows per rank (hbm; launch-latency bound)" % rows_local
        if hbm:
            others["kcols_kernel (streaming GP kernel), " + hbm_name] = dict(brief(hbm), traffic=hbm.get("traffic"))
        for name, w_ in ow.items():
            r_ = w_.get("roofline")
            if not r_:
                continue
            if "bound" in r_:
                others[name] = brief(r_)
            else:                                 # a dict of kernels (k = 8 workloads: t = 7, 8; MCMI: block + scorer)
                for kn, rr in r_.items():
                    others["%s/%s" % (name, kn)] = brief(rr)
        if roof is not None:
            roof["others"] = others
            roof["others_note"] = ("dominant kernel of every other workload of this run (other_workloads.<name>.roofline has the "
                                   "details): fp64-valu = algorithmic (Phi, Phi^-1) pairs x flops of the isolated chain / time")
        out["roofline"] = roof
        if cpu_base:
            cpu_base = dict(cpu_base)
            cpu_base["others"] = {name: {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                                         "extrapolated": True, "gpu_value": ow.get(name, {}).get("candidates_per_s"),
                                         "speedup": (ow[name]["candidates_per_s"] / cb["value"]) if name in ow else None,
                                         "sample": cb["sample"][:160]}
                                  for name, cb in cpu_other.items()}
        out["cpu_baseline"] = cpu_base
        if cpu_base:
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu_base["value"]
        # last key = the tail of the line: one row per BASELINE configuration
        summ = {"C2' 9298x256 k4 (value)": [round(dt / args.steps * 1e3, 3), round(scored / dt), roof and round(roof["frac"], 3),
                                             cpu_base and round(cpu_base["value"])]}
        for label_, name in (("C3' 25000x512 k8", "ital_k8_25000x512"), ("C4' 50000x2048 k8", "ital_k8_50000x2048"),
                             ("C5' share 125000x512 k16 mc1", "ital_k16_mc1_125000x512"), ("C5' 1Mx512 k16 mc1", "ital_k16_mc1_1Mx512"),
                             ("noisy user 9298x256 k4", "ital_general_user_k4")):
            w_ = ow.get(name)
            if not w_:
                continue
            r_ = w_.get("roofline") or {}
            fr = r_.get("frac") if "bound" in r_ else max([rr.get("frac", 0.0) for rr in r_.values()] or [None])
            cb = w_.get("cpu_baseline")
            summ[label_] = [round(w_["ms_per_round"], 2), round(w_["candidates_per_s"]), fr and round(fr, 3), cb and round(cb["value"], 1)]
        if "cov_block_20000x512" in ow:
            summ["cov_block 20000^2x512 (mfma)"] = [round(ow["cov_block_20000x512"]["ms_per_launch"], 2), None,
                                                    round(ow["cov_block_20000x512"]["roofline"]["frac"], 3), None]
        if hbm:
            summ[hbm_name] = [round(hbm["avg_launch_ms"], 3), None, round(hbm["frac"], 3), None]
        if strong:
            summ["strong 1Mx512 k4"] = [round(strong["ms_per_round"], 2), round(strong["candidates_per_s"]), strong["speedup_vs_n1"] and
                                        round(strong["speedup_vs_n1"], 2), None]
        out["summary"] = {"columns": ["ms_per_round", "gpu_candidates_per_s", "roofline_frac (strong: speedup_vs_n1)",
                                      "cpu_baseline_candidates_per_s (extrapolated beyond C2')"], "rows": summ,
                          "ms_per_step_seeds_0_1_2": [round(seed_ms[q], 3) for q in sorted(seed_ms)], "cpu_cores": cpu_base and cpu_base["cores"],
                          "bench_wall_s": round(time.perf_counter() - t_process, 1)}
    # the JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which a redirected stdout holds
    # back until the process ends -- every rank pushes its own out before the final barrier, rank 0 prints after it
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass
    sys.stdout.flush()
    if group is not None:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)
    if not head_agree or (scale is not None and not scale["picks_agree_across_ranks"]):
        raise SystemExit("bench.py: the ranks did NOT pick the same batches -- the line above is not a valid measurement")


if __name__ == "__main__":
    main()
