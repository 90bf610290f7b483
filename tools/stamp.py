#!/usr/bin/env python3
"""Which kernels a committed profile was taken with: sha256 over the kernel sources (ital_amd/csrc/*, include/ital_hip.h --
stable across rebuilds of the same code, unlike the bytes of a rebuilt .so; since round 5 over the include closure of the
translation units the profiled workload runs, so that a change to another scorer does not void a profile) plus, for the record, the commit and the sha256
of the library binary that ran.  bench.py quotes counters out of committed files only while `csrc_sha` still matches.

    python tools/stamp.py profiles/r4_stamp.json profiles/r4_headline_pmc_summary.csv profiles/r4_round_gaps.json ...
    python tools/stamp.py --check profiles/r5_stamp.json        (which stamped profiles no longer match the tree)

On the GPU box (no .git there) the commit comes from ITAL_COMMIT; files are keyed as profiles/<basename> whatever directory
they are written to first (tools/profile_r4.sh writes under gpurun_out/, the summaries are then copied to profiles/).
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


CSRC = os.path.join(ROOT, "ital_amd", "csrc")
HEADER = os.path.join(ROOT, "include", "ital_hip.h")
# translation units whose kernels a profile's numbers come from (by the workload in the file name, profiles/r5_<workload>_*);
# anything else: every source of the library
UNITS = {"headline": ["score.hip", "round.hip", "rbf.hip", "chol.hip", "select.hip"], "k8": ["score.hip", "round.hip", "rbf.hip"],
         "general": ["gen_pipeline.hip", "score_generic.hip"], "c5": ["gen_pipeline.hip", "score_generic.hip"],
         "mcmi": ["mcmi.hip"], "mcmiall": ["mcmi.hip"], "kcols": ["rbf.hip"], "cesub": ["gen_pipeline.hip", "score_generic.hip"], "round": ["score.hip", "round.hip", "rbf.hip", "chol.hip", "select.hip"]}


def closure(units):
    """The units and every file of ital_amd/csrc / include they include, transitively (sorted base names -> paths)."""
    import re
    todo, seen = [os.path.join(CSRC, u) for u in units], {}
    while todo:
        path = todo.pop()
        base = os.path.basename(path)
        if base in seen or not os.path.isfile(path):
            continue
        seen[base] = path
        with open(path) as f:
            for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', f.read(), flags=re.M):
                todo.append(HEADER if inc == "ital_hip.h" else os.path.join(CSRC, inc))
    return dict(sorted(seen.items()))


def units_of(profile_name):
    """Translation units behind a committed profile, from the workload in its file name (r5_<workload>_...)."""
    parts = os.path.basename(profile_name).split("_")
    return UNITS.get(parts[1]) if len(parts) > 1 else None


def csrc_sha(units=None):
    """sha256 (16 hex digits) over the kernel sources: the include closure of `units`, or all of ital_amd/csrc + the header."""
    h = hashlib.sha256()
    if units:
        files = closure(units)
    else:
        files = {name: os.path.join(CSRC, name) for name in sorted(os.listdir(CSRC)) if os.path.isfile(os.path.join(CSRC, name))}
        files["ital_hip.h"] = HEADER
    for name, path in files.items():
        h.update(name.encode() + b"\0")
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def lib_sha():
    path = os.path.join(ROOT, "ital_amd", "libital_hip.so")
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        return None


def check(path):
    """`stamp.py --check profiles/r6_stamp.json`: which of the stamped profiles were taken with other kernel sources than
    the tree's (exit code 1 if any)."""
    with open(path) as f:
        stamp = json.load(f)
    stale = [name for name, e in sorted(stamp.items()) if e.get("csrc_sha") != csrc_sha(e.get("units"))]
    for name in stale:
        print("stale:", name, "(units %s)" % (stamp[name].get("units"),))
    print("%d of %d stamped profiles match the tree's kernel sources" % (len(stamp) - len(stale), len(stamp)))
    return 1 if stale else 0


def main():
    if len(sys.argv) >= 3 and sys.argv[1] == "--check":
        sys.exit(check(sys.argv[2]))
    if len(sys.argv) < 2 or any(a.startswith("-") for a in sys.argv[1:]):
        # (round 5: `stamp.py --check` wrote a file named "--check" into the repo root)
        sys.exit("usage: stamp.py OUT.json PROFILE...  |  stamp.py --check STAMP.json")
    out, files = sys.argv[1], sys.argv[2:]
    stamp = {}
    if os.path.exists(out):
        with open(out) as f:
            stamp = json.load(f)
    entry = None
    for name in files:
        units = units_of(name)
        entry = {"csrc_sha": csrc_sha(units), "units": units, "lib_sha256": lib_sha(), "commit": commit() or os.environ.get("ITAL_COMMIT")}
        stamp["profiles/" + os.path.basename(name)] = entry
    with open(out, "w") as f:
        json.dump(stamp, f, indent=1, sort_keys=True)
    print(entry)


if __name__ == "__main__":
    main()
