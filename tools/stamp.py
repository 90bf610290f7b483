#!/usr/bin/env python3
"""Which kernels a committed profile was taken with: sha256 over the kernel sources (ital_amd/csrc/*, include/ital_hip.h --
stable across rebuilds of the same code, unlike the bytes of a rebuilt .so) plus, for the record, the commit and the sha256
of the library binary that ran.  bench.py quotes counters out of committed files only while `csrc_sha` still matches.

    python tools/stamp.py profiles/r4_stamp.json profiles/r4_headline_pmc_summary.csv profiles/r4_round_gaps.json ...

On the GPU box (no .git there) the commit comes from ITAL_COMMIT; files are keyed as profiles/<basename> whatever directory
they are written to first (tools/profile_r4.sh writes under gpurun_out/, the summaries are then copied to profiles/).
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ital_amd", "csrc")
    for name in sorted(os.listdir(d)) + [os.path.join("..", "..", "include", "ital_hip.h")]:
        path = os.path.join(d, name)
        if os.path.isfile(path):
            h.update(os.path.basename(name).encode() + b"\0")
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


def main():
    out, files = sys.argv[1], sys.argv[2:]
    stamp = {}
    if os.path.exists(out):
        with open(out) as f:
            stamp = json.load(f)
    entry = {"csrc_sha": csrc_sha(), "lib_sha256": lib_sha(), "commit": commit() or os.environ.get("ITAL_COMMIT")}
    for name in files:
        stamp["profiles/" + os.path.basename(name)] = entry
    with open(out, "w") as f:
        json.dump(stamp, f, indent=1, sort_keys=True)
    print(entry)


if __name__ == "__main__":
    main()
