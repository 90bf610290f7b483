"""Builds libital_hip.so (gfx950) in-tree with hipcc.  `python -m ital_amd.build [--force]`."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
LIB = os.path.join(HERE, "libital_hip.so")
SOURCES = ["api.hip", "rbf.hip", "chol.hip", "score.hip", "select.hip", "mcmi.hip", "score_generic.hip", "gen_pipeline.hip", "topk.hip", "exchange.hip", "round.hip", "ctx.hip",
           "mvn_stream.cpp", "np_legacy.cpp"]    # .cpp: host-only translation units (no HIP), also built by tools/asan_host.sh
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
EXTRA = os.environ.get("ITAL_HIPCC_EXTRA", "").split()          # added to every .hip compile (kernel-variant experiments)
HOST_EXTRA = os.environ.get("ITAL_HOST_EXTRA", "").split()      # added to the host-only .cpp compiles
LINK_EXTRA = os.environ.get("ITAL_LINK_EXTRA", "").split()      # added to the link (e.g. -fsanitize=..., tools/asan_host.sh)
LIB = os.environ.get("ITAL_HIP_LIB_OUT", LIB)
FLAGS = EXTRA + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
         "-I", INCLUDE, "-I", CSRC, "-Wall", "-Wno-unused-function"]
# host-only units: no fused multiply-adds (np_legacy.cpp reproduces numpy's doubles bit for bit)
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I", INCLUDE, "-I", CSRC, "-Wall"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
              [os.path.join(INCLUDE, "ital_hip.h")]
    objdir = os.environ.get("ITAL_OBJ_DIR", os.path.join(HERE, "_obj"))
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append([HIPCC] + (FLAGS if src.endswith(".hip") else HOST_FLAGS + HOST_EXTRA) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + LINK_EXTRA + objs + ["-ldl", "-lpthread"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
