#!/usr/bin/env python3
"""Vector-ALU instructions inside the loops of one kernel (static count over the gfx950 ISA), with the lane moves of
spilled scalars listed separately -- a quick check of a variant before GPU time is spent on it:
    python tools/loop_valu.py score.hip qmc_main_kernelILi4E [-D...]"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, pat = sys.argv[1], sys.argv[2]
    path = src if os.path.exists(src) else os.path.join(ROOT, "ital_amd", "csrc", src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I",
           os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ital_amd", "csrc"), "-S", "--cuda-device-only",
           "-o", "-", path] + sys.argv[3:]
    asm = subprocess.run(cmd, capture_output=True, text=True).stdout.splitlines()
    inside = inloop = False
    tot = Counter()
    for line in asm:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            inside, inloop = pat in m.group(1), False
            continue
        if not inside:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            inside = False
        if re.match(r"^\.LBB\d+_\d+:", t):
            inloop = "Loop" in t
            continue
        if inloop and t and not t.startswith((".", ";")):
            tot[t.split()[0]] += 1
    valu = sum(v for k, v in tot.items() if k.startswith("v_"))
    print("VALU in loops %d  (fma/fmac %d, cndmask %d, readlane/writelane %d, v_mov %d)  SALU %d  LDS %d  scratch %d" % (
        valu, sum(v for k, v in tot.items() if k.startswith(("v_fma", "v_fmac"))),
        sum(v for k, v in tot.items() if k.startswith("v_cndmask")),
        tot["v_readlane_b32"] + tot["v_writelane_b32"], sum(v for k, v in tot.items() if k.startswith("v_mov")),
        sum(v for k, v in tot.items() if k.startswith("s_")), sum(v for k, v in tot.items() if k.startswith("ds_")),
        sum(v for k, v in tot.items() if k.startswith("scratch_"))))


if __name__ == "__main__":
    main()
