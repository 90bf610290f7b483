#!/usr/bin/env python3
"""Static instruction mix of one kernel's gfx950 ISA (whole kernel text: the lattice loops dominate it):
    python tools/isa_mix.py score.hip 'qmc_main_kernelILi8E' [-D...]
Counts by class -- vector ALU, of which FP64 arithmetic, lane moves of spilled scalars (v_readlane / v_writelane),
scalar ALU, LDS, scratch -- to compare variants before spending GPU time on them."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, pat = sys.argv[1], sys.argv[2]
    extra = sys.argv[3:]
    path = src if os.path.exists(src) else os.path.join(ROOT, "ital_amd", "csrc", src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I",
           os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ital_amd", "csrc"), "-S", "--cuda-device-only",
           "-o", "-", path] + extra
    asm = subprocess.run(cmd, capture_output=True, text=True).stdout
    inside, name, counts = False, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m and pat in m.group(1):
            inside, name, counts = True, m.group(1), {}
            continue
        if not inside:
            continue
        t = line.strip()
        if t.startswith("s_endpgm"):
            inside = False
            v = sum(c for k, c in counts.items() if k.startswith("v_"))
            f64 = sum(c for k, c in counts.items() if re.match(r"v_(fma|fmac|mul|add|rcp|rsq|ldexp|frexp\w*|floor|rndne|fract|cvt)_\w*f64", k))
            lane = counts.get("v_readlane_b32", 0) + counts.get("v_writelane_b32", 0)
            s = sum(c for k, c in counts.items() if k.startswith("s_"))
            ds = sum(c for k, c in counts.items() if k.startswith("ds_"))
            sc = sum(c for k, c in counts.items() if k.startswith("scratch_"))
            mov = counts.get("v_mov_b32_e32", 0) + counts.get("v_mov_b64_e32", 0) + counts.get("v_accvgpr_write_b32", 0) + counts.get("v_accvgpr_read_b32", 0)
            cnd = counts.get("v_cndmask_b32_e32", 0) + counts.get("v_cndmask_b32_e64", 0)
            print("%s\n  valu %d (fp64 arith %d, lane moves %d, moves %d, selects %d)  salu %d  lds %d  scratch %d" %
                  (subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:90], v, f64, lane, mov, cnd, s, ds, sc))
            continue
        op = t.split()[0] if t and not t.startswith((";", ".", "/")) and not t.endswith(":") else None
        if op:
            counts[op] = counts.get(op, 0) + 1


if __name__ == "__main__":
    main()
