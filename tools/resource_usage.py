#!/usr/bin/env python3
"""Register / scratch / occupancy table of the kernels of one translation unit, from the compiler's own report
(`-Rpass-analysis=kernel-resource-usage`, gfx950): python tools/resource_usage.py score.hip [name-filter] [-D...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    flt = [a for a in sys.argv[2:] if not a.startswith("-")]
    extra = [a for a in sys.argv[2:] if a.startswith("-")]
    path = src if os.path.exists(src) else os.path.join(ROOT, "ital_amd", "csrc", src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-I",
           os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ital_amd", "csrc"),
           "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(Function Name|[A-Za-z ]+): (\S+)", line) or \
            re.search(r":\s+(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        key, val = m.group(1), m.group(2)
        if key in ("Function Name", "Name"):
            cur = {"name": subprocess.run(["c++filt", val], capture_output=True, text=True).stdout.strip()}
            rows.append(cur)
        elif cur is not None:
            cur[key.split(" [")[0]] = val
    print("%-64s %5s %5s %5s %8s %4s %6s %6s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "sspill", "vspill", "LDS"))
    for r in rows:
        name = re.sub(r"\(.*", "", r["name"]).replace("void ital::", "")
        if flt and not any(f in name for f in flt):
            continue
        print("%-64s %5s %5s %5s %8s %4s %6s %6s %7s" % (name[:64], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"),
                                                          r.get("ScratchSize"), r.get("Occupancy"), r.get("SGPRs Spill"),
                                                          r.get("VGPRs Spill"), r.get("LDS Size")))


if __name__ == "__main__":
    main()
