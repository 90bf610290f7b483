#!/bin/bash
# A/B of lattice-sum variants for 11 .. 15 variables (three chains at FOUR waves per SIMD instead of four at three): one
# fetch_unlabelled(16) with monte_carlo_num_rel = 1 on 40 000 x 512 per library variant of build_variants/, twice, same box.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_m
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  for v in ${VARIANTS:-base v1 v2 v3}; do
    ITAL_HIP_LIB=$ROOT/build_variants/libital_$v.so timeout 300 python3 tools/scale_probe.py 40000 512 16 1 2>&1 | grep -v "^[EW]20\|amdgpu.ids" | tail -n 3 | head -n 2 > $OUT/${v}_rep$rep.log
  done
done
python3 - <<'PY'
import ast, glob, os, re
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r6_m")
rows = {}
for f in sorted(glob.glob(out + "/*_rep*.log")):
    txt = open(f).read()
    m = re.search(r"(\{'cross_cov.*\})", txt)
    sec = re.search(r": ([0-9.]+) s ->", txt)
    picks = re.search(r"picks (\[.*\])", txt)
    d = ast.literal_eval(m.group(1)) if m else {}
    rows[os.path.basename(f)] = (float(sec.group(1)) if sec else None, [d.get("score_generic_t%d" % t) for t in range(9, 17)], hash(picks.group(1)) if picks else None)
print("variant            fetch_s   ms per step t = 9 .. 16                                   picks")
for k, (s, steps, p) in rows.items():
    print("%-18s %7s   %s   %s" % (k, s, " ".join("%7.1f" % v if v else "   n/a " for v in steps), p))
PY
