#!/bin/bash
# Round 6, GPU run F: the round's profiles once more with the final sources (a comment in include/ital_hip.h had voided the
# stamps), then the default bench line quoting them.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_f
mkdir -p $OUT
cd $ROOT
bash tools/profile_r6.sh headline k8 general c5 mcmi kcols > $OUT/profile.log 2>&1
cp $ROOT/gpurun_out/prof_r6/r6_* $ROOT/profiles/ 2>/dev/null
python3 tools/c5_fractions.py profiles/r6_c5_kernel_stats.csv profiles/r6_c5_pmc_summary.csv > $ROOT/gpurun_out/prof_r6/r6_c5_kernel_fractions.txt
python3 tools/stamp.py --check profiles/r6_stamp.json > $OUT/stamp_check.txt 2>&1
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
cat $OUT/stamp_check.txt
tail -c 1600 $OUT/bench_default.json
