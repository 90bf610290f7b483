#!/bin/bash
# Round-3 profiles on the GPU box (run through gpurun from the repo root): per workload one kernel-trace + stats run, then
# the PMC counters in runs of their own (FETCH_SIZE, WRITE_SIZE, SQ_*: never combined with sys/hip/hsa tracing).
#   headline  bench.py default workload (USPS-shaped 9298 x 256, k = 4), + idle gaps of one round (tools/round_gaps.py)
#   k8        tools/scale_probe.py 25000 512 8 (BASELINE configs[2] shape: the t = 7, 8 lattice sums)
#   general   bench.py --label-prob 0.5 --mistake-prob 0.25
#   mcmi6     tools/mcmi_split_bench.py 6
# Summaries land in gpurun_out/prof_r3/ (copy the r3_* files to profiles/).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITAL_BENCH_NO_EXTRAS=1
WHICH=${@:-headline k8 general mcmi6}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
passes() {   # name, program and arguments...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -o stats -- "$@" > $OUT/${name}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}_fetch -o fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}_write -o write -- "$@" > $OUT/${name}_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/${name}_sq -o sq -- "$@" > $OUT/${name}_sq.log 2>&1
  cp $(find $OUT/${name}_stats -name "*kernel_stats.csv" | head -1) $OUT/r3_${name}_kernel_stats.csv
  python3 $ROOT/tools/pmc_summary.py $(find $OUT/${name}_fetch $OUT/${name}_write $OUT/${name}_sq -name "*counter_collection.csv") > $OUT/r3_${name}_pmc_summary.csv
}
for w in $WHICH; do
  case $w in
    headline) passes headline python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scaling-workload
              grep '^{"metric"' $OUT/headline_stats.log > $OUT/r3_headline_bench_under_rocprof.json
              python3 $ROOT/tools/round_gaps.py $(find $OUT/headline_stats -name "*kernel_trace.csv" | head -1) $OUT/r3_round_gaps.json > $OUT/r3_round_gaps.txt ;;
    k8)       passes k8 python3 $ROOT/tools/scale_probe.py 25000 512 8
              grep -v "^[EW]20" $OUT/k8_stats.log | tail -4 > $OUT/r3_k8_probe.log ;;
    general)  passes general python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-workload --label-prob 0.5 --mistake-prob 0.25 ;;
    mcmi6)    passes mcmi6 python3 $ROOT/tools/mcmi_split_bench.py 6 ;;
  esac
done
ls -la $OUT/r3_* 2>/dev/null
