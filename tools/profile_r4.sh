#!/bin/bash
# Round-4 profiles on the GPU box (run through gpurun from the repo root): per workload one kernel-trace + stats run, then
# the PMC counters in runs of their own (FETCH_SIZE, WRITE_SIZE, SQ_*: never combined with sys/hip/hsa tracing).
#   headline  bench.py default workload (USPS-shaped 9298 x 256, k = 4), + idle gaps of one round (tools/round_gaps.py)
#   k8        tools/scale_probe.py 25000 512 8 (BASELINE configs[2] shape: the t = 7, 8 lattice sums)
#   general   bench.py --label-prob 0.5 --mistake-prob 0.25 (noisy user: the pipeline of gen_pipeline.hip)
# Summaries land in gpurun_out/prof_r4/ together with r4_stamp.json (tools/stamp.py: the kernel sources they were taken
# with); copy the r4_* files to profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r4
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITAL_BENCH_NO_EXTRAS=1
WHICH=${@:-headline k8 general}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
passes() {   # name, program and arguments...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -o stats -- "$@" > $OUT/${name}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}_fetch -o fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}_write -o write -- "$@" > $OUT/${name}_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/${name}_sq -o sq -- "$@" > $OUT/${name}_sq.log 2>&1
  cp $(find $OUT/${name}_stats -name "*kernel_stats.csv" | head -1) $OUT/r4_${name}_kernel_stats.csv
  python3 $ROOT/tools/pmc_summary.py $(find $OUT/${name}_fetch $OUT/${name}_write $OUT/${name}_sq -name "*counter_collection.csv") > $OUT/r4_${name}_pmc_summary.csv
  python3 $ROOT/tools/stamp.py $OUT/r4_stamp.json $OUT/r4_${name}_kernel_stats.csv $OUT/r4_${name}_pmc_summary.csv
}
for w in $WHICH; do
  case $w in
    headline) passes headline python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scaling-workload
              grep '^{"metric"' $OUT/headline_stats.log > $OUT/r4_headline_bench_under_rocprof.json
              python3 $ROOT/tools/round_gaps.py $(find $OUT/headline_stats -name "*kernel_trace.csv" | head -1) $OUT/r4_round_gaps.json > $OUT/r4_round_gaps.txt
              python3 $ROOT/tools/stamp.py $OUT/r4_stamp.json $OUT/r4_round_gaps.json $OUT/r4_headline_bench_under_rocprof.json ;;
    k8)       passes k8 python3 $ROOT/tools/scale_probe.py 25000 512 8
              grep -v "^[EW]20" $OUT/k8_stats.log | tail -4 > $OUT/r4_k8_probe.log ;;
    general)  passes general python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-workload --label-prob 0.5 --mistake-prob 0.25 ;;
  esac
done
# only the summaries travel back (the raw traces are hundreds of MB)
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $OUT/r4_* 2>/dev/null
