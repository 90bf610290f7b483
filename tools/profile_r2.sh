#!/bin/bash
# Round-2 profiles on the GPU box (run through gpurun from the repo root): for each workload one kernel-trace + stats run,
# then the PMC counters in runs of their own (FETCH_SIZE, WRITE_SIZE, SQ_*: never combined with sys/hip/hsa tracing).
#   headline  bench.py default (USPS-shaped 9298 x 256, k = 4, perfect user)
#   general   bench.py --label-prob 0.5 --mistake-prob 0.25 (the general scorer, noisy user)
#   mcmi      tools/mcmi_bench.py
#   c5share   tools/scale_probe.py 125000 512 16 1 (kernel stats only: 30 s per run)
# Summaries (the files bench.py names, to be copied to profiles/) land in gpurun_out/prof_r2/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITAL_BENCH_NO_EXTRAS=1
WHICH=${@:-headline general mcmi c5share}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
passes() {   # name, program and arguments...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -o stats -- "$@" > $OUT/${name}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}_fetch -o fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}_write -o write -- "$@" > $OUT/${name}_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/${name}_sq -o sq -- "$@" > $OUT/${name}_sq.log 2>&1
  cp $(find $OUT/${name}_stats -name "*kernel_stats.csv" | head -1) $OUT/r2_${name}_kernel_stats.csv
  python3 $ROOT/tools/pmc_summary.py $(find $OUT/${name}_fetch $OUT/${name}_write $OUT/${name}_sq -name "*counter_collection.csv") > $OUT/r2_${name}_pmc_summary.csv
}
for w in $WHICH; do
  case $w in
    headline) passes headline python3 $ROOT/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-scaling-workload
              grep '^{"metric"' $OUT/headline_stats.log > $OUT/r2_headline_bench_under_rocprof.json ;;
    general)  passes general python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-workload --label-prob 0.5 --mistake-prob 0.25 ;;
    mcmi)     passes mcmi python3 $ROOT/tools/mcmi_bench.py ;;
    c5share)  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5share_stats -o stats -- python3 $ROOT/tools/scale_probe.py 125000 512 16 1 > $OUT/r2_c5share_probe.log 2>&1
              cp $(find $OUT/c5share_stats -name "*kernel_stats.csv" | head -1) $OUT/r2_c5share_kernel_stats.csv ;;
  esac
done
ls -la $OUT/*.csv $OUT/*.json 2>/dev/null
