#!/bin/bash
# Round 6, GPU run L: the driver's own bench command (all default workloads; without the CPU baselines, which fork worker pools and
# must not run under the profiler's preloaded runtime) under rocprofv3 --kernel-trace --stats: one kernel-stats table for the line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_l
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench_stats -o stats -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
echo "rc $?" >> $OUT/bench_under_rocprof.log
cp $(find $OUT/bench_stats -name "*kernel_stats.csv" | head -1) $OUT/r6_bench_default_kernel_stats.csv
grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/r6_bench_default_under_rocprof.json
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
head -n 12 $OUT/r6_bench_default_kernel_stats.csv | cut -c1-170
tail -c 600 $OUT/r6_bench_default_under_rocprof.json
