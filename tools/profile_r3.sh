#!/bin/bash
# Round-3 GPU session script (run through gpurun from the repo root): stages chosen by name.
#   tests      python -m pytest tests -m gpu  -> gpurun_out/r3/gpu_tests_<TAG>.log
#   trace      headline kernel trace + per-round idle gaps (tools/round_gaps.py)
#   variants   tools/variant_bench.sh over build_variants/ (BENCH_ARGS, STEPS)
#   k8stats    kernel stats of one k = 8 round at 25 000 x 512 (tools/scale_probe.py)
#   bench      the default bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${TAG:-a}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
export TMPDIR=/tmp
for stage in "$@"; do
  case $stage in
    tests)    (cd $ROOT && timeout ${TEST_TIMEOUT:-1500} python -m pytest tests -m gpu -x -q ${TEST_ARGS:-} > $OUT/gpu_tests_$TAG.log 2>&1; tail -5 $OUT/gpu_tests_$TAG.log) ;;
    trace)    (cd /tmp && ITAL_BENCH_NO_EXTRAS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$TAG -o t -- python3 $ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-scaling-workload > $OUT/trace_$TAG.log 2>&1
               python3 $ROOT/tools/round_gaps.py $(find $OUT/trace_$TAG -name "*kernel_trace.csv" | head -1) > $OUT/round_gaps_$TAG.txt 2>&1
               cp $(find $OUT/trace_$TAG -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats_$TAG.csv; tail -3 $OUT/round_gaps_$TAG.txt) ;;
    variants) (cd $ROOT && ITAL_BENCH_NO_EXTRAS=1 BENCH_ARGS="${BENCH_ARGS:---no-scaling-workload}" tools/variant_bench.sh > $OUT/variants_$TAG.txt 2>&1; cat $OUT/variants_$TAG.txt) ;;
    k8stats)  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/k8_$TAG -o t -- python3 $ROOT/tools/scale_probe.py 25000 512 8 > $OUT/k8_probe_$TAG.log 2>&1
               cp $(find $OUT/k8_$TAG -name "*kernel_stats.csv" | head -1) $OUT/k8_kernel_stats_$TAG.csv; head -12 $OUT/k8_kernel_stats_$TAG.csv; tail -3 $OUT/k8_probe_$TAG.log) ;;
    bench)    (cd $ROOT && python bench.py ${BENCH_ARGS:-} > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err; tail -c 600 $OUT/bench_$TAG.json) ;;
  esac
done
