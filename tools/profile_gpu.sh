#!/bin/bash
# Profiles bench.py on the GPU box (run through gpurun from the repo root):
#   pass 1: kernel trace + stats;  passes 2-4: PMC counters, each in its own run (FETCH_SIZE and WRITE_SIZE do not fit
#   one pass on gfx950; never combined with sys/hip/hsa tracing).
# Outputs land in gpurun_out/prof_<tag>/; summarise with tools/rocpd_stats.py / tools/pmc_summary.py.
set -u
TAG=${1:-r1}
shift || true
ARGS=${@:---steps 4 --warmup 1 --no-cpu-baseline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITAL_BENCH_NO_EXTRAS=1   # profile the headline workload only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/bench.py $ARGS > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/bench.py $ARGS > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -o write -- python3 $ROOT/bench.py $ARGS > $OUT/write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $ROOT/bench.py $ARGS > $OUT/sq.log 2>&1
find $OUT -name "*.csv" | head -20
