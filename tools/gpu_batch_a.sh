#!/bin/bash
# scratch batch: A/B of library variants, host profile, kernel stats
cd $GRAFT_REPO_ROOT
STEPS=10 tools/variant_bench.sh > gpurun_out/r2_variants_a.log 2>&1
python tools/round_breakdown.py > gpurun_out/r2_breakdown_a.log 2>&1
python tools/host_profile.py > gpurun_out/r2_hostprof_a.log 2>&1
cd /tmp && export TMPDIR=/tmp
export ITAL_BENCH_NO_EXTRAS=1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2_stats_a -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2_stats_a.log 2>&1
