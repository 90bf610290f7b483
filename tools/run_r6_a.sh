#!/bin/bash
# Round 6, GPU run A: the whole GPU suite, the default bench line (+ the N = 1 time of the scaling workload for
# profiles/scaling_picks_n1.json), then six fuzz campaigns of the label_estimation kinds side by side.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_a
mkdir -p $OUT
cd $ROOT
( nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null; python3 -c "import os; print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())"; free -g | head -2 ) > $OUT/box.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -x -q > $OUT/gputests.log 2>&1
echo "pytest rc $?" >> $OUT/gputests.log
ITAL_BENCH_WRITE_PICKS=$OUT/scaling_picks_n1.json timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
LIMIT=${FUZZ_LIMIT:-700}
run() {  # kinds seed cases maxd
  local kinds=$1 seed=$2 cases=$3 maxd=$4
  ( [ -n "$maxd" ] && export FUZZ_MAX_D=$maxd; FUZZ_KINDS=$kinds timeout $LIMIT python3 tools/fuzz_parity.py $cases $seed > $OUT/fuzz_${kinds//,/_}_seed${seed}.log 2>&1 ) &
}
run optnoisy,optbig 311 450 3
run optnoisy,optbig 313 450 ""
run optclip 317 450 3
run optclip 331 350 ""
run optwide 337 160 3
run optwide 347 160 ""
wait
tail -n 3 $OUT/gputests.log
tail -c 1500 $OUT/bench_default.json
for f in $OUT/fuzz_*.log; do echo "== $f"; grep -c " ok" $f; grep -v " ok" $f | grep -v amdgpu.ids | tail -n 6; done
cat $OUT/box.txt
