#!/bin/bash
# Round 6, GPU run C: the paced progress-deadline test, the round's profiles (tools/profile_r6.sh: kernel stats + PMC passes per
# workload, stamped), then the default bench line quoting them.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_c
mkdir -p $OUT
cd $ROOT
timeout 600 python3 -m pytest tests/test_gpu_exchange_timeout.py -m gpu -q > $OUT/gputests_exchange.log 2>&1
echo "pytest rc $?" >> $OUT/gputests_exchange.log
bash tools/profile_r6.sh headline k8 general c5 mcmi kcols > $OUT/profile.log 2>&1
cp $ROOT/gpurun_out/prof_r6/r6_* $ROOT/profiles/ 2>/dev/null
python3 tools/stamp.py --check profiles/r6_stamp.json > $OUT/stamp_check.txt 2>&1
cd $ROOT
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
tail -n 6 $OUT/gputests_exchange.log
tail -n 12 $OUT/profile.log
cat $OUT/stamp_check.txt
tail -c 1800 $OUT/bench_default.json
