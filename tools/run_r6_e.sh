#!/bin/bash
# Round 6, GPU run E (closing run): the whole GPU suite with the final tree, the 2-rank rehearsal of `bench.py --gpus 2` on the
# one-GPU box (gloo, both ranks on cuda:0: the launch path, the N > 1 line and its keys), the default bench line.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_e
mkdir -p $OUT
cd $ROOT
timeout 1800 python3 -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1
echo "pytest rc $?" >> $OUT/gputests.log
ITAL_BENCH_ONE_DEVICE=1 ITAL_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err
echo "bench2 rc $?" >> $OUT/bench_2rank_gloo.err
timeout 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc $?" >> $OUT/bench_default.err
tail -n 6 $OUT/gputests.log
tail -n 3 $OUT/bench_2rank_gloo.err
tail -c 1500 $OUT/bench_2rank_gloo.json
echo
tail -c 1500 $OUT/bench_default.json
