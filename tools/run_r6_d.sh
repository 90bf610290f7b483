#!/bin/bash
# Round 6, GPU run D: the paced progress-deadline test, the host-side split of config 5's gaps between greedy steps
# (ITAL_MC_TIMING), then fuzz campaigns over ALL kinds with the final build, six side by side.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_d
mkdir -p $OUT
cd $ROOT
timeout 600 python3 -m pytest tests/test_gpu_exchange_timeout.py -m gpu -q > $OUT/gputests_exchange.log 2>&1
echo "pytest rc $?" >> $OUT/gputests_exchange.log
ITAL_MC_TIMING=1 timeout 300 python3 tools/scale_probe.py 125000 512 16 1 2>&1 | grep -v "^[EW]20\|amdgpu.ids" > $OUT/c5_mc_timing.log
LIMIT=${FUZZ_LIMIT:-900}
run() {  # tag kinds seed cases maxd
  local tag=$1 kinds=$2 seed=$3 cases=$4 maxd=$5
  ( [ -n "$maxd" ] && export FUZZ_MAX_D=$maxd; [ -n "$kinds" ] && export FUZZ_KINDS=$kinds; timeout $LIMIT python3 tools/fuzz_parity.py $cases $seed > $OUT/fuzz_${tag}_seed${seed}.log 2>&1 ) &
}
run all "" 401 700 ""
run all "" 409 700 ""
run noisy_mix "noisy,motivated,mix,optnoisy" 419 500 ""
run lowd "" 421 700 3
run optclip_optbig "optclip,optbig" 431 500 ""
run mc_mcwide "mc,mcwide,optwide" 433 250 ""
wait
tail -n 8 $OUT/gputests_exchange.log
grep "^t=" $OUT/c5_mc_timing.log | grep -v range | head -n 20
for f in $OUT/fuzz_*.log; do echo "== $f"; grep -c " ok" $f; grep -v " ok" $f | grep -v amdgpu.ids | tail -n 6; done
