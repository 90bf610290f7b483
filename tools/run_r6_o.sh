#!/bin/bash
# Why do ranges at 7 .. 9 variables not pay?  Config 5's share with ITAL_MC_CHUNK_FROM=7 under a kernel trace (tools/step_shares.py)
# and with the host-side timing of every range (ITAL_MC_TIMING).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_o
mkdir -p $OUT
cd $ROOT
ITAL_MC_CHUNK_FROM=7 ITAL_MC_TIMING=1 timeout 300 python3 tools/scale_probe.py 125000 512 16 1 2>&1 | grep -v "^[EW]20\|amdgpu.ids" > $OUT/timing_from7.log
cd /tmp && export TMPDIR=/tmp
ITAL_MC_CHUNK_FROM=7 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $ROOT/tools/scale_probe.py 125000 512 16 1 > $OUT/trace.log 2>&1
python3 $ROOT/tools/step_shares.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) > $OUT/step_shares_from7.txt
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
grep "^t=[789]:\|^t=[789] " $OUT/timing_from7.log | cut -c1-150
cut -c1-75 $OUT/step_shares_from7.txt | head -n 20
tail -n 1 $OUT/step_shares_from7.txt | cut -c1-330
