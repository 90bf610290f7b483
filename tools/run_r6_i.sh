#!/bin/bash
# Does a degenerate case depend on what ran before it in the process?  (seed 461 case 532 reported a STREAM mismatch inside the
# campaign and none on its own.)  Cases 524 .. 532 of that campaign, twice.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_i
mkdir -p $OUT
cd $ROOT
for rep in 1 2; do
  FUZZ_FROM=524 FUZZ_KINDS=noisy,motivated,mix,subset,clip timeout 600 python3 tools/fuzz_parity.py 533 461 2>&1 | grep -v amdgpu > $OUT/from524_rep$rep.log
done
FUZZ_FROM=531 FUZZ_KINDS=noisy,motivated,mix,subset,clip timeout 600 python3 tools/fuzz_parity.py 533 461 2>&1 | grep -v amdgpu > $OUT/from531.log
FUZZ_KINDS=noisy,motivated,mix,subset,clip timeout 600 python3 tools/fuzz_parity.py 1500 461 532 2>&1 | grep -v amdgpu | grep "round\|cand" | head -n 80 > $OUT/only532_verbose.log
cat $OUT/from524_rep1.log $OUT/from524_rep2.log $OUT/from531.log | cut -c1-330
