#!/bin/bash
# Round 6, GPU run H: the two "STREAM" cases of the long campaign (kind mix, a twin inside the change-estimation subset) with the
# tool's degenerate-round rule, against this tree's library and the round-5 one.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_h
mkdir -p $OUT
cd $ROOT
for lib in new old; do
  [ $lib = old ] && export ITAL_HIP_LIB=$ROOT/build_variants/libital_r5.so
  timeout 300 python3 tools/fuzz_parity.py 2000 443 468 2>&1 | grep -v "cand \|^round\|amdgpu" | tail -n 2 > $OUT/${lib}_all_443_case468.log
  FUZZ_KINDS=noisy,motivated,mix,subset,clip timeout 300 python3 tools/fuzz_parity.py 1500 461 532 2>&1 | grep -v "cand \|^round\|amdgpu" | tail -n 2 > $OUT/${lib}_noisy_461_case532.log
done
cat $OUT/*.log
