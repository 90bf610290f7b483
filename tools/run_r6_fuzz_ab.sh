#!/bin/bash
# Round 6, A/B of the exact-equality recompute (label_estimation 'optimistic' / 'pessimistic' on the general scorer): the same
# fuzz campaigns (kinds optclip / optwide / optnoisy / optbig at d <= 3) against the round-5 library (build_variants/libital_r5.so:
# flat sums in the single kernel and above 8 variables) and against the tree's.  Six processes side by side (the oracle is
# the slow side: one host core each).   gpurun --timeout 2400 -- 'bash tools/run_r6_fuzz_ab.sh'
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_fuzz_ab
mkdir -p $OUT
cd $ROOT
LIMIT=${FUZZ_LIMIT:-1500}
run() {  # tag lib kinds seed cases
  local tag=$1 lib=$2 kinds=$3 seed=$4 cases=$5
  ( [ -n "$lib" ] && export ITAL_HIP_LIB=$lib; FUZZ_KINDS=$kinds FUZZ_MAX_D=3 timeout $LIMIT python3 tools/fuzz_parity.py $cases $seed > $OUT/${tag}_${kinds//,/_}_seed$seed.log 2>&1 ) &
}
run new "" optclip 301 150
run new "" optwide 303 60
run new "" optnoisy,optbig 307 300
run old $ROOT/build_variants/libital_r5.so optclip 301 150
run old $ROOT/build_variants/libital_r5.so optwide 303 60
run old $ROOT/build_variants/libital_r5.so optnoisy,optbig 307 300
wait
for f in $OUT/*.log; do echo "== $f"; grep -c " ok" $f; grep -v " ok" $f | tail -n 12; done
