#!/bin/bash
# Round 6, GPU run K: second long fuzz campaign with the final build (other seeds; the baseline kinds on their own).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_k
mkdir -p $OUT
cd $ROOT
LIMIT=${FUZZ_LIMIT:-1500}
run() {  # tag kinds seed cases maxd
  local tag=$1 kinds=$2 seed=$3 cases=$4 maxd=$5
  ( [ -n "$maxd" ] && export FUZZ_MAX_D=$maxd; [ -n "$kinds" ] && export FUZZ_KINDS=$kinds; timeout $LIMIT python3 tools/fuzz_parity.py $cases $seed > $OUT/fuzz_${tag}_seed${seed}.log 2>&1 ) &
}
run all "" 467 2500 ""
run all "" 479 2500 ""
run base "mcmi,emoc,entropy,borderdiv,topcand,perfect" 487 2500 ""
run lowd "" 491 2500 3
run subset "subset,mix,clip,bigk" 499 1500 ""
run wide "mcwide,optwide,mc" 503 600 ""
wait
for f in $OUT/fuzz_*.log; do echo "== $f"; grep -c " ok" $f; grep -v " ok" $f | grep -v amdgpu.ids | tail -n 5; done
