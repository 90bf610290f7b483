#!/bin/bash
# Round 6, GPU run G: smoke() of the driver's entry point, then a long fuzz campaign with the final build (six processes).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_g
mkdir -p $OUT
cd $ROOT
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke rc $?" >> $OUT/smoke.log
LIMIT=${FUZZ_LIMIT:-1500}
run() {  # tag kinds seed cases maxd
  local tag=$1 kinds=$2 seed=$3 cases=$4 maxd=$5
  ( [ -n "$maxd" ] && export FUZZ_MAX_D=$maxd; [ -n "$kinds" ] && export FUZZ_KINDS=$kinds; timeout $LIMIT python3 tools/fuzz_parity.py $cases $seed > $OUT/fuzz_${tag}_seed${seed}.log 2>&1 ) &
}
run all "" 439 2000 ""
run all "" 443 2000 ""
run lowd "" 449 2000 3
run opt "optclip,optbig,optnoisy,optwide,optimistic" 457 1500 3
run noisy "noisy,motivated,mix,subset,clip" 461 1500 ""
run lowd2 "" 463 2000 2
wait
tail -n 3 $OUT/smoke.log
for f in $OUT/fuzz_*.log; do echo "== $f"; grep -c " ok" $f; grep -v " ok" $f | grep -v amdgpu.ids | tail -n 6; done
