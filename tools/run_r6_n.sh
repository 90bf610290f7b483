#!/bin/bash
# A/B at full size: config 5's share (125 000 x 512, k = 16, mc = 1) and C3' (25 000 x 512, k = 8) per library variant, three times each,
# interleaved (base, p1, p2, base, ...) so that clock / box drift hits all alike.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2 3; do for v in ${VARIANTS:-base p1 p2}; do
  a=$(ITAL_HIP_LIB=$ROOT/build_variants/libital_$v.so timeout 300 python3 tools/scale_probe.py 125000 512 16 1 2>&1 | grep "fetch_unlabelled(16)" | sed 's/.*512: \([0-9.]*\) s.*/\1/')
  b=$(ITAL_HIP_LIB=$ROOT/build_variants/libital_$v.so timeout 300 python3 tools/scale_probe.py 25000 512 8 2>&1 | grep "fetch_unlabelled(8)" | sed 's/.*512: \([0-9.]*\) s.*/\1/')
  echo "$v rep$rep  c5share $a s   c3 $b s"
done; done
