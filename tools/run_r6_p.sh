#!/bin/bash
# Host threads of the pattern sampling x range plan (config 5's share): does the LAPACK pool compete with the HIP runtime's own
# threads for the box's 16 CPUs while the GPU runs?  Two runs each, interleaved.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for rep in 1 2; do for plan in "10 16" "10 12" "10 8" "7 16" "7 12" "7 8"; do
  set -- $plan
  a=$(ITAL_MC_CHUNK_FROM=$1 ITAL_HOST_THREADS=$2 timeout 300 python3 tools/scale_probe.py 125000 512 16 1 2>&1 | grep "fetch_unlabelled(16)" | sed 's/.*512: \([0-9.]*\) s.*/\1/')
  echo "rep$rep chunk_from=$1 host_threads=$2: $a s"
done; done
ITAL_MC_CHUNK_FROM=7 ITAL_HOST_THREADS=12 ITAL_MC_TIMING=1 timeout 300 python3 tools/scale_probe.py 125000 512 16 1 2>&1 | grep "^t=[789] range" | cut -c1-100
