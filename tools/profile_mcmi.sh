#!/bin/bash
# rocprofv3 passes over tools/mcmi_bench.py (MCMI_min on 1000 and 9273 candidates): kernel stats, then MFMA / VALU counters.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_mcmi_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $ROOT/tools/mcmi_bench.py > $OUT/stats.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/sq -o sq -- python3 $ROOT/tools/mcmi_bench.py > $OUT/sq.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -o fetch -- python3 $ROOT/tools/mcmi_bench.py > $OUT/fetch.log 2>&1
ls $OUT/*/
