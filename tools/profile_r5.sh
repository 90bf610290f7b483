#!/bin/bash
# Round-5 profiles on the GPU box (run through gpurun from the repo root): per workload one kernel-trace + stats run, then
# the PMC counters in runs of their own (FETCH_SIZE, WRITE_SIZE, SQ_*: never combined with sys/hip/hsa tracing).
#   headline  bench.py default workload (USPS-shaped 9298 x 256, k = 4), + idle gaps of one round (tools/round_gaps.py)
#   k8        tools/scale_probe.py 25000 512 8 (BASELINE configs[2] shape: the t = 7, 8 lattice sums)
#   general   bench.py --label-prob 0.5 --mistake-prob 0.25 (noisy user: the pipeline of gen_pipeline.hip)
#   c5        tools/scale_probe.py 125000 512 16 1 (BASELINE configs[4] as one of 8 ranks sees it: gen_main_kernel<3..16>)
#   mcmi      tools/mcmi_bench.py (MCMI_min, subsample 1000 and all candidates)
#   kcols     tools/kcols_probe.py (the HBM-bound streaming kernel at 1M rows: bench.py's roofline_hbm probe)
#   cesub     tools/cesub_bench.py (change_estimation_subset: the monolithic score_generic_kernel)
# Summaries land in gpurun_out/prof_r5/ together with r5_stamp.json (tools/stamp.py: the kernel sources they were taken
# with); copy the r5_* files to profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITAL_BENCH_NO_EXTRAS=1
WHICH=${@:-headline k8 general c5 mcmi kcols}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"
passes() {   # name, program and arguments...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -o stats -- "$@" > $OUT/${name}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}_fetch -o fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}_write -o write -- "$@" > $OUT/${name}_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/${name}_sq -o sq -- "$@" > $OUT/${name}_sq.log 2>&1
  rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/${name}_sq2 -o sq2 -- "$@" > $OUT/${name}_sq2.log 2>&1
  cp $(find $OUT/${name}_stats -name "*kernel_stats.csv" | head -1) $OUT/r5_${name}_kernel_stats.csv
  python3 $ROOT/tools/pmc_summary.py $(find $OUT/${name}_fetch $OUT/${name}_write $OUT/${name}_sq $OUT/${name}_sq2 -name "*counter_collection.csv") > $OUT/r5_${name}_pmc_summary.csv
  python3 $ROOT/tools/stamp.py $OUT/r5_stamp.json $OUT/r5_${name}_kernel_stats.csv $OUT/r5_${name}_pmc_summary.csv
}
for w in $WHICH; do
  case $w in
    headline) passes headline python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scaling-workload
              grep '^{"metric"' $OUT/headline_stats.log > $OUT/r5_headline_bench_under_rocprof.json
              python3 $ROOT/tools/round_gaps.py $(find $OUT/headline_stats -name "*kernel_trace.csv" | head -1) $OUT/r5_round_gaps.json > $OUT/r5_round_gaps.txt
              python3 $ROOT/tools/stamp.py $OUT/r5_stamp.json $OUT/r5_round_gaps.json $OUT/r5_headline_bench_under_rocprof.json ;;
    k8)       passes k8 python3 $ROOT/tools/scale_probe.py 25000 512 8
              grep -v "^[EW]20" $OUT/k8_stats.log | tail -4 > $OUT/r5_k8_probe.log ;;
    general)  passes general python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-workload --label-prob 0.5 --mistake-prob 0.25 ;;
    c5)       passes c5 python3 $ROOT/tools/scale_probe.py 125000 512 16 1
              grep -v "^[EW]20" $OUT/c5_stats.log | tail -4 > $OUT/r5_c5_probe.log
              python3 $ROOT/tools/step_shares.py $(find $OUT/c5_stats -name "*kernel_trace.csv" | head -1) > $OUT/r5_c5_step_shares.txt ;;
    mcmi)     passes mcmi python3 $ROOT/tools/mcmi_bench.py
              grep -v "^[EW]20" $OUT/mcmi_stats.log | tail -3 > $OUT/r5_mcmi_probe.log ;;
    kcols)    passes kcols python3 $ROOT/tools/kcols_probe.py
              grep '^{"bound"' $OUT/kcols_stats.log > $OUT/r5_kcols_probe.json ;;
    cesub)    passes cesub python3 $ROOT/tools/cesub_bench.py
              grep -v "^[EW]20" $OUT/cesub_stats.log | tail -6 > $OUT/r5_cesub_probe.log ;;
  esac
done
# only the summaries travel back (the raw traces are hundreds of MB)
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $OUT/r5_* 2>/dev/null
