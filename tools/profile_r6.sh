#!/bin/bash
# Round-6 profiles on the GPU box (run through gpurun from the repo root): per workload one kernel-trace + stats run, then
# the PMC counters in runs of their own (FETCH_SIZE, WRITE_SIZE, SQ_*: never combined with sys/hip/hsa tracing).
#   headline  bench.py default workload (USPS-shaped 9298 x 256, k = 4), + idle gaps of one round (tools/round_gaps.py)
#   k8        tools/scale_probe.py 25000 512 8 (BASELINE configs[2] shape: the t = 7, 8 lattice sums)
#   general   bench.py --label-prob 0.5 --mistake-prob 0.25 (noisy user: the pipeline of gen_pipeline.hip)
#   c5        tools/scale_probe.py 125000 512 16 1 (BASELINE configs[4] as one of 8 ranks sees it: gen_main_kernel<3..16>)
#   mcmi      tools/mcmi_bench.py 1000 (MCMI_min at the reference's subsample: the size bench.py times; counters per launch of ONE size)
#   mcmiall   tools/mcmi_bench.py all (all 9298 candidates)
#   kcols     tools/kcols_probe.py (the HBM-bound streaming kernel at 1M rows: bench.py's roofline_hbm probe)
#   cesub     tools/cesub_bench.py (change_estimation_subset: the monolithic score_generic_kernel)
# Summaries land in gpurun_out/prof_r6/ together with r6_stamp.json (tools/stamp.py: the kernel sources they were taken
# with); copy the r6_* files to profiles/.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_r6
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export ITAL_BENCH_NO_EXTRAS=1
WHICH=${@:-headline k8 general c5 mcmi kcols}
SQ="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU"
SQ3="SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU"
SQ2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES"
passes() {   # name, program and arguments...
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${name}_stats -o stats -- "$@" > $OUT/${name}_stats.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${name}_fetch -o fetch -- "$@" > $OUT/${name}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${name}_write -o write -- "$@" > $OUT/${name}_write.log 2>&1
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/${name}_sq -o sq -- "$@" > $OUT/${name}_sq.log 2>&1
  rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $OUT/${name}_sq2 -o sq2 -- "$@" > $OUT/${name}_sq2.log 2>&1
  if [ "$name" = "c5" ]; then   # the scalar unit's share (round-5 verdict, item 3)
    rocprofv3 --pmc $SQ3 --kernel-trace --output-format csv -d $OUT/${name}_sq3 -o sq3 -- "$@" > $OUT/${name}_sq3.log 2>&1
  fi
  cp $(find $OUT/${name}_stats -name "*kernel_stats.csv" | head -1) $OUT/r6_${name}_kernel_stats.csv
  python3 $ROOT/tools/pmc_summary.py $(find $OUT/${name}_fetch $OUT/${name}_write $OUT/${name}_sq $OUT/${name}_sq2 $OUT/${name}_sq3 -name "*counter_collection.csv" 2>/dev/null) > $OUT/r6_${name}_pmc_summary.csv
  python3 $ROOT/tools/stamp.py $OUT/r6_stamp.json $OUT/r6_${name}_kernel_stats.csv $OUT/r6_${name}_pmc_summary.csv
}
for w in $WHICH; do
  case $w in
    headline) passes headline python3 $ROOT/bench.py --quick --steps 10 --warmup 2 --no-cpu-baseline --no-scaling-workload
              grep '^{"metric"' $OUT/headline_stats.log > $OUT/r6_headline_bench_under_rocprof.json
              python3 $ROOT/tools/round_gaps.py $(find $OUT/headline_stats -name "*kernel_trace.csv" | head -1) $OUT/r6_round_gaps.json > $OUT/r6_round_gaps.txt
              python3 $ROOT/tools/stamp.py $OUT/r6_stamp.json $OUT/r6_round_gaps.json $OUT/r6_headline_bench_under_rocprof.json ;;
    k8)       passes k8 python3 $ROOT/tools/scale_probe.py 25000 512 8
              grep -v "^[EW]20" $OUT/k8_stats.log | tail -4 > $OUT/r6_k8_probe.log ;;
    general)  passes general python3 $ROOT/bench.py --quick --steps 3 --warmup 1 --no-cpu-baseline --no-scaling-workload --label-prob 0.5 --mistake-prob 0.25 ;;
    c5)       passes c5 python3 $ROOT/tools/scale_probe.py 125000 512 16 1
              grep -v "^[EW]20" $OUT/c5_stats.log | tail -4 > $OUT/r6_c5_probe.log
              python3 $ROOT/tools/step_shares.py $(find $OUT/c5_stats -name "*kernel_trace.csv" | head -1) > $OUT/r6_c5_step_shares.txt ;;
    mcmi)     passes mcmi python3 $ROOT/tools/mcmi_bench.py 1000
              grep -v "^[EW]20" $OUT/mcmi_stats.log | tail -3 > $OUT/r6_mcmi_probe.log ;;
    mcmiall)  passes mcmiall python3 $ROOT/tools/mcmi_bench.py all
              grep -v "^[EW]20" $OUT/mcmiall_stats.log | tail -3 > $OUT/r6_mcmiall_probe.log ;;
    kcols)    passes kcols python3 $ROOT/tools/kcols_probe.py
              grep '^{"bound"' $OUT/kcols_stats.log > $OUT/r6_kcols_probe.json ;;
    cesub)    passes cesub python3 $ROOT/tools/cesub_bench.py
              grep -v "^[EW]20" $OUT/cesub_stats.log | tail -6 > $OUT/r6_cesub_probe.log ;;
  esac
done
# only the summaries travel back (the raw traces are hundreds of MB)
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $OUT/r6_* 2>/dev/null
