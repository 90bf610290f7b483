#!/bin/bash
# Round 6, GPU run B: whole GPU suite (all failures listed), the two optclip stream cases against the round-5 library, the
# noisy-user round with the device-sized chunk launches, config 5's share under a kernel trace (true idle per greedy step:
# tools/step_shares.py) and with other range plans of the pattern sampling.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r6_b
mkdir -p $OUT
cd $ROOT
timeout 1500 python3 -m pytest tests -m gpu -q > $OUT/gputests.log 2>&1
echo "pytest rc $?" >> $OUT/gputests.log
for c in 12 133; do
  ITAL_HIP_LIB=$ROOT/build_variants/libital_r5.so FUZZ_KINDS=optclip FUZZ_MAX_D=3 timeout 300 python3 tools/fuzz_parity.py 450 317 $c 2>&1 | grep -v "cand \|^round" | tail -n 3 > $OUT/old_optclip_317_case$c.log
  FUZZ_KINDS=optclip FUZZ_MAX_D=3 timeout 300 python3 tools/fuzz_parity.py 450 317 $c 2>&1 | grep -v "cand \|^round" | tail -n 3 > $OUT/new_optclip_317_case$c.log
done
timeout 300 python3 bench.py --quick --label-prob 0.5 --mistake-prob 0.25 --no-cpu-baseline --no-scaling-workload --steps 6 --warmup 2 > $OUT/bench_noisy.json 2> $OUT/bench_noisy.err
ITAL_HIP_LIB=$ROOT/build_variants/libital_r5.so timeout 300 python3 bench.py --quick --label-prob 0.5 --mistake-prob 0.25 --no-cpu-baseline --no-scaling-workload --steps 6 --warmup 2 > $OUT/bench_noisy_r5lib.json 2> $OUT/bench_noisy_r5lib.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/general_stats -o stats -- python3 $ROOT/bench.py --quick --label-prob 0.5 --mistake-prob 0.25 --no-cpu-baseline --no-scaling-workload --steps 3 --warmup 1 > $OUT/general_stats.log 2>&1
cp $(find $OUT/general_stats -name "*kernel_stats.csv" | head -1) $OUT/r6_general_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -o stats -- python3 $ROOT/tools/scale_probe.py 125000 512 16 1 > $OUT/c5_stats.log 2>&1
cp $(find $OUT/c5_stats -name "*kernel_stats.csv" | head -1) $OUT/r6_c5_kernel_stats_b.csv
python3 $ROOT/tools/step_shares.py $(find $OUT/c5_stats -name "*kernel_trace.csv" | head -1) > $OUT/r6_c5_step_shares_b.txt
cd $ROOT
for plan in "4 10" "4 7" "5 7" "5 4" "6 3"; do
  set -- $plan
  ITAL_MC_CHUNKS=$1 ITAL_MC_CHUNK_FROM=$2 timeout 300 python3 tools/scale_probe.py 125000 512 16 1 2>&1 | grep -v "^[EW]20\|amdgpu.ids" | tail -n 3 > $OUT/c5_plan_$1_$2.log
done
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
tail -n 15 $OUT/gputests.log
cat $OUT/*_case*.log
tail -c 600 $OUT/bench_noisy.json; echo; tail -c 600 $OUT/bench_noisy_r5lib.json; echo
grep "gen_main_kernel<4\|fillBuffer\|gen_build" $OUT/r6_general_kernel_stats.csv | cut -c1-160
tail -n 3 $OUT/r6_c5_step_shares_b.txt
for f in $OUT/c5_plan_*.log; do echo $f; head -n 2 $f | cut -c1-200; done
