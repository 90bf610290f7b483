set -u
mkdir -p gpurun_out
export PYTHONHASHSEED=0
timeout 2400 bash tools/profile_r5.sh headline k8 general c5 mcmi kcols cesub > gpurun_out/r5_profile_run.log 2>&1; echo "profile rc=$?"
cat gpurun_out/prof_r5/r5_c5_step_shares.txt | tail -3
# the bench quotes the fresh counters when they sit in profiles/ with their stamp
cp gpurun_out/prof_r5/r5_*.csv gpurun_out/prof_r5/r5_*.json gpurun_out/prof_r5/r5_*.txt gpurun_out/prof_r5/r5_*.log profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
( ITAL_BENCH_BACKEND=gloo ITAL_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r5_bench_2rank_selflaunch.json 2> gpurun_out/r5_bench_2rank_selflaunch.err; echo "selflaunch rc=$?" )
( timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gputests_final.log; tail -3 gpurun_out/r5_gputests_final.log )
( FUZZ_KINDS=mc,noisy timeout 2300 python tools/fuzz_parity.py 1200 227 > gpurun_out/r5_fuzz_mc_noisy_seed227.log 2>&1; tail -1 gpurun_out/r5_fuzz_mc_noisy_seed227.log )
