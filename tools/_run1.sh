set -u
mkdir -p gpurun_out
( ITAL_BENCH_BACKEND=gloo ITAL_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r5_bench_2rank_selflaunch.json 2> gpurun_out/r5_bench_2rank_selflaunch.err; echo "selflaunch rc=$?" )
tail -c 600 gpurun_out/r5_bench_2rank_selflaunch.json
timeout 1500 bash tools/profile_r5.sh c5 mcmi kcols
cat gpurun_out/prof_r5/r5_c5_step_shares.txt
