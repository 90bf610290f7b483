set -u
mkdir -p gpurun_out
export PYTHONHASHSEED=0
rm -f gpurun_out/prof_r5/r5_stamp.json
timeout 2400 bash tools/profile_r5.sh headline k8 general c5 mcmi kcols cesub > gpurun_out/r5_profile_run.log 2>&1; echo "profile rc=$?"
tail -1 gpurun_out/prof_r5/r5_c5_step_shares.txt
cp gpurun_out/prof_r5/r5_*.csv gpurun_out/prof_r5/r5_*.json gpurun_out/prof_r5/r5_*.txt gpurun_out/prof_r5/r5_*.log profiles/ 2>/dev/null
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
( ITAL_BENCH_BACKEND=gloo ITAL_BENCH_ONE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 5 --warmup 1 > gpurun_out/r5_bench_2rank_selflaunch.json 2> gpurun_out/r5_bench_2rank_selflaunch.err; echo "selflaunch rc=$?" )
timeout 900 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-scaling-workload --extra c4,c5k16 > gpurun_out/r5_bench_extra_c4_c5k16.json 2> gpurun_out/r5_bench_extra.err; echo "extra rc=$?"
