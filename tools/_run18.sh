set -u
mkdir -p gpurun_out
export PYTHONHASHSEED=0
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
( timeout 2400 python tools/fuzz_parity.py 1200 239 > gpurun_out/r5_fuzz_allkinds_seed239.log 2>&1; tail -1 gpurun_out/r5_fuzz_allkinds_seed239.log; grep -c "^case" gpurun_out/r5_fuzz_allkinds_seed239.log )
