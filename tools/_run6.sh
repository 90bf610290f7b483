set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests_2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gputests_2.log )
tail -4 gpurun_out/r5_gputests_2.log
timeout 1500 python bench.py --steps 20 --warmup 3 --extra c4,c5k16 > gpurun_out/r5_bench_b.json 2> gpurun_out/r5_bench_b.err; echo "bench rc=$?"
tail -c 300 gpurun_out/r5_bench_b.json
