set -u
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
( timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gputests_final.log; tail -3 gpurun_out/r5_gputests_final.log )
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
