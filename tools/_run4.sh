set -u
mkdir -p gpurun_out
( timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests_1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gputests_1.log )
tail -5 gpurun_out/r5_gputests_1.log
python tools/scale_probe.py 20000 64 16 1 2>&1 | grep "fetch_un\|score_generic" | cut -c1-1200
python tools/scale_probe.py 125000 512 16 1 2>&1 | grep "fetch_un\|score_generic" | cut -c1-1500
python tools/scale_probe.py 25000 512 8 2>&1 | grep "fetch_un\|qmc_" | cut -c1-600
timeout 900 python bench.py --steps 20 --warmup 3 > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err; echo "bench rc=$?"
tail -c 400 gpurun_out/r5_bench_a.json
