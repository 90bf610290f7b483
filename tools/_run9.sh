set -u
mkdir -p gpurun_out
export PYTHONHASHSEED=0
( timeout 900 python -m pytest tests/test_gpu_scale.py::test_c5_share_125000x512_k16_monte_carlo tests/test_gpu_parity_limits.py tests/test_gpu_multirank_scale.py -x -q 2>&1 | tail -4 )
python tools/scale_probe.py 125000 512 16 1 2>&1 | grep "fetch_un\|score_generic" | cut -c1-1500
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-scaling-workload --extra c5k16 > gpurun_out/r5_bench_c.json 2> gpurun_out/r5_bench_c.err; echo "bench rc=$?"
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_c.json').read().strip().splitlines()[-1])
c5=d['other_workloads']['ital_k16_mc1_1Mx512']
print('C5 whole', c5['ms_per_round'], c5['fetch_s'], c5['roofline']['frac'], sum(c5['step_ms'].values()))
PY
( FUZZ_KINDS=mc,noisy timeout 1500 python tools/fuzz_parity.py 2000 211 > gpurun_out/r5_fuzz_mc_noisy_seed211.log 2>&1; tail -2 gpurun_out/r5_fuzz_mc_noisy_seed211.log )
( FUZZ_KINDS=mcwide timeout 900 python tools/fuzz_parity.py 250 223 > gpurun_out/r5_fuzz_mcwide_seed223.log 2>&1; tail -2 gpurun_out/r5_fuzz_mcwide_seed223.log )
