set -u
mkdir -p gpurun_out
export PYTHONHASHSEED=0
( timeout 1500 python -m pytest tests/test_gpu_scale.py::test_c5_share_125000x512_k16_monte_carlo tests/test_gpu_parity.py tests/test_gpu_multirank_scale.py tests/test_gpu_parity_limits.py -x -q 2>&1 | tail -6 )
python tools/scale_probe.py 20000 64 16 1 2>&1 | grep "fetch_un\|score_generic" | cut -c1-1200
python tools/scale_probe.py 125000 512 16 1 2>&1 | grep "fetch_un\|score_generic" | cut -c1-1500
( FUZZ_KINDS=mcwide timeout 1200 python tools/fuzz_parity.py 150 201 > gpurun_out/r5_fuzz_mcwide_seed201.log 2>&1; tail -3 gpurun_out/r5_fuzz_mcwide_seed201.log )
