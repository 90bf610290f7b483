set -u
mkdir -p gpurun_out
export PYTHONHASHSEED=0
for i in 1 2 3; do ( timeout 600 python -m pytest "tests/test_gpu_parity.py::test_golden_fixture_general_scorer" -x -q 2>&1 | tail -2 ); done
( timeout 600 python -m pytest tests/test_gpu_ctx_api.py tests/test_gpu_exchange_timeout.py tests/test_gpu_host_cpp.py "tests/test_gpu_parity.py::test_golden_fixture_general_scorer" -x -q 2>&1 | tail -3 )
( ITAL_TEST_SKIP_C5_FULL=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | cut -c1-300 )
python tools/cesub_bench.py 2>&1 | grep "^{" | cut -c1-500
( FUZZ_KINDS=subset,mix timeout 900 python tools/fuzz_parity.py 400 229 > gpurun_out/r5_fuzz_subset_mix_seed229.log 2>&1; tail -1 gpurun_out/r5_fuzz_subset_mix_seed229.log; grep -c "^case" gpurun_out/r5_fuzz_subset_mix_seed229.log )
