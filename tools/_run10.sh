set -u
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_gpu_ctx_api.py tests/test_gpu_host_cpp.py tests/test_gpu_exchange_timeout.py tests/test_gpu_parity_limits.py -x -q 2>&1 | tail -15 )
