set -u
mkdir -p gpurun_out
export PYTHONHASHSEED=0
python tools/_diag_ce.py 2>&1 | grep -v amdgpu | cut -c1-300
( FUZZ_KINDS=subset,mix timeout 300 python tools/fuzz_parity.py 250 233 62 2>&1 | tail -3 | cut -c1-400 )
( timeout 1700 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gputests_final.log; tail -4 gpurun_out/r5_gputests_final.log )
python tools/cesub_bench.py 2>&1 | grep "^{" | cut -c1-200
( FUZZ_KINDS=subset,mix timeout 700 python tools/fuzz_parity.py 300 233 > gpurun_out/r5_fuzz_subset_mix_seed233.log 2>&1; tail -1 gpurun_out/r5_fuzz_subset_mix_seed233.log )
