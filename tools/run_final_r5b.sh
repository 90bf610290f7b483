# Closing run of a change to the general scorer's sources: build + smoke, the GPU suite, the profiles of the workloads whose
# kernels it touches (general, c5, cesub), then the bench line with the counters of those fresh profiles.
set -u
ROOT=$(pwd)
mkdir -p gpurun_out
rm -rf gpurun_out/prof_r5
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2
( timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r5_gputests_final.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r5_gputests_final.log; tail -3 gpurun_out/r5_gputests_final.log )
( timeout 900 bash tools/profile_r5.sh general c5 cesub > gpurun_out/r5_profile_b.log 2>&1; echo "profile rc=$?" )
cd $ROOT
python tools/merge_profiles.py
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 3 > gpurun_out/r5_bench_final.json 2> gpurun_out/r5_bench_final.err; echo "bench rc=$?"
