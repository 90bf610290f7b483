cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r3
export ITAL_BENCH_NO_EXTRAS=1 ITAL_BENCH_STEP_TIMES=1
A="--steps 20 --warmup 5 --no-cpu-baseline --no-scaling-workload"
python bench.py $A 2> gpurun_out/r3/ab1.err | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('events   ', j['ms_per_step'], j['ms_per_step_unfrozen_heap'])"
ITAL_BENCH_NO_EVENTS=1 python bench.py $A 2> gpurun_out/r3/ab2.err | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('no events', j['ms_per_step'], j['ms_per_step_unfrozen_heap'])"
tail -2 gpurun_out/r3/ab1.err; tail -2 gpurun_out/r3/ab2.err
