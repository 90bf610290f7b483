#!/bin/bash
# Times the greedy-step kernels of bench.py for every library variant under build_variants/ (one JSON line each).
cd "$(dirname "$0")/.."
for lib in build_variants/libital_*.so; do
  v=$(basename $lib .so)
  ITAL_HIP_LIB=$PWD/$lib python bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('$v', 'ms/step %.3f' % j['ms_per_step'], {k: round(v,3) for k,v in j['kernel_ms'].items() if k[:5] in ('score', 'qmc_m', 'cross')})"
done
