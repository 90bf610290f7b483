#!/bin/bash
# every library variant under build_variants/ on the noisy-user workload and on a Monte-Carlo pattern round (k = 16)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r3
export ITAL_BENCH_NO_EXTRAS=1
echo "--- noisy user (9298 x 256, k = 4, label_prob 0.5, mistake_prob 0.25)"
for lib in build_variants/libital_*.so; do
  ITAL_HIP_LIB=$PWD/$lib python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-scaling-workload --label-prob 0.5 --mistake-prob 0.25 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$(basename $lib .so)', 'ms/step %.3f' % j['ms_per_step'])"
done
echo "--- 40 000 x 512, k = 16, monte_carlo_num_rel = 1"
tools/variant_probe.sh 40000 512 16 1 | cut -c1-1300
