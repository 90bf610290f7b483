#!/bin/bash
# every library variant under build_variants/ on the headline workload and on 25 000 x 256, k = 8 (lattice-sum times)
cd "$(dirname "$0")/.."; mkdir -p gpurun_out/r3
export ITAL_BENCH_NO_EXTRAS=1
echo "--- headline (9298 x 256, k = 4)"; STEPS=10 BENCH_ARGS="--no-scaling-workload" tools/variant_bench.sh
echo "--- 25 000 x 256, k = 8";          STEPS=2  BENCH_ARGS="--rows 25000 --batch 8 --no-scaling-workload" tools/variant_bench.sh
