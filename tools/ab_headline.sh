#!/bin/bash
cd "$(dirname "$0")/.."
export ITAL_BENCH_NO_EXTRAS=1
for rep in 1 2; do
echo "--- headline (9298 x 256, k = 4)"; STEPS=20 BENCH_ARGS="--no-scaling-workload" tools/variant_bench.sh
done
