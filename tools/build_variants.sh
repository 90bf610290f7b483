#!/bin/bash
# Builds kernel variants of libital_hip.so for A/B timing: each argument is  name=extra hipcc flags
#   tools/build_variants.sh "base=" "ilp=-mllvm -amdgpu-sched-strategy=max-ilp" "nh3=-DITAL_QMC_NH=3"
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ITAL_HIPCC_EXTRA="$flags" ITAL_OBJ_DIR=build_variants/obj_$name ITAL_HIP_LIB_OUT=$PWD/build_variants/libital_$name.so python -m ital_amd.build > build_variants/build_$name.log 2>&1 &
done
wait
ls -la build_variants/*.so
