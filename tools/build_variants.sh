#!/bin/bash
# Builds kernel variants of libital_hip.so for A/B timing: each argument is  name=extra hipcc flags
#   tools/build_variants.sh "base=" "ilp=-mllvm -amdgpu-sched-strategy=max-ilp" "nh3=-DITAL_QMC_NH=3"
# VARIANT_TU=score (default: all): only that translation unit is recompiled with the flags, the other objects are
# copied from the in-tree build (ital_amd/_obj must be up to date).
set -e
cd "$(dirname "$0")/.."
mkdir -p build_variants
python -m ital_amd.build > /dev/null
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  if [ -n "${VARIANT_TU:-}" ]; then
    mkdir -p build_variants/obj_$name
    cp -p ital_amd/_obj/*.o build_variants/obj_$name/
    for tu in $VARIANT_TU; do rm -f build_variants/obj_$name/$tu.o; done
    rm -f build_variants/libital_$name.so
  fi
  ITAL_HIPCC_EXTRA="$flags" ITAL_OBJ_DIR=build_variants/obj_$name ITAL_HIP_LIB_OUT=$PWD/build_variants/libital_$name.so python -m ital_amd.build > build_variants/build_$name.log 2>&1 &
done
wait
ls -la build_variants/*.so
