#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer build of the HOST side of libital_hip.so (argument validation, descriptor
# handling, the stream bookkeeping and numpy-stream walker, RCCL lookup) and a driver that exercises it on the CPU: no
# GPU needed (GPU ASan is not available on this pool; device code is compiled as usual and never launched by the driver).
#   tools/asan_host.sh            builds build_asan/libital_hip_asan.so + build_asan/host_driver, runs the driver
set -e
cd "$(dirname "$0")/.."
mkdir -p build_asan
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -Wno-option-ignored"
ITAL_HIPCC_EXTRA="$SAN" ITAL_HOST_EXTRA="$SAN" ITAL_LINK_EXTRA="$SAN" ITAL_OBJ_DIR=$PWD/build_asan/obj \
  ITAL_HIP_LIB_OUT=$PWD/build_asan/libital_hip_asan.so python -m ital_amd.build > build_asan/build.log 2>&1 || { tail -20 build_asan/build.log; exit 1; }
if [ ! -e build_asan/host_driver ] || [ tests/host_asan_driver.cpp -nt build_asan/host_driver ] || [ build_asan/libital_hip_asan.so -nt build_asan/host_driver ]; then
  /opt/rocm/bin/hipcc $SAN -std=c++17 -I include tests/host_asan_driver.cpp -o build_asan/host_driver -L build_asan -lital_hip_asan -Wl,-rpath,$PWD/build_asan -lpthread
fi
ASAN_OPTIONS=detect_leaks=0:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 build_asan/host_driver
