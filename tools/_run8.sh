set -u
export PYTHONHASHSEED=0
for v in cur old6 nopin nopu; do
  echo "== $v"
  FUZZ_KINDS=mcwide ITAL_HIP_LIB=$PWD/build_variants/libital_$v.so timeout 600 python tools/fuzz_parity.py 150 201 133 2>&1 | grep -v amdgpu.ids | tail -25 | cut -c1-400
done
