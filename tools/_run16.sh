set -u
export PYTHONHASHSEED=0
echo "== current"
( FUZZ_KINDS=subset,mix timeout 300 python tools/fuzz_parity.py 250 233 62 2>&1 | grep -v amdgpu | tail -60 | cut -c1-250 )
echo "== nosub (single kernel for subsets)"
( FUZZ_KINDS=subset,mix ITAL_HIP_LIB=$PWD/build_variants/libital_nosub.so timeout 300 python tools/fuzz_parity.py 250 233 62 2>&1 | grep -v amdgpu | tail -8 | cut -c1-250 )
