set -u
mkdir -p gpurun_out
summ() { python3 -c "
import sys,re,ast
name=None
for line in sys.stdin:
    line=line.strip()
    if line.startswith('=='): name=line[3:]
    elif line.startswith('fetch'): print(name, line.split(';')[0][:70], 'picks_hash', hash(line.split('picks')[1])%100000)
    elif line.startswith('{'):
        d=ast.literal_eval(line)
        print('   ', ' '.join('%s=%.2f'%(k.replace('score_generic_','g').replace('qmc_main_','q').replace('qmc_slabs','qs'),v) for k,v in d.items() if not k.startswith('cross') and not k.startswith('score_t')))
"; }
probe() { pat=$1; shift
  for lib in build_variants/libital_$pat.so; do
    echo "== $(basename $lib .so)"
    ITAL_HIP_LIB=$PWD/$lib python tools/scale_probe.py "$@" 2>&1 | grep "fetch_un\|score_generic\|qmc_"
  done
}
export PYTHONHASHSEED=0
( timeout 900 python -m pytest tests/test_gpu_exchange_timeout.py tests/test_gpu_host_cpp.py tests/test_gpu_cabi_only.py tests/test_gpu_scale.py::test_c5_share_125000x512_k16_monte_carlo -x -q 2>&1 | tail -15 )
for i in 1 2; do
  probe '*' 20000 64 16 1 | summ
done
python tools/scale_probe.py 125000 512 16 1 2>&1 | grep "fetch_un\|score_generic" | cut -c1-1500
ITAL_HIP_LIB=$PWD/build_variants/libital_inl.so python tools/scale_probe.py 125000 512 16 1 2>&1 | grep "fetch_un\|score_generic" | cut -c1-1500
