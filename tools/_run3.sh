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
probe() { # pattern, args
  pat=$1; shift
  for lib in build_variants/libital_$pat.so; do
    echo "== $(basename $lib .so)"
    ITAL_HIP_LIB=$PWD/$lib python tools/scale_probe.py "$@" 2>&1 | grep "fetch_un\|score_generic\|qmc_"
  done
}
export PYTHONHASHSEED=0
for i in 1 2; do
  probe '[a-e]' 20000 64 16 1 | summ
  probe 'pin' 20000 64 16 1 | summ
  probe 's*' 25000 512 8 | summ
done
