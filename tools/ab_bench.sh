#!/bin/bash
# Same-box A/B of two builds of libleaf_hip.so: alternates `bench.py` runs with LEAF_HIP_LIB = $1 (A) and the in-tree library (B).
# usage: tools/ab_bench.sh tools/diag/libleaf_hip_prev.so [rounds] [extra bench args...]
A=$1; R=${2:-2}; shift; shift
for i in $(seq $R); do
  for which in A B; do
    if [ $which = A ]; then export LEAF_HIP_LIB=$PWD/$A; else unset LEAF_HIP_LIB; fi
    python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-dense-leg "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
sh={(s['kernel'],s['N'],s['K']):s for s in d['roofline']['shapes']}
def g(k): 
    s=sh.get(k); return '%.2f'%s['ms_per_step'] if s else '-'
print('$which', '%.2f ms'%d['ms_per_step'], 'fc',g(('gemm_nt256_half_kernel<F16,6>',3072,768)),'qkv',g(('gemm_nt256_half_kernel<F16,5>',2304,768)),'cproj',g(('gemm_nt256_half_kernel<F16,7>',768,3072)),'out',g(('gemm_nt256_half_kernel<F16,7>',768,768)), flush=True)"
  done
done
