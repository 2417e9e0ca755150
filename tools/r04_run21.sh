#!/bin/bash
# N-group size of the 256^2 GEMM's tile order (LEAF_GEMM_NGROUP) inside the step, now that the QKV launch is fused: c_fc / c_proj / out_proj time per step
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2; do
  for G in default 2 3 4 6 12; do
    if [ $G = default ]; then unset LEAF_GEMM_NGROUP; else export LEAF_GEMM_NGROUP=$G; fi
    timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
f={(s['kernel'][:32],s['N'],s['K']):s for s in d['roofline']['shapes'] if s['big_launches']}
g=lambda k,n,kk: f[(k,n,kk)]['ms_per_step']
print('NGROUP=$G  %.2f ms/step' % d['ms_per_step'], ' c_proj %.2f  out_proj %.2f  c_fc %.2f  fused %.2f' % (g('gemm_nt256_half_kernel<F16,7>',768,3072), g('gemm_nt256_half_kernel<F16,7>',768,768), g('gemm_nt256_half_kernel<F16,6>',3072,768), g('qkv_attn_kernel<F16,5>',2304,768)), flush=True)" || exit 1
  done
done
