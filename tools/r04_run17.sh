#!/bin/bash
# fp32 output of the residual GEMMs: non-temporal stores (shipped) against regular write-back stores (-DLEAF_F32_REGULAR)
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
for L in stamps f32reg_stamps; do
  echo "== epilogue stamps: $L"
  LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_$L.so timeout -k 10 300 python tools/epilogue_burst_probe.py 2>&1 | grep -E "^out|^proj" || exit 1
done
for L in shipped f32reg; do
  echo "== stand-alone: $L"
  if [ $L = shipped ]; then unset LEAF_HIP_LIB; else export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_$L.so; fi
  timeout -k 10 300 python tools/lib_gemm_ref.py 2>&1 | grep -E "out |proj |layer" || exit 1
done
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2 3; do
  for L in shipped f32reg; do
    if [ $L = shipped ]; then unset LEAF_HIP_LIB; else export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_$L.so; fi
    timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
f={(s['kernel'][:32],s['N'],s['K']):s for s in d['roofline']['shapes'] if s['big_launches']}
g=lambda k,n,kk: f[(k,n,kk)]['ms_per_step']
print('$L  %.2f ms/step' % d['ms_per_step'], ' c_proj %.2f  out_proj %.2f  c_fc %.2f  fused %.2f' % (g('gemm_nt256_half_kernel<F16,7>',768,3072), g('gemm_nt256_half_kernel<F16,7>',768,768), g('gemm_nt256_half_kernel<F16,6>',3072,768), g('qkv_attn_kernel<F16,5>',2304,768)), flush=True)" || exit 1
  done
done
