#!/bin/bash
# residual-GEMM epilogue with two passes of residual loads in flight: parity, epilogue stamps, stand-alone time, step A/B against the previous build
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gemm" 2>&1 | tail -3 || exit 1
echo "== epilogue stamps (new build)"
LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_stamps.so timeout -k 10 300 python tools/epilogue_burst_probe.py 2>&1 | grep -E "^out|^proj|^qkv" || exit 1
echo "== stand-alone, previous build"
LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_prev.so timeout -k 10 300 python tools/lib_gemm_ref.py 2>&1 | grep -E "out |proj |layer" || exit 1
echo "== stand-alone, new build"
timeout -k 10 300 python tools/lib_gemm_ref.py 2>&1 | grep -E "out |proj |layer" || exit 1
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2 3; do
  for L in prev new; do
    if [ $L = prev ]; then export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_prev.so; else unset LEAF_HIP_LIB; fi
    timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
f={(s['N'],s['K']):s for s in d['roofline']['shapes'] if s['kernel'].startswith('gemm_nt256_half_kernel<F16,7>') and s['big_launches']}
print('$L  %.2f ms/step' % d['ms_per_step'], ' c_proj %.2f ms %.0f TF/s  out_proj %.2f ms %.0f TF/s' % (f[(768,3072)]['ms_per_step'], f[(768,3072)]['tflops'], f[(768,768)]['ms_per_step'], f[(768,768)]['tflops']), flush=True)" || exit 1
  done
done
