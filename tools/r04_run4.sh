#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
for V in 2 1; do
  echo "== LEAF_QKVATTN_V=$V parity"
  LEAF_QKVATTN_V=$V timeout -k 10 600 python -m pytest tests/test_gpu_fused_attn.py -x -q 2>&1 | tail -15 || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_forward.py -x -q -k "prefix or packed or trim or random_shapes or two_stream or token_identical" 2>&1 | tail -5 || exit 1
for V in 1 2; do LEAF_QKVATTN_V=$V timeout -k 10 300 python tools/qkv_attn_bench.py || exit 1; done
LEAF_QKVATTN_V=2 LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_stamps.so timeout -k 10 300 python tools/qkv_attn_bench.py || exit 1
line() {
  python -c "
import sys,json
d=json.loads(sys.stdin.read())
sh={(s['kernel'],s['N'],s['K']):s for s in d['roofline']['shapes']}
def g(k):
    s=sh.get(k); return '%.2f'%s['ms_per_step'] if s else '-'
print('$1', '%.2f ms'%d['ms_per_step'], 'fc',g(('gemm_nt256_half_kernel<F16,6>',3072,768)),'qkv',g(('gemm_nt256_half_kernel<F16,5>',2304,768)),'fused',g(('qkv_attn_kernel<F16,5>',2304,768)),'cproj',g(('gemm_nt256_half_kernel<F16,7>',768,3072)),'out',g(('gemm_nt256_half_kernel<F16,7>',768,768)), flush=True)"
}
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2; do
  LEAF_FUSE_ATTN=0 timeout -k 10 200 $B 2>/dev/null | tail -1 | line "two kernels" || exit 1
  for V in 1 2; do
    LEAF_QKVATTN_V=$V timeout -k 10 200 $B 2>/dev/null | tail -1 | line "fused v$V" || exit 1
  done
done
