#!/bin/bash
# Round-4 first GPU call: (1) tile-order A/B of the 256^2 kernel (LEAF_GEMM_MPANEL), (2) upper bound of a fused QKV-GEMM -> attention
# kernel (diagnostic builds, garbage results, fixed row plan), (3) which kernel should take the 3.2 k-row launches.
# Run on the MI355X box from the repo root: tools/r04_run1.sh > gpurun_out/r04_run1.txt
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
line() {
  python -c "
import sys,json
d=json.loads(sys.stdin.read())
sh={(s['kernel'],s['N'],s['K']):s for s in d['roofline']['shapes']}
def g(k):
    s=sh.get(k); return '%.2f'%s['ms_per_step'] if s else '-'
print('$1', '%.2f ms'%d['ms_per_step'], 'fc',g(('gemm_nt256_half_kernel<F16,6>',3072,768)),'qkv',g(('gemm_nt256_half_kernel<F16,5>',2304,768)),'cproj',g(('gemm_nt256_half_kernel<F16,7>',768,3072)),'out',g(('gemm_nt256_half_kernel<F16,7>',768,768)), flush=True)"
}
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
echo "== (1) tile order: LEAF_GEMM_MPANEL (0 = N groups, the shipped order), lr as shipped"
for r in 1 2; do
  for P in 0 4 8 6 12; do
    LEAF_GEMM_MPANEL=$P timeout -k 10 200 $B 2>/dev/null | tail -1 | line "mpanel=$P" || exit 1
  done
done
echo "== (2) fusion bound (LEAF_DIAG_FIXED_WINNER=1: same row plan in every arm)"
export LEAF_DIAG_FIXED_WINNER=1
for r in 1 2; do
  for V in shipped noqkvstore attnl2 fusebound; do
    if [ $V = shipped ]; then unset LEAF_HIP_LIB; else export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_$V.so; fi
    timeout -k 10 200 $B 2>/dev/null | tail -1 | line "$V" || exit 1
  done
done
unset LEAF_HIP_LIB LEAF_DIAG_FIXED_WINNER
echo "== (3) 3.2 k-row launches: 256^2 ring vs 64x128 ring (columns), then the same with the 128x256 ping-pong kernel forced (both columns)"
MS=3219,6400 timeout -k 10 200 python tools/small_gemm_sweep.py || exit 1
LEAF_GEMM_PP=1 LEAF_GEMM_PP_MIN_TILES=1 MS=3219,6400 timeout -k 10 200 python tools/small_gemm_sweep.py || exit 1
echo "== (4) phases"
timeout -k 10 200 python tools/phase_bench.py || exit 1
