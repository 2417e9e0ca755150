#!/bin/bash
# out_proj alone on the round-3 ping-pong kernel (diagnostic build, LEAF_GEMM_PP=2) against the shipped dispatch, same box
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_variants.so
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2 3; do
  for pp in 0 2; do
    LEAF_GEMM_PP=$pp timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
out=[s for s in d['roofline']['shapes'] if s['N']==768 and s['K']==768 and s.get('big_launches')]
print('LEAF_GEMM_PP=$pp  %.2f ms/step' % d['ms_per_step'], ' out_proj:', ['%s %.2f ms' % (s['kernel'], s['ms_per_step']) for s in out], flush=True)" || exit 1
  done
done
