#!/bin/bash
# head-split XCD order of the fused QKV+attention launch (LEAF_QKVATTN_HSPLIT): parity, stand-alone time, step time, beyond-L2 fetches
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out/hs && export TMPDIR=/tmp
LEAF_QKVATTN_HSPLIT=2 timeout -k 10 600 python -m pytest tests/test_gpu_fused_attn.py -x -q 2>&1 | tail -5 || exit 1
for H in 1 2 4; do
  echo "== LEAF_QKVATTN_HSPLIT=$H"
  LEAF_QKVATTN_HSPLIT=$H timeout -k 10 300 python tools/qkv_attn_bench.py 2>/dev/null || exit 1
done
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2 3; do
  for H in 1 2 4; do
    LEAF_QKVATTN_HSPLIT=$H timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
f=[s for s in d['roofline']['shapes'] if s['kernel'].startswith('qkv_attn')]
print('LEAF_QKVATTN_HSPLIT=$H  %.2f ms/step' % d['ms_per_step'], ' fused: %.2f ms' % f[0]['ms_per_step'] if f else '', flush=True)" || exit 1
  done
done
for H in 1 2 4; do
  export LEAF_QKVATTN_HSPLIT=$H
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d gpurun_out/hs/p$H -o run -- python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense-leg > gpurun_out/hs/p$H.log 2>&1 || exit 1
  python tools/hs_counters.py gpurun_out/hs/p$H $H || exit 1
done
