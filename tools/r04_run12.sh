#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_fused_attn.py -x -q 2>&1 | tail -5 || exit 1
timeout -k 10 300 python tools/qkv_attn_bench.py 2>/dev/null || exit 1
LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_stamps.so timeout -k 10 300 python tools/qkv_attn_bench.py 2>/dev/null || exit 1
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2 3; do
    timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
f=[s for s in d['roofline']['shapes'] if s['kernel'].startswith('qkv_attn')]
print('%.2f ms/step' % d['ms_per_step'], ' fused: %.2f ms' % f[0]['ms_per_step'] if f else '', flush=True)" || exit 1
done
