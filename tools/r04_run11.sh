#!/bin/bash
# head-split XCD order of the fused launch on the other towers (ViT-H: 16 heads, bigG: 20 heads): fused-kernel time per step under 1 / 2 / 4 head groups
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_fused_attn.py -x -q 2>&1 | tail -3 || exit 1
for c in "--config 3 --steps 12 --warmup 3" "--config 4 --steps 6 --warmup 2"; do
  for H in 1 2 4 1 2 4; do
    LEAF_QKVATTN_HSPLIT=$H timeout -k 10 500 python bench.py $c --no-cpu-baseline --no-dense-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
f=[s for s in d['roofline']['shapes'] if s['kernel'].startswith('qkv_attn')]
print('bench.py $c  LEAF_QKVATTN_HSPLIT=$H  %.2f ms/step' % d['ms_per_step'], ' fused: %.2f ms %.0f TF/s' % (f[0]['ms_per_step'], f[0]['tflops']) if f else '', flush=True)" || exit 1
  done
done
