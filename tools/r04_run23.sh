#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_forward.py tests/test_gpu_fused_attn.py -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 200 python tools/stage_hole_probe.py 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
timeout -k 10 200 python tools/stage_hole_probe.py ViT-H-14 32 2>&1 | grep -v amdgpu.ids | tail -4 || exit 1
for r in 1 2 3; do timeout -k 10 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-dense-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))"; done
timeout -k 10 300 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-dense-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('config 3:', round(d['value'],1), round(d['ms_per_step'],2))"
