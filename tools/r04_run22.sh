#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_step.py tests/test_gpu_fused_attn.py tests/test_gpu_fullshape.py -x -q 2>&1 | tail -3 || exit 1
for r in 1 2 3; do timeout -k 10 200 python bench.py --steps 60 --warmup 8 --no-cpu-baseline --no-dense-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); s=[x for x in d['roofline']['shapes'] if x['N'] in (1536,2304) and not x['big_launches']]
print(round(d['ms_per_step'],2), [(x['kernel'],x['N'],round(x['ms_per_step'],3),x['launches']) for x in s])"; done
