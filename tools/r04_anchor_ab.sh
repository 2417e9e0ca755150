#!/bin/bash
# Where on the device's time line the frozen model's anchor forward runs (LEAF_ANCHOR_AT = free | tail | gap), same box, alternating
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
B="python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-dense-leg"
for r in 1 2 3; do
  for w in free tail; do
    LEAF_ANCHOR_AT=$w timeout -k 10 200 $B 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('LEAF_ANCHOR_AT=$w  %.2f ms/step  %.1f samples/s' % (d['ms_per_step'], d['value']), flush=True)" || exit 1
  done
done
