#!/bin/bash
# BASELINE.json configs[2..4] on ONE GPU through the presets the scaling run will use (bench.py --config), short runs from a cold model.
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
for c in "--config 2 --steps 12 --warmup 3" "--config 3 --steps 12 --warmup 3" "--config 4 --steps 8 --warmup 2" "--model ViT-L-14 --steps 20 --warmup 5"; do
  echo "== bench.py $c"
  timeout -k 10 500 python bench.py $c --no-cpu-baseline --no-dense-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
print('  %.1f samples/s  %.1f ms/step  dominant %s frac %.3f  (%s)' % (d['value'], d['ms_per_step'], r['kernel'], r['frac'], d['config']['workload'][:90]))
for s in r['shapes'][:5]: print('     ', s['kernel'], s['N'], s['K'], '%.2f ms/step %.0f TF/s' % (s['ms_per_step'], s['tflops']))" || exit 1
done
