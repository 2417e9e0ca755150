#!/bin/bash
# A/B of the ping-pong GEMM (gemm128pp.hip, LEAF_GEMM_PP=1) against the 256 x 256 half-stage kernel on one box:
# the four GEMM shapes of a ViT-L block alone (tools/gemm_bench.py), then alternating bench.py runs.
export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_variants.so   # the ping-pong kernel lives in the diagnostic build only (make -C leaf_amd/csrc variants)
for pp in 0 1 0 1; do echo "== gemm_bench LEAF_GEMM_PP=$pp"; LEAF_GEMM_PP=$pp SEQS=${SEQS:-1200} timeout -k 10 120 python tools/gemm_bench.py || exit 1; done
for pp in 0 1 0 1; do
  echo "== bench.py LEAF_GEMM_PP=$pp"
  LEAF_GEMM_PP=$pp timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-dense-leg 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('%.2f ms/step  %.0f samples/s  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))
for s in d['roofline']['shapes'][:5]: print('   ', s['kernel'], s['N'], s['K'], '%.2f ms/step %.0f TF/s' % (s['ms_per_step'], s['tflops']))" || exit 1
done
