#!/bin/bash
# beyond-L2 fetches and L2 hit rate of the fused launch under the head-split XCD orders (separate PMC passes)
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out/hs && export TMPDIR=/tmp
for H in 1 2 4; do
  export LEAF_QKVATTN_HSPLIT=$H
  for pass in "fetch FETCH_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum"; do
    set -- $pass; name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/hs/p$H/$name -o run -- python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense-leg > gpurun_out/hs/p${H}_$name.log 2>&1 || exit 1
  done
  python tools/hs_counters.py gpurun_out/hs/p$H $H || exit 1
done
