#!/bin/bash
# What the data-parallel machinery costs with ONE rank through RCCL on one GPU (no transfer happens): bench.py without a process
# group, with the flat all-reduce (LEAF_DP_OVERLAP=0) and with the per-bucket collectives behind the backward, then a kernel trace
# of the last form.  usage (on the GPU box): bash tools/dp_probe.sh   -> gpurun_out/dp_*.json, gpurun_out/dp_prof/
set -e
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
for mode in nodist flat overlap; do
  case $mode in
    nodist) export LEAF_BENCH_FORCE_DIST=0; unset LEAF_DP_OVERLAP;;
    flat) export LEAF_BENCH_FORCE_DIST=1; export LEAF_DP_OVERLAP=0;;
    overlap) export LEAF_BENCH_FORCE_DIST=1; export LEAF_DP_OVERLAP=1;;
  esac
  python bench.py --steps 30 --warmup 5 > gpurun_out/dp_$mode.json 2> gpurun_out/dp_$mode.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/dp_$mode.json").read().strip().splitlines()[-1])
print("$mode", round(d["ms_per_step"],2), d.get("per_rank",{}).get("exposed_collective_ms_per_step"))
PY
done
export LEAF_BENCH_FORCE_DIST=1; export LEAF_DP_OVERLAP=1
rocprofv3 --kernel-trace --stats -d gpurun_out/dp_prof -o run -- python bench.py --steps 4 --warmup 2 > gpurun_out/dp_prof.json 2> gpurun_out/dp_prof.err
ls gpurun_out/dp_prof | head
