#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE passes) of BASELINE.json configs[2..4] through bench.py --config N: profiles/r04_traffic_cfgN.json,
# which bench.py attaches to lines of the same workload and kernel sources.  Run on the MI355X box from the repo root.
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out/r04 && export TMPDIR=/tmp PMC_TRAFFIC_ONLY=1
for c in 2 3 4; do
  O=gpurun_out/r04/cfg$c
  timeout -k 10 500 tools/pmc_passes.sh $O --config $c || exit 1
  python tools/pmc_summary.py $O/fetch $O/write gpurun_out/r04/r04_cfg$c "" $O/fetch.bench.json || exit 1
  mv gpurun_out/r04/r04_cfg${c}_traffic.json gpurun_out/r04/r04_traffic_cfg$c.json
done
