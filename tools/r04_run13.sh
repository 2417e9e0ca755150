#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 || exit 1
bash tools/r04_traffic_configs.sh 2>&1 | tail -12 || exit 1
