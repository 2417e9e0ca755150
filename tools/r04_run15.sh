#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || exit 1
