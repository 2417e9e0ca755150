#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_dp2.py tests/test_gpu_cli.py -x -q 2>&1 | tail -25 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || exit 1
tail -c 1500 gpurun_out/r04_bench_default.json
