#!/bin/bash
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 300 python tools/qkv_attn_bench.py || exit 1
LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_stamps.so timeout -k 10 300 python tools/qkv_attn_bench.py || exit 1
