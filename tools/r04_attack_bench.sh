#!/bin/bash
# the real string search (attack_text) at the round-4 build: letters-only and punctuated / two-sentence captions, with and without --constrain,
# fused QKV+attention on (default) and off
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
for F in 1 0; do
  echo "== LEAF_FUSE_ATTN=$F"
  LEAF_FUSE_ATTN=$F timeout -k 10 300 python tools/attack_bench.py --constrain 2>&1 | grep -E "^native" || exit 1
  LEAF_FUSE_ATTN=$F timeout -k 10 300 python tools/attack_bench.py --constrain --tokenizer treebank --punct --sentences 0.15 --punkt native 2>&1 | grep -E "^native" || exit 1
done
