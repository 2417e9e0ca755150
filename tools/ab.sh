#!/bin/bash
# Same-box A/B on the MI355X box: alternates bench.py runs over ARMS, each arm = LABEL 'VAR=VAL VAR=VAL ...' (LEAF_HIP_LIB=path selects
# another build of the library; an empty string = the shipped configuration), prints one line per run (tools/bench_line.py).
#   tools/ab.sh [-r ROUNDS] [-s STEPS] [-w WARMUP] [-x 'extra bench args'] [-t 'pytest args'] [-p 'command run once before'] [-S] \
#               LABEL_A 'ENV_A' LABEL_B 'ENV_B' [LABEL_C 'ENV_C' ...]
# e.g. tools/ab.sh -r 3 two-kernels 'LEAF_FUSE_ATTN=0' fused ''          tools/ab.sh prev "LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_prev.so" new ''
# -S prints the B-caption (small-launch) shapes instead of the scoring passes'.  Every GPU step runs under its own `timeout -k 10`;
# a failed step ends the script (no GPU step is started after one that timed out).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$PWD}" && mkdir -p gpurun_out
R=2; STEPS=40; WARM=8; EXTRA=""; PYT=""; PRE=""; SMALL=""
while getopts "r:s:w:x:t:p:S" o; do
  case $o in r) R=$OPTARG;; s) STEPS=$OPTARG;; w) WARM=$OPTARG;; x) EXTRA=$OPTARG;; t) PYT=$OPTARG;; p) PRE=$OPTARG;; S) SMALL="--small";; *) exit 2;; esac
done
shift $((OPTIND - 1))
if [ -n "$PYT" ]; then timeout -k 10 900 python -m pytest $PYT -x -q 2>&1 | tail -5 || exit 1; fi
if [ -n "$PRE" ]; then timeout -k 10 600 bash -c "$PRE" || exit 1; fi
for i in $(seq $R); do
  set -- "$@"
  args=("$@")
  for ((a = 0; a < ${#args[@]}; a += 2)); do
    label=${args[a]}; envs=${args[a + 1]}
    # shellcheck disable=SC2086
    env $envs timeout -k 10 300 python bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline --no-dense-leg $EXTRA 2>/dev/null | tail -1 \
      | python tools/bench_line.py "$label" $SMALL || exit 1
  done
done
