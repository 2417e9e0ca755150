#!/bin/bash
# The profile set of a round, on the MI355X box from the repo root:  tools/profile_round.sh r05 [part ...]
# parts (default: bench stats pmc): bench = default + driver-form lines; stats = rocprofv3 --kernel-trace --stats of a short run;
# pmc = the five PMC passes + traffic / utilisation summaries; configs = bench lines + traffic of BASELINE.json configs[2..4];
# suite = the whole GPU test suite + smoke first.  Outputs under gpurun_out/<round>/, named as they go under profiles/.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-$PWD}"
RD=${1:?usage: tools/profile_round.sh rNN [suite] [bench] [stats] [pmc] [configs]}; shift
PARTS=${*:-bench stats pmc}
O=gpurun_out/$RD; mkdir -p $O
export TMPDIR=/tmp
for part in $PARTS; do case $part in
suite)
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
  python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1;;
bench)
  timeout -k 10 500 python bench.py > $O/${RD}_bench_default.json 2> $O/bench_default.err || exit 1
  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/${RD}_bench_20steps.json 2> $O/bench_20steps.err || exit 1
  for f in $O/${RD}_bench_default.json $O/${RD}_bench_20steps.json; do tail -1 $f | python tools/bench_line.py $(basename $f); done;;
stats)
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dense-leg > $O/stats.log 2>&1 || exit 1
  cp $(ls $O/stats/*/run_kernel_stats.csv $O/stats/run_kernel_stats.csv 2>/dev/null | head -1) $O/${RD}_kernel_stats.csv || exit 1;;
pmc)
  timeout -k 10 900 tools/pmc_passes.sh $O/pmc || exit 1
  python tools/pmc_summary.py $O/pmc/fetch $O/pmc/write $O/$RD "" $O/pmc/fetch.bench.json || exit 1
  python tools/pmc_util.py $O/${RD}_mfma_util.json $O/pmc/sq1 $O/pmc/sq2 $O/pmc/tcc > $O/${RD}_mfma_util.txt || exit 1;;
configs)
  for c in 2 3 4; do
    st=20; wu=5; [ $c = 4 ] && { st=8; wu=2; }
    timeout -k 10 600 python bench.py --config $c --steps $st --warmup $wu --no-cpu-baseline > $O/${RD}_bench_cfg$c.json 2>> $O/cfg.err || exit 1
    tail -1 $O/${RD}_bench_cfg$c.json | python tools/bench_line.py config$c
    timeout -k 10 500 tools/pmc_passes.sh $O/cfg$c --config $c || exit 1
    python tools/pmc_summary.py $O/cfg$c/fetch $O/cfg$c/write $O/${RD}_cfg$c "" $O/cfg$c/fetch.bench.json || exit 1
    mv $O/${RD}_cfg${c}_traffic.json $O/${RD}_traffic_cfg$c.json
  done;;
*) echo "unknown part $part"; exit 2;;
esac; done
echo "profile_round $RD: $PARTS done"
