#!/bin/bash
# Round-4 profiles (run on the MI355X box from the repo root): kernel stats, PMC passes (default configuration; the two tile orders
# of the stand-alone QKV / c_fc GEMMs with the fused launch switched off), the default and the driver-form bench lines.
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out/r04 && export TMPDIR=/tmp
O=gpurun_out/r04
echo "== bench default" && timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || exit 1
echo "== bench driver form" && timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_20steps.json 2> $O/bench_20steps.err || exit 1
echo "== kernel stats"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python3 $PWD/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-dense-leg > $O/stats.log 2>&1 || exit 1
echo "== PMC passes (default)"
timeout -k 10 900 tools/pmc_passes.sh $O/pmc || exit 1
python tools/pmc_summary.py $O/pmc/fetch $O/pmc/write $O/r04 "" $O/pmc/fetch.bench.json || exit 1
python tools/pmc_util.py $O/r04_mfma_util.json $O/pmc/sq1 $O/pmc/sq2 $O/pmc/tcc > $O/r04_mfma_util.txt || exit 1
echo "== tile order PMC (two kernels, N groups vs M super-panels of 6)"
for P in 0 6; do
  export LEAF_FUSE_ATTN=0 LEAF_GEMM_MPANEL=$P
  B="python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense-leg"
  for pass in "fetch FETCH_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum" "sq SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    set -- $pass; name=$1; shift
    timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/order_p$P/$name -o run -- $B > $O/order_p${P}_$name.log 2>&1 || exit 1
  done
  python tools/pmc_util.py $O/order_p$P.json $O/order_p$P/fetch $O/order_p$P/tcc $O/order_p$P/sq > $O/order_p$P.txt || exit 1
  unset LEAF_FUSE_ATTN LEAF_GEMM_MPANEL
done
echo done
