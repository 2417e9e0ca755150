#!/bin/bash
# rocprofv3 PMC passes over a short bench run (run on the MI355X box from the repo root):  tools/pmc_passes.sh OUTDIR [bench args]
# One pass per counter group (SQ 8 slots, GRBM 2, TCC 4; FETCH_SIZE / WRITE_SIZE in passes of their own), the program directly
# behind `--`, --kernel-trace only.  PMC_TRAFFIC_ONLY=1: only the FETCH_SIZE / WRITE_SIZE passes.  Summaries: tools/pmc_util.py (MFMA busy, clock, L2, LDS) and tools/pmc_summary.py (HBM traffic).
set -e
OUT=$1; shift
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-dense-leg $*"
run() { name=$1; shift; export LEAF_BENCH_JSON_OUT=$PWD/$OUT/$name.bench.json; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -o run -- $BENCH > $OUT/$name.log 2>&1; echo "pass $name done"; }
mkdir -p $OUT
if [ -z "$PMC_TRAFFIC_ONLY" ]; then
run sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
fi
run fetch FETCH_SIZE
run write WRITE_SIZE
