#!/bin/bash
# the SAME counters on the vendor's hipBLASLt kernel and on this repo's K loops, same box, same operands.
#   tools/vendor_counters.sh OUTDIR      (on the MI355X box, from the repo root; then: python tools/vendor_counters.py OUTDIR > table)
# One rocprofv3 --pmc pass per counter group (program directly behind `--`, --kernel-trace only), one GEMM shape per process
# (ONLY=fc | proj) so that the vendor kernel's dispatches of a run all belong to one shape, two row counts.
set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
OUT=${1:?usage: tools/vendor_counters.sh OUTDIR}
export TMPDIR=/tmp
mkdir -p "$OUT"
run() { tag=$1; name=$2; shift 2; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$tag/$name" -o run -- python3 "$PWD/tools/lib_gemm_ref.py" > "$OUT/$tag.$name.log" 2>&1; echo "pass $tag $name done"; }
for ROWS in ${ROWS_LIST:-107520 118016}; do
  for ONLY in ${SHAPES:-fc proj}; do
    export ROWS ONLY ITERS=6
    tag=${ONLY}_${ROWS}
    run $tag sq1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE
    run $tag sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE
    run $tag tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
    run $tag fetch FETCH_SIZE
    run $tag write WRITE_SIZE
  done
done
