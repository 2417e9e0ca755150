#!/bin/bash
# counters of the GEMM kernels alone (tools/gemm_bench.py) for the ping-pong kernel and the 256 x 256 kernel: tools/pp_pmc.sh OUTDIR
OUT=$1; mkdir -p $OUT; export TMPDIR=/tmp
for pp in 0 1; do
  export LEAF_GEMM_PP=$pp SEQS=1200 LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_variants.so
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pp${pp}_sq1 -o run -- python3 $PWD/tools/gemm_bench.py > $OUT/pp${pp}_sq1.log 2>&1 || exit 1
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pp${pp}_sq2 -o run -- python3 $PWD/tools/gemm_bench.py > $OUT/pp${pp}_sq2.log 2>&1 || exit 1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --kernel-trace --output-format csv -d $OUT/pp${pp}_tcc -o run -- python3 $PWD/tools/gemm_bench.py > $OUT/pp${pp}_tcc.log 2>&1 || exit 1
  python tools/pmc_util.py $OUT/pp${pp}_util.json $OUT/pp${pp}_sq1 $OUT/pp${pp}_sq2 $OUT/pp${pp}_tcc | grep gemm
done
