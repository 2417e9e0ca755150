#!/bin/bash
# Where the fused QKV + attention launch's LDS bank conflicts come from: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of
# tools/qkv_attn_bench.py on the shipped build, on a build without the q|k|v staging stores and on one without the attention stage
# (make -C leaf_amd/csrc qa_nostage qa_noattn; garbage results).  One rocprofv3 --pmc pass per build, the program directly behind `--`.
# Second pass (shipped and qa_noattn): vector-ALU instructions and VALU-busy cycles -- their difference is the attention stage's.
#   tools/qa_conflicts.sh OUTDIR
set -e
cd "${GRAFT_REPO_ROOT:-$PWD}"
OUT=${1:?usage: tools/qa_conflicts.sh OUTDIR}; mkdir -p "$OUT"
export TMPDIR=/tmp
for b in shipped qa_nostage qa_noattn; do
  if [ $b = shipped ]; then unset LEAF_HIP_LIB; else export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_$b.so; fi
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/$b" -o run -- python3 "$PWD/tools/qkv_attn_bench.py" --iters 4 > "$OUT/$b.log" 2>&1
  python3 - "$OUT/$b" $b <<'PY'
import csv, glob, sys
from collections import defaultdict
d, name = sys.argv[1], sys.argv[2]
acc = defaultdict(float); n = set(); dur = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qkv_attn_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
k = max(len(n), 1)
print(f"{name:12s} dispatches {k:3d}  mean {sum(dur.values()) / k / 1e3:7.1f} us  LDS conflict cycles / LDS active cycles = "
      f"{acc['SQ_LDS_BANK_CONFLICT'] / max(acc['SQ_LDS_IDX_ACTIVE'], 1):.3f}  (conflict {acc['SQ_LDS_BANK_CONFLICT'] / k:.3g}, active {acc['SQ_LDS_IDX_ACTIVE'] / k:.3g}, LDS instructions {acc['SQ_INSTS_LDS'] / k:.3g} per dispatch)")
PY
done
for b in shipped qa_noattn; do
  if [ $b = shipped ]; then unset LEAF_HIP_LIB; else export LEAF_HIP_LIB=$PWD/tools/diag/libleaf_hip_$b.so; fi
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/${b}_valu" -o run -- python3 "$PWD/tools/qkv_attn_bench.py" --iters 4 > "$OUT/${b}_valu.log" 2>&1
  python3 - "$OUT/${b}_valu" $b <<'PY'
import csv, glob, sys
from collections import defaultdict
d, name = sys.argv[1], sys.argv[2]
acc = defaultdict(float); n = set(); dur = {}
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "qkv_attn_kernel" not in r["Kernel_Name"]:
            continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
k = max(len(n), 1)
print(f"{name:12s} dispatches {k:3d}  mean {sum(dur.values()) / k / 1e3:7.1f} us  per dispatch: VALU instructions {acc['SQ_INSTS_VALU'] / k:.4g}, VALU-active "
      f"{acc['SQ_ACTIVE_INST_VALU'] / k:.4g}, SALU {acc['SQ_INSTS_SALU'] / k:.4g}, LDS {acc['SQ_INSTS_LDS'] / k:.4g}, busy cycles {acc['SQ_BUSY_CYCLES'] / k:.4g}, "
      f"wave cycles {acc['SQ_WAVE_CYCLES'] / k:.4g}, GRBM_GUI_ACTIVE {acc['GRBM_GUI_ACTIVE'] / k:.4g}")
PY
done
