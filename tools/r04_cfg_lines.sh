#!/bin/bash
# bench lines of BASELINE.json configs[2..4] on one GPU (driver form: --steps 20 --warmup 5; bigG 8 / 2), PMC traffic of the same workload attached
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 400 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_cfg2.json 2> gpurun_out/cfg.err || exit 1
timeout -k 10 400 python bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r04_bench_cfg3.json 2>> gpurun_out/cfg.err || exit 1
timeout -k 10 600 python bench.py --config 4 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r04_bench_cfg4.json 2>> gpurun_out/cfg.err || exit 1
python - <<'PY'
import json
for c in (2, 3, 4):
    d = json.loads(open(f"gpurun_out/r04_bench_cfg{c}.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print(c, round(d["value"], 1), "samples/s", round(d["ms_per_step"], 1), "ms/step; dominant", r["kernel"], "frac", round(r["frac"], 3), "traffic", r["traffic"], "| dense leg", d.get("dense", {}).get("value"))
PY
