#!/bin/bash
# round-end check on the GPU box: the whole GPU suite, smoke, then the two bench lines that go under profiles/
set -o pipefail
cd "${GRAFT_REPO_ROOT:?}" && mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/b.err || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r04_bench_20steps.json 2>> gpurun_out/b.err || exit 1
python - <<'PY'
import json
for f in ["gpurun_out/r04_bench_default.json", "gpurun_out/r04_bench_20steps.json"]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["roofline"]["traffic"], round(d["roofline"]["frac"], 4), d["roofline"]["sampled_steps"])
PY
