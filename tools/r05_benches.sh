#!/bin/bash
# tools/r05_benches.sh: the bench lines quoted in DESIGN 6 (round 4), one JSON per workload under gpurun_out/r05bench/
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05bench
mkdir -p $O
run() { local name=$1; shift; timeout -k 10 280 python3 $R/bench.py "$@" > $O/$name.json 2> $O/$name.err || echo "$name failed"; python3 - <<PY
import json
try:
    d = json.load(open("$O/$name.json")); r = d["roofline"]
    print("$name", d["value"], d["ms_per_step"], r.get("kernels_ms"), "frac", r.get("frac"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print("$name: no line", e)
PY
}
run normal_driverstyle --steps 20 --warmup 5
run normal
run normal_exact --soft-mode exact
run rach --workload rach
run rach_exact --workload rach --soft-mode exact --no-cpu-baseline
run config4 --workload config4
run config4_512 --workload config4 --streams 512 --chunks 25 --no-cpu-baseline
run config4_stateless --workload config4 --stateless-frontend --no-cpu-baseline
run config4_wideband --workload config4 --wideband 8
run config4_reference_chain --workload config4 --reference-chain
run config5 --workload config5
