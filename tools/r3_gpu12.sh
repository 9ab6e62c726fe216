#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/r3_t9.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t9.log
timeout -k 10 500 python bench.py --specialize sync --steps 30 > gpurun_out/r3_b3_sync.json 2>gpurun_out/r3_b3_sync.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r3_b3_sync.json").read().strip().split("\n")[-1])
print("sync", round(d["value"],1), round(d["ms_per_step"],3), d.get("parity"), d["cpu_baseline"]["value"])
for o in d.get("other_configs", []): print("   ", o["workload"], round(o["value"],1), round(o["ms_per_step"],3))
PY
