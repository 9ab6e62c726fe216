#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/r3_t10.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/r3_t10.log
timeout -k 10 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r03_bench.json").read().strip().split("\n")[-1])
print("headline", round(d["value"],1), round(d["ms_per_step"],3), d.get("parity"), d["config"]["one_frame_alone"]["Mray/s"], d["config"]["slots_agree"])
for o in d.get("other_configs", []): print("   ", o["workload"], round(o["value"],1), round(o["ms_per_step"],3), o["frames_in_flight"], o["slots_agree"])
PY
