#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --steps 40 > gpurun_out/r3_b4.json 2>gpurun_out/r3_b4.err; echo "bench rc=$?"; tail -3 gpurun_out/r3_b4.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3_b4.json").read().strip().split("\n")[-1])
print("headline", round(d["value"],1), round(d["ms_per_step"],3), d.get("parity"), d["config"]["one_frame_alone"], d["config"]["slots_agree"])
for o in d.get("other_configs", []): print("   ", o["workload"], round(o["value"],1), round(o["ms_per_step"],3), o["frames_in_flight"], o["slots_agree"])
PY
timeout -k 10 300 python bench.py --steps 20 --frames-in-flight 1 --no-other-configs --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('in flight 1', round(d['value'],1), round(d['ms_per_step'],3))"
EU_BENCH_SMOKE_GLOO=1 timeout -k 10 400 python bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r03_smoke_gloo.log 2>&1; echo "smoke gloo rc=$?"; tail -c 600 gpurun_out/r03_smoke_gloo.log
