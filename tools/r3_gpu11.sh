#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_low_precision.py -q -m gpu -x > gpurun_out/r3_t8.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_t8.log
for spec in sync off; do timeout -k 10 500 python bench.py --specialize $spec --steps 30 > gpurun_out/r3_b2_$spec.json 2>gpurun_out/r3_b2_$spec.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r3_b2_$spec.json").read().strip().split("\n")[-1])
print("$spec", round(d["value"],1), round(d["ms_per_step"],3), d.get("parity"), d["cpu_baseline"]["value"])
for o in d.get("other_configs", []): print("   ", o["workload"], round(o["value"],1), round(o["ms_per_step"],3))
PY
done
