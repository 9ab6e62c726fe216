#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python /tmp/diag.py > gpurun_out/r3_diag4.log 2>&1 || true
cat > /tmp/diag.py <<'PY'
PY
timeout -k 10 1500 python -m pytest tests -q -m gpu -x --timeout 900 > gpurun_out/r3_t3.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r3_t3.log
timeout -k 10 300 python bench.py --no-other-configs --specialize off --steps 20 > gpurun_out/r3_b_off.json 2> gpurun_out/r3_b_off.err; echo "bench off rc=$?"
timeout -k 10 400 python bench.py --specialize sync --steps 20 > gpurun_out/r3_b_sync.json 2> gpurun_out/r3_b_sync.err; echo "bench sync rc=$?"
python - <<'PY'
import json
for n in ("off","sync"):
    try:
        d=json.loads(open("gpurun_out/r3_b_%s.json"%n).read().strip().split("\n")[-1])
        print(n, round(d["value"],1), "Mray/s", round(d["ms_per_step"],3), "ms kernel_ms", round(d["roofline"]["kernel_ms"],3), d.get("parity"))
        for o in d.get("other_configs", []): print("   ", o["workload"], round(o["value"],1), round(o["ms_per_step"],3))
    except Exception as e:
        print(n, "failed", e)
PY
