#!/bin/bash
# round 3, first GPU pass: core parity on both kernel paths, then the headline with and without specialisation
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scene_parity or time_and_crosshair or row_tiles or edge_cases or trace_screen_point or strip_partition" > gpurun_out/r3_t1.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r3_t1.log
tail -5 gpurun_out/r3_t1.log
timeout -k 10 300 python bench.py --no-other-configs --specialize off --steps 20 > gpurun_out/r3_b_off.json 2> gpurun_out/r3_b_off.err; echo "bench off rc=$?"
timeout -k 10 300 python bench.py --no-other-configs --specialize sync --steps 20 > gpurun_out/r3_b_sync.json 2> gpurun_out/r3_b_sync.err; echo "bench sync rc=$?"
python - <<'PY'
import json
for n in ("off","sync"):
    try:
        d=json.loads(open("gpurun_out/r3_b_%s.json"%n).read().strip().split("\n")[-1])
        print(n, round(d["value"],1), "Mray/s", round(d["ms_per_step"],3), "ms kernel_ms", round(d["roofline"]["kernel_ms"],3), d.get("parity"))
    except Exception as e:
        print(n, "failed", e)
PY
