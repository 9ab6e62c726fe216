#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "scene_parity or strip or row_tiles or edge or banded or render_multi" > gpurun_out/r3_t6.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r3_t6.log
run() { timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs --steps 30 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(sys.argv[1:], round(d['value'],1), round(d['ms_per_step'],3))" "$@"; }
for sc in "3d_room.json 8" "3d_hallways.json 12" "4d_frame.json 8"; do set -- $sc
  run --scene $1 --max-depth $2
  run --scene $1 --max-depth $2 --jit-flags "-DEU_SHADE_WAVES=4"
  run --scene $1 --max-depth $2 --jit-flags "-DEU_SHADE_WAVES=4 -DEU_ISECT_WAVES=4"
  run --scene $1 --max-depth $2 --jit-flags "-DEU_ISECT_PREFETCH=0"
done
run --specialize off
