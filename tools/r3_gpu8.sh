#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for fl in 0 2 4 6; do for sc in "3d_room.json 8" "3d_hallways.json 12"; do set -- $sc; timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --scene $1 --max-depth $2 --renderer-flags $fl --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('flags $fl $1', round(d['value'],1), round(d['ms_per_step'],3))"; done; done
