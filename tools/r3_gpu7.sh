#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -q -m gpu --timeout 900 -x > gpurun_out/r3_t5.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/r3_t5.log
for spec in off sync; do timeout -k 10 400 python bench.py --no-cpu-baseline --specialize $spec --steps 30 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$spec', round(d['value'],1), round(d['ms_per_step'],3))
for o in d.get('other_configs', []): print('   ', o['workload'], round(o['value'],1), round(o['ms_per_step'],3))"; done
for s in 1 3; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --streams $s --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('streams $s', round(d['value'],1), round(d['ms_per_step'],3))"; done
for n in 2 3; do timeout -k 10 200 python tools/frames_in_flight.py $n 3d_room.json 8 sync 1 2>/dev/null | tail -1; done
