#!/bin/bash
# the bench record with every one-frame-alone number measured before the pipelined runs; then random-scene hunts on the final library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d.get('value_one_frame_at_a_time'), d['parity']['mismatch'])
for v in d['other_configs']: print(v['workload'][:44], round(v['value']), round(v['one_frame_alone']['Mray/s']), round(v['one_frame_alone']['ms'],3), v['vs_oracle']['frame_equal'])
PY
HUNT_BASE=800000 HUNT_PART=1 bash tools/r4_hunt.sh | tail -6
HUNT_BASE=800000 HUNT_PART=3 bash tools/r4_hunt.sh | tail -3
