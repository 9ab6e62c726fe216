#!/bin/bash
# the bench record with twelve frames in flight on twelve hardware queues; the N = 2 rehearsal of the same script
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo bench rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('value_one_frame_at_a_time'), d['parity']['mismatch'], d['config']['frames_in_flight'], d['config']['gpu_max_hw_queues'], d['config']['timed_regions'])
for v in d['other_configs']: print(v['workload'][:44], round(v['value']), v['frames_in_flight'], round(v['one_frame_alone']['Mray/s']), round(v['one_frame_alone']['ms'],3), v['vs_oracle']['frame_equal'], v['slots_agree'])
PY
N=2 bash tools/smoke_multirank.sh > gpurun_out/r04/multirank_final3_n2.txt 2>&1; tail -3 gpurun_out/r04/multirank_final3_n2.txt | cut -c1-300
timeout -k 10 300 python bench.py --animate --steps 300 --warmup 20 > gpurun_out/r04_animate_bench.json 2>/dev/null; echo animate rc=$?
