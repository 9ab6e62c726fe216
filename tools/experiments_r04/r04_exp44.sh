#!/bin/bash
# frames in flight on the round's final kernels: 6 / 8 / 12 / 16 (eight hardware queues)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
for n in 8 6 12 16 8; do python bench.py --no-other-configs --no-cpu-baseline --no-alone --repeats 3 --frames-in-flight $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('in flight', d['config']['frames_in_flight'], d['value'], d['ms_per_step'])"; done > gpurun_out/r04/frames_in_flight_final.txt 2>&1
cat gpurun_out/r04/frames_in_flight_final.txt
