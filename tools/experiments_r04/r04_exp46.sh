#!/bin/bash
# frames in flight x hardware queues, second sweep
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
: > gpurun_out/r04/frames_in_flight_final3.txt
for qn in 12:12 12:16 12:24 24:12 24:24 12:12 8:8; do q=${qn%%:*}; n=${qn##*:}; GPU_MAX_HW_QUEUES=$q python bench.py --no-other-configs --no-cpu-baseline --no-alone --repeats 3 --frames-in-flight $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues', d['config']['gpu_max_hw_queues'], 'in flight', d['config']['frames_in_flight'], round(d['value']), round(d['ms_per_step'],4), d['config']['slots_agree'])" >> gpurun_out/r04/frames_in_flight_final3.txt; done
cat gpurun_out/r04/frames_in_flight_final3.txt
