#!/bin/bash
# frames in flight x hardware queues on the round's final kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
: > gpurun_out/r04/frames_in_flight_final2.txt
for q in 8 12 16; do for n in 10 12 14; do GPU_MAX_HW_QUEUES=$q python bench.py --no-other-configs --no-cpu-baseline --no-alone --repeats 3 --frames-in-flight $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('queues', d['config']['gpu_max_hw_queues'], 'in flight', d['config']['frames_in_flight'], round(d['value']), round(d['ms_per_step'],4), d['config']['slots_agree'])" >> gpurun_out/r04/frames_in_flight_final2.txt; done; done
cat gpurun_out/r04/frames_in_flight_final2.txt
