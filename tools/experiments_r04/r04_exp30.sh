#!/bin/bash
# what bounds 4d_cylinders with frames in flight (0.386 ms per frame with and without runs of congruent entities): the private hit stack?
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
B="python bench.py --no-other-configs --no-cpu-baseline --no-alone --repeats 3"
for fl in "" "--jit-flags=-DEU_JIT_NO_RUNS" "--frames-in-flight 8" "--frames-in-flight 3"; do echo "4d_cylinders $fl"; $B --scene 4d_cylinders.json $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['frames_in_flight'])"; done > gpurun_out/r04/bench_4dc_in_flight.txt 2>&1
for fl in "" "--jit-flags=-DEU_HS_PRIVATE"; do echo "4d_frame $fl"; $B --scene 4d_frame.json $fl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['config']['frames_in_flight'])"; done >> gpurun_out/r04/bench_4dc_in_flight.txt 2>&1
cat gpurun_out/r04/bench_4dc_in_flight.txt
