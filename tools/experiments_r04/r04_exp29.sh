#!/bin/bash
# final numbers of the round on the final kernels: the bench line, the animation, smoke, then random-scene hunts (specialised kernels, mixed kernels, interpreter)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo bench rc=$?
timeout -k 10 300 python bench.py --animate --steps 300 --warmup 20 > gpurun_out/r04_animate_bench.json 2>/dev/null; echo animate rc=$?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1; echo smoke rc=$?; tail -4 gpurun_out/r04_smoke.txt
HUNT_BASE=700000 HUNT_PART=4 bash tools/r4_hunt.sh | tail -4
HUNT_BASE=700000 HUNT_PART=2 bash tools/r4_hunt.sh | tail -3
