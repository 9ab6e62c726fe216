#!/bin/bash
# the generator with fold loops: the whole GPU suite, hunts on specialised kernels (random scenes, rows of congruent entities), the bench record
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu_final3.txt 2>&1; tail -3 gpurun_out/r04/pytest_gpu_final3.txt
HUNT_BASE=900000 HUNT_PART=2 bash tools/r4_hunt.sh | tail -2
python tools/congruent_scene_hunt.py 2000 2040 > gpurun_out/r04/congruent_hunt3.txt 2>&1; tail -1 gpurun_out/r04/congruent_hunt3.txt
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo bench rc=$?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1; echo smoke rc=$?
