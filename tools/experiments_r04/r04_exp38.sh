#!/bin/bash
# hunt for the runs of congruent entities: random rows of identical shape programs, specialised kernels against the oracle
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
python tools/congruent_scene_hunt.py 0 45 > gpurun_out/r04/congruent_hunt.txt 2>&1; tail -4 gpurun_out/r04/congruent_hunt.txt
