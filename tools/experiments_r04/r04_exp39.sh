#!/bin/bash
# the hunt for the runs of congruent entities, 200 more scenes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
python tools/congruent_scene_hunt.py 45 245 > gpurun_out/r04/congruent_hunt2.txt 2>&1; tail -3 gpurun_out/r04/congruent_hunt2.txt
