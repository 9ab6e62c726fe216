#!/bin/bash
# the generator on the F = f32 pair: runs of congruent entities and mixed kernels
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
HUNT_F32=1 python tools/congruent_scene_hunt.py 1000 1040 > gpurun_out/r04/congruent_hunt_f32.txt 2>&1; tail -2 gpurun_out/r04/congruent_hunt_f32.txt
HUNT_SPECIALIZE=sync HUNT_MIXED=1 HUNT_F32=1 python tools/random_scene_hunt.py 714000 714012 > gpurun_out/r04/mixed_hunt_f32.txt 2>&1; tail -2 gpurun_out/r04/mixed_hunt_f32.txt
