#!/bin/bash
# left folds of congruent operands inside one shape program as a loop (4d_frame's four inner boxes, 4d_cylinders' union of eight spheres) against the straight line
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
N=-DEU_JIT_NO_FOLDS
for s in 4d_frame.json:8 3d_frame.json:8 4d_cylinders.json:8; do sc=${s%%:*}; d=${s##*:}; python tools/band_sweep.py $sc $d 1920 1080 0:0 0:0:$N 0:0 0:0:$N; done > gpurun_out/r04/sweep_folds.txt 2>&1
grep -v "^GPU_MAX\|amdgpu.ids" gpurun_out/r04/sweep_folds.txt
