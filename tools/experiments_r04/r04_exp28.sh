#!/bin/bash
# eu_jit_fshade with launch bounds of four waves per SIMD (128 VGPRs, 58-76 spilled) against its natural 148-154 (three waves), 4-D scenes
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
W=-DEU_FSHADE_WAVES=4
for s in 4d_room.json 4d_frame.json; do python tools/band_sweep.py $s 8 1920 1080 0:0 0:0:$W 0:0 0:0:$W 3:0 3:0:$W; done > gpurun_out/r04/sweep_fshade_waves_4d.txt 2>&1
grep -v "^GPU_MAX\|amdgpu.ids" gpurun_out/r04/sweep_fshade_waves_4d.txt
