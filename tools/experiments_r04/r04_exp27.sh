#!/bin/bash
# identical surfaces share one function (3d_room's two glass solids): code 176 -> 144 KB, but eu_jit_fshade 127 -> 129 VGPRs (three waves
# per SIMD instead of four) unless its launch bounds ask for four (128 VGPRs, 3 spilled)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
N=-DEU_JIT_NO_SURFACE_SHARING
W=-DEU_FSHADE_WAVES=4
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0:$N 0:0 0:0:$W 0:0:$N,$W 0:0:$N 0:0 0:0:$W 0:0:$N,$W > gpurun_out/r04/sweep_surface_sharing.txt 2>&1
grep -v "^GPU_MAX\|amdgpu.ids" gpurun_out/r04/sweep_surface_sharing.txt
