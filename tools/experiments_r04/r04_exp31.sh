#!/bin/bash
# 4d_frame's generation-0 intersect (65 % of its frame) at four waves per SIMD: private hit stack (LDS no longer limits) and launch bounds of four
# (128 VGPRs, 85 spilled) against three (168)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
P=-DEU_HS_PRIVATE
python tools/band_sweep.py 4d_frame.json 8 1920 1080 0:0 0:0:$P 0:0:$P,-DEU_ISECT_WAVES=4 0:0 0:0:$P 0:0:$P,-DEU_ISECT_WAVES=4 > gpurun_out/r04/sweep_4dframe_isect_waves.txt 2>&1
grep -v "^GPU_MAX\|amdgpu.ids" gpurun_out/r04/sweep_4dframe_isect_waves.txt
