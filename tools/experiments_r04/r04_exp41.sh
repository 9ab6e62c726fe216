#!/bin/bash
# shade window size and launch bounds of the fused specialised kernels, one frame at a time (the round-3 tuning was done on the two-kernel pipeline)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 0:0:-DEU_WF_WIN=1024 0:0:-DEU_SHADE_WAVES=3 0:0 0:0:-DEU_WF_WIN=1024 0:0:-DEU_SHADE_WAVES=3 > gpurun_out/r04/sweep_win_waves.txt 2>&1
grep -v "^GPU_MAX\|amdgpu.ids" gpurun_out/r04/sweep_win_waves.txt
