#!/bin/bash
# where 4d_frame's generation-0 intersect spends a wave's time (s_memtime shares of the shape routines, specialised kernels)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
EU_PROFILE_JIT=1 python tools/shape_profile.py 4d_frame.json 8 > gpurun_out/r04/shape_profile_4dframe.txt 2>&1
EU_PROFILE_JIT=1 python tools/shape_profile.py 3d_hallways.json 12 > gpurun_out/r04/shape_profile_hallways.txt 2>&1
cat gpurun_out/r04/shape_profile_4dframe.txt gpurun_out/r04/shape_profile_hallways.txt | grep -v amdgpu.ids
