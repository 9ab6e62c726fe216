#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_jit.py -q -m gpu -x 2>&1 | tail -3
{
HUNT_SPECIALIZE=sync python tools/random_scene_hunt.py 130000 130120
HUNT_SPECIALIZE=sync HUNT_F32=1 python tools/random_scene_hunt.py 131000 131040
} 2>&1 | tee gpurun_out/r03_random_scene_hunt_jit.txt | grep -v "amdgpu.ids"
