#!/bin/bash
# (1) shade windows' chunks in the order "most expensive surface first" against the entity order of rounds 1-3, one frame at a time;
# (2) the mixed kernels of scenes beyond the generator's budgets: their tests, and the random scenes on the specialised pass.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
K=-DEU_SHADE_KEY_BY_ENTITY
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 0:0:$K 0:0 0:0:$K 1:0 1:0:$K > gpurun_out/r04/sweep_shade_key_room.txt 2>&1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 0:0 0:0:$K 0:0 0:0:$K > gpurun_out/r04/sweep_shade_key_hallways.txt 2>&1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 0:0 0:0:$K 0:0 0:0:$K > gpurun_out/r04/sweep_shade_key_4dc.txt 2>&1
cat gpurun_out/r04/sweep_shade_key_*.txt | grep -v "^GPU_MAX\|amdgpu.ids"
timeout -k 10 900 python -m pytest tests/test_gpu_jit.py -x -q > gpurun_out/r04/pytest_jit_mixed.txt 2>&1; tail -5 gpurun_out/r04/pytest_jit_mixed.txt
timeout -k 10 900 python -m pytest tests/test_gpu_random_scenes.py -x -q -k "jit and (many_entities or parity)" > gpurun_out/r04/pytest_random_mixed.txt 2>&1; tail -5 gpurun_out/r04/pytest_random_mixed.txt
