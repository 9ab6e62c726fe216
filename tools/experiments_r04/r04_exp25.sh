#!/bin/bash
# (1) the mixed kernels under several budgets against the interpreter kernels; (2) runs of congruent entities (4d_cylinders) against the
# straight-line form; (3) the stack kernel on small frames; (4) the two ways a host waits for a frame.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
python tools/mixed_kernels_check.py 3d_room.json 8 > gpurun_out/r04/mixed_check_room.txt 2>&1; cat gpurun_out/r04/mixed_check_room.txt | grep -v amdgpu.ids
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 0:0 0:0:-DEU_JIT_NO_RUNS 0:0 0:0:-DEU_JIT_NO_RUNS > gpurun_out/r04/sweep_runs_4dc.txt 2>&1; grep -v "^GPU_MAX\|amdgpu.ids" gpurun_out/r04/sweep_runs_4dc.txt
for k in wavefront stack; do python tools/fixed_cost.py off 3d_room.json 8 $k; done > gpurun_out/r04/fixed_cost_stack_kernel.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04/fixed_cost_stack_kernel.txt
python tools/alone_sync_forms.py > gpurun_out/r04/alone_sync_forms.txt 2>&1; grep -v amdgpu.ids gpurun_out/r04/alone_sync_forms.txt
timeout -k 10 600 python -m pytest tests/test_gpu_random_scenes.py -x -q -k "congruent" > gpurun_out/r04/pytest_congruent.txt 2>&1; tail -5 gpurun_out/r04/pytest_congruent.txt
