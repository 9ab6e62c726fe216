#!/bin/bash
# resolve passes that visit a list of the generation's two-child nodes (one entry per node, appended by the shade kernel) instead of scanning a byte per ray slot
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 0:0 1:0 > gpurun_out/r04/sweep_trans_list.txt 2>&1
python tools/band_sweep.py 3d_room.json 10 1920 1080 0:0 >> gpurun_out/r04/sweep_trans_list.txt 2>&1
grep -v "^GPU_MAX\|amdgpu.ids" gpurun_out/r04/sweep_trans_list.txt
timeout -k 10 300 python bench.py --no-other-configs --no-cpu-baseline --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['config']['one_frame_alone']['ms'], d['config']['one_frame_alone']['Mray/s'])"
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py -x -q > gpurun_out/r04/pytest_trans_list.txt 2>&1; tail -4 gpurun_out/r04/pytest_trans_list.txt
BENCH_ARGS="--frames-in-flight 1" bash tools/kernel_stats.sh r04tl 2>&1 | head -8
