#!/bin/bash
# What produced profiles/r04_*: kernel statistics, dispatch traces and PMC passes of configs 2-4 (one frame at a time, one band: the counters
# of overlapping kernels would mix), workgroup time stamps, the bench lines, small frames, the multi-rank rehearsal.  Usage (GPU box): bash tools/r04_measure.sh
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile_all_configs.sh r04 2>&1 | tail -4
python tools/wg_profile.py 3d_room.json 8 --streams 1 > gpurun_out/r04_wg_profile_one_band.txt 2>&1
python tools/wg_profile.py 3d_room.json 8 --streams 3 --permille 1000 > gpurun_out/r04_wg_profile_three_bands.txt 2>&1
python tools/wg_profile.py 3d_room.json 8 64 64 > gpurun_out/r04_wg_profile_64x64.txt 2>&1
python tools/wg_profile.py 3d_hallways.json 12 > gpurun_out/r04_wg_profile_hallways.txt 2>&1
python tools/wg_profile.py 4d_frame.json 8 > gpurun_out/r04_wg_profile_4dframe.txt 2>&1
timeout -k 10 900 python bench.py > gpurun_out/r04_bench.json 2> gpurun_out/r04_bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --specialize off --frames-in-flight 1 --no-other-configs > gpurun_out/r04_bench_interpreter_one_frame.json 2>/dev/null; echo "bench interp rc=$?"
timeout -k 10 300 python bench.py --animate --steps 300 --warmup 20 > gpurun_out/r04_animate_bench.json 2>/dev/null; echo "animate rc=$?"
python tools/fixed_cost.py sync > gpurun_out/r04_fixed_cost.txt 2>&1; python tools/fixed_cost.py sync 3d_hallways.json 12 >> gpurun_out/r04_fixed_cost.txt 2>&1
lscpu | head -20 > gpurun_out/r04_gpu_box_host.txt; rocm-smi --showproductname 2>/dev/null | head -12 >> gpurun_out/r04_gpu_box_host.txt
