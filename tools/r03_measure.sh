#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
bash tools/profile_all_configs.sh r03 2>&1 | tail -4
timeout -k 10 600 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --specialize off --frames-in-flight 1 --no-other-configs > gpurun_out/r03_bench_interpreter_one_frame.json 2>/dev/null; echo "bench interp rc=$?"
timeout -k 10 300 python bench.py --animate --steps 300 --warmup 20 > gpurun_out/r03_animate_bench.json 2>/dev/null; echo "animate rc=$?"
python tools/fixed_cost.py sync > gpurun_out/r03_fixed_cost.txt 2>&1; python tools/fixed_cost.py sync 3d_hallways.json 12 >> gpurun_out/r03_fixed_cost.txt 2>&1
lscpu | head -20 > gpurun_out/r03_gpu_box_host.txt; rocm-smi --showproductname 2>/dev/null | head -12 >> gpurun_out/r03_gpu_box_host.txt
