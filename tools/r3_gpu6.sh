#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/r3_t4.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/r3_t4.log
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH_ARGS="" bash tools/kernel_stats.sh r3_room_jit > gpurun_out/r3_room_jit_kernel_stats.txt 2>&1; cat gpurun_out/r3_room_jit_kernel_stats.txt
BENCH_ARGS="--streams 1" bash tools/kernel_stats.sh r3_room_jit_1s > gpurun_out/r3_room_jit_kernel_stats_single_stream.txt 2>&1; cat gpurun_out/r3_room_jit_kernel_stats_single_stream.txt
BENCH_ARGS="--streams 1 --no-other-configs" bash tools/frame_dispatches.sh r3_room_jit > gpurun_out/r3_room_jit_dispatches.txt 2>&1; head -60 gpurun_out/r3_room_jit_dispatches.txt
for s in 1 2 3 4; do timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --streams $s --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('streams $s', round(d['value'],1), round(d['ms_per_step'],3))"; done
for n in 1 2 3; do timeout -k 10 200 python tools/frames_in_flight.py $n 3d_room.json 8 sync 2>/dev/null | tail -1; done
for n in 2 3; do timeout -k 10 200 python tools/frames_in_flight.py $n 3d_room.json 8 sync 1 2>/dev/null | tail -1; done
