#!/bin/bash
# Instruction-cache and wait counters of the fused specialised kernels (their code is 176-560 KB per kernel; the instruction cache 64 KB
# per two CUs): 3d_room and 4d_cylinders, one band, one frame in flight.
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
BENCH_ARGS="--repeats 1 --no-alone --frames-in-flight 1" PMC_DIR=r04_lat_room bash tools/pmc_latency_passes.sh > gpurun_out/r04_lat_room.txt 2>&1
BENCH_ARGS="--repeats 1 --no-alone --frames-in-flight 1 --scene 4d_cylinders.json" PMC_DIR=r04_lat_4dc bash tools/pmc_latency_passes.sh > gpurun_out/r04_lat_4dc.txt 2>&1
echo done
