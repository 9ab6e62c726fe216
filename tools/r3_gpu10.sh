#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 1500 python -m pytest tests -q -m gpu --timeout 900 > gpurun_out/r3_t7.log 2>&1
echo "pytest rc=$?"; tail -6 gpurun_out/r3_t7.log
bash tools/profile_all_configs.sh r03 2>&1 | tail -5
timeout -k 10 300 python tools/partition_balance.py > gpurun_out/r03_partition_balance.log 2>&1; tail -5 gpurun_out/r03_partition_balance.log
timeout -k 10 300 python bench.py --abi-child 1 > gpurun_out/r03_abi_child.log 2>&1; tail -2 gpurun_out/r03_abi_child.log
EU_BENCH_SMOKE_GLOO=1 timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r03_smoke_gloo.log 2>&1; echo "smoke gloo rc=$?"; tail -c 1500 gpurun_out/r03_smoke_gloo.log
