#!/bin/bash
# the whole GPU suite on the kernels with mixed budgets and runs of congruent entities
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu_runs.txt 2>&1; tail -6 gpurun_out/r04/pytest_gpu_runs.txt
