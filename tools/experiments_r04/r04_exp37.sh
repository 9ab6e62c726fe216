#!/bin/bash
# the N-rank control flow of the final bench.py on one GPU (gloo; N = 2 and 4), and per-kernel statistics of 4d_cylinders with and without runs of congruent entities
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out/r04
N=2 bash tools/smoke_multirank.sh > gpurun_out/r04/multirank_final2_n2.txt 2>&1; tail -3 gpurun_out/r04/multirank_final2_n2.txt | cut -c1-300
N=4 bash tools/smoke_multirank.sh > gpurun_out/r04/multirank_final2_n4.txt 2>&1; tail -2 gpurun_out/r04/multirank_final2_n4.txt | cut -c1-300
BENCH_ARGS="--scene 4d_cylinders.json --frames-in-flight 1" bash tools/kernel_stats.sh r04_4dc_runs > gpurun_out/r04/kernel_stats_4dc_runs.txt 2>&1; head -5 gpurun_out/r04/kernel_stats_4dc_runs.txt
BENCH_ARGS="--scene 4d_cylinders.json --frames-in-flight 1 --jit-flags=-DEU_JIT_NO_RUNS" bash tools/kernel_stats.sh r04_4dc_noruns > gpurun_out/r04/kernel_stats_4dc_noruns.txt 2>&1; head -5 gpurun_out/r04/kernel_stats_4dc_noruns.txt
