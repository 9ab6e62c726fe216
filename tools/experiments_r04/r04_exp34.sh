#!/bin/bash
# the whole GPU suite, then the bench record, on the round's final library
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu_final2.txt 2>&1; tail -4 gpurun_out/r04/pytest_gpu_final2.txt
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo bench rc=$?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1; echo smoke rc=$?; tail -4 gpurun_out/r04_smoke.txt
