set -x
python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err; echo rc=$?
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r04_smoke.txt 2>&1; echo smoke rc=$?
tail -4 gpurun_out/r04_smoke.txt
