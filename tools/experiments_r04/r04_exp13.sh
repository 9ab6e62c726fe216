set -x
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu13.txt 2>&1
tail -6 gpurun_out/r04/pytest_gpu13.txt
python bench.py > gpurun_out/r04/bench13.json 2> gpurun_out/r04/bench13.err || { tail -5 gpurun_out/r04/bench13.err; exit 1; }
echo done
