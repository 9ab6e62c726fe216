set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 1:0 0:0 > gpurun_out/r04/sweep20_room.txt 2>&1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 0:0 0:0 > gpurun_out/r04/sweep20_hall.txt 2>&1
python bench.py --no-other-configs --no-cpu-baseline --no-alone > gpurun_out/r04/bench20.json 2>/dev/null
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py -x -q -m gpu > gpurun_out/r04/pytest_gpu20.txt 2>&1
tail -3 gpurun_out/r04/pytest_gpu20.txt
echo done
