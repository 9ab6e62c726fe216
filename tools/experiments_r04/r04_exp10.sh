set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 2:1000 > gpurun_out/r04/sweep10_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 64 64 1:0 > gpurun_out/r04/sweep10_hall64.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 64 64 1:0 > gpurun_out/r04/sweep10_room64.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 > gpurun_out/r04/sweep10_4df.txt 2>&1 || exit 1
python bench.py > gpurun_out/r04/bench10.json 2> gpurun_out/r04/bench10.err || { tail -5 gpurun_out/r04/bench10.err; exit 1; }
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jit.py -x -q -m gpu > gpurun_out/r04/pytest_gpu10.txt 2>&1
tail -3 gpurun_out/r04/pytest_gpu10.txt
echo done
