set -x
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu14.txt 2>&1
tail -6 gpurun_out/r04/pytest_gpu14.txt
python bench.py > gpurun_out/r04/bench14.json 2> gpurun_out/r04/bench14.err || { tail -5 gpurun_out/r04/bench14.err; exit 1; }
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 1:0 > gpurun_out/r04/sweep14_room.txt 2>&1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 0:0 > gpurun_out/r04/sweep14_4dc.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 0:0 > gpurun_out/r04/sweep14_4dc_q8.txt 2>&1
python tools/multirank_memory.py 8 8 > gpurun_out/r04/multirank_memory.txt 2>&1
echo done
