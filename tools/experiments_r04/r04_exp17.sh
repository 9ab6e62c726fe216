set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 1:0 > gpurun_out/r04/sweep17_room.txt 2>&1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 0:0 > gpurun_out/r04/sweep17_hall.txt 2>&1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 0:0 > gpurun_out/r04/sweep17_4df.txt 2>&1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 0:0 > gpurun_out/r04/sweep17_4dc.txt 2>&1
python tools/band_sweep.py 3d_room.json 1 1920 1080 0:0 > gpurun_out/r04/sweep17_room_d1.txt 2>&1
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu17.txt 2>&1
tail -4 gpurun_out/r04/pytest_gpu17.txt
python bench.py --no-other-configs > gpurun_out/r04/bench17.json 2>/dev/null
echo done
