set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 1:0::2 2:1000 2:1000::2 3:1000 > gpurun_out/r04/sweep11_room.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 1:0::2 2:1000 > gpurun_out/r04/sweep11_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 1:0::2 > gpurun_out/r04/sweep11_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 1:0 1:0::2 > gpurun_out/r04/sweep11_4dc.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 64 64 1:0 1:0::2 > gpurun_out/r04/sweep11_room64.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 64 64 1:0 1:0::2 > gpurun_out/r04/sweep11_hall64.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 640 360 1:0 1:0::2 > gpurun_out/r04/sweep11_room360.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 > gpurun_out/r04/wgprof11_room.txt 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu11.txt 2>&1
tail -5 gpurun_out/r04/pytest_gpu11.txt
echo done
