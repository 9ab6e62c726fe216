set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 2:1000 2:1000:-DEU_WF_WIN=1024 3:1000:-DEU_WF_WIN=1024 > gpurun_out/r04/sweep8_room.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 2:1000 > gpurun_out/r04/sweep8_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 2:1000 > gpurun_out/r04/sweep8_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 64 64 1:0 > gpurun_out/r04/sweep8_room64.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 64 64 1:0 > gpurun_out/r04/sweep8_hall64.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 64 64 > gpurun_out/r04/wgprof8_room64.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_hallways.json 12 > gpurun_out/r04/wgprof8_hall.txt 2>&1 || exit 1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu8.txt 2>&1
tail -5 gpurun_out/r04/pytest_gpu8.txt
echo done
