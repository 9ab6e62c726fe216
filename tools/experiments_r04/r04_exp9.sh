set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 2:1000 3:1000 4:1000 2:1000:-DEU_WF_WIN=1024 3:1000:-DEU_WF_WIN=1024 4:1000:-DEU_WF_WIN=1024 3:667:-DEU_WF_WIN=1024 4:500:-DEU_WF_WIN=1024 6:500:-DEU_WF_WIN=1024 > gpurun_out/r04/sweep9_room.txt 2>&1 || exit 1
GPU_MAX_HW_QUEUES=4 python tools/band_sweep.py 3d_room.json 8 1920 1080 2:1000 3:1000 2:1000:-DEU_WF_WIN=1024 3:1000:-DEU_WF_WIN=1024 > gpurun_out/r04/sweep9_room_q4.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --streams 2 --permille 1000 > gpurun_out/r04/wgprof9_room_s2.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 2:1000 > gpurun_out/r04/sweep9_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 2:1000 3:1000 > gpurun_out/r04/sweep9_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 1:0 2:1000 3:1000 > gpurun_out/r04/sweep9_4dc.txt 2>&1 || exit 1
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r04/pytest_gpu9.txt 2>&1
tail -5 gpurun_out/r04/pytest_gpu9.txt
echo done
