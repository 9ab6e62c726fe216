set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
python tools/wg_profile.py 3d_room.json 8 > gpurun_out/r04/wgprof2_room_s1.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --jit-flags=-DEU_WF_WIN_MIN=1024 > gpurun_out/r04/wgprof2_room_s1_w1024.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --jit-flags=-DEU_WF_WIN_MIN=256 > gpurun_out/r04/wgprof2_room_s1_w256.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 2:1000 3:1000 1:0:-DEU_WF_DYNAMIC=0 2:1000:-DEU_WF_DYNAMIC=0 \
   1:0:-DEU_WF_WIN_MIN=1024 2:1000:-DEU_WF_WIN_MIN=1024 1:0:-DEU_WF_WIN_MIN=256 2:1000:-DEU_WF_WIN_MIN=256 \
   1:0:-DEU_WF_DEAL_FACTOR=8 2:1000:-DEU_WF_DEAL_FACTOR=8 1:0:-DEU_WF_DEAL_FACTOR=2 \
   1:0:-DEU_WF_STATIC_PCT=25 1:0:-DEU_WF_STATIC_PCT=75 2:1000:-DEU_WF_STATIC_PCT=25 > gpurun_out/r04/sweep2_room.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 2:1000 1:0:-DEU_WF_DYNAMIC=0 1:0:-DEU_WF_WIN_MIN=256 > gpurun_out/r04/sweep2_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 2:1000 1:0:-DEU_WF_DYNAMIC=0 > gpurun_out/r04/sweep2_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 1:0 2:1000 1:0:-DEU_WF_DYNAMIC=0 > gpurun_out/r04/sweep2_4dc.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 64 64 1:0 1:0:-DEU_WF_DYNAMIC=0 > gpurun_out/r04/sweep2_room_64.txt 2>&1 || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r04/pytest_parity2.txt 2>&1
tail -3 gpurun_out/r04/pytest_parity2.txt
echo done
