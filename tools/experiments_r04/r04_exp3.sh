set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
python tools/wg_profile.py 3d_room.json 8 --jit-flags=-DEU_WF_WIN_MIN=2048 > gpurun_out/r04/wgprof3_room_s1_w2048.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0:-DEU_WF_DYNAMIC=0 2:1000:-DEU_WF_DYNAMIC=0 \
   1:0:-DEU_WF_WIN_MIN=2048 2:1000:-DEU_WF_WIN_MIN=2048 1:0:-DEU_WF_WIN_MIN=1024 2:1000:-DEU_WF_WIN_MIN=1024 1:0 2:1000 \
   1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_SPREAD=2 1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_SPREAD=4 2:1000:-DEU_WF_WIN_MIN=2048,-DEU_WF_SPREAD=4 1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_SPREAD=8 \
   1:0:-DEU_WF_WIN_MIN=1024,-DEU_WF_SPREAD=4 2:1000:-DEU_WF_WIN_MIN=1024,-DEU_WF_SPREAD=4 \
   1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_STATIC_PCT=25 1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_STATIC_PCT=75 > gpurun_out/r04/sweep3_room.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0:-DEU_WF_DYNAMIC=0 1:0:-DEU_WF_WIN_MIN=2048 1:0:-DEU_WF_WIN_MIN=1024 1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_SPREAD=4 2:1000:-DEU_WF_WIN_MIN=2048 > gpurun_out/r04/sweep3_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0:-DEU_WF_DYNAMIC=0 1:0:-DEU_WF_WIN_MIN=2048 1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_SPREAD=4 > gpurun_out/r04/sweep3_4df.txt 2>&1 || exit 1
echo done
