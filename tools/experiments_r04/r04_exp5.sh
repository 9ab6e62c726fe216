set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
N="-DEU_WF_DEAL_ISECT=0,-DEU_WF_DEAL_SHADE=0"
I="-DEU_WF_DEAL_SHADE=0"
H="-DEU_WF_DEAL_ISECT=0"
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0:$N 2:1000:$N 1:0:$I 2:1000:$I 1:0:$I,-DEU_WORK_SHARDS=32 1:0:$I,-DEU_WORK_SHARDS=8 1:0:$I,-DEU_WF_STATIC_PCT=25 2:1000:$I,-DEU_WF_STATIC_PCT=25 1:0:$I,-DEU_WF_STATIC_PCT=0 \
   1:0:$H,-DEU_WF_WIN_MIN=1024 2:1000:$H,-DEU_WF_WIN_MIN=1024 1:0:$H,-DEU_WF_WIN_MIN=512 2:1000:$H,-DEU_WF_WIN_MIN=512 1:0:$H,-DEU_WF_WIN_MIN=1024,-DEU_WF_DEAL_FACTOR=2 \
   1:0:-DEU_WF_WIN_MIN=1024 2:1000:-DEU_WF_WIN_MIN=1024 3:1000:-DEU_WF_WIN_MIN=1024 1:0:-DEU_WF_WIN_MIN=1024,-DEU_WF_STATIC_PCT=25 > gpurun_out/r04/sweep5_room.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --jit-flags="-DEU_WF_WIN_MIN=1024" > gpurun_out/r04/wgprof5_room_dyn.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0:$N 1:0:$I 1:0:-DEU_WF_WIN_MIN=1024 2:1000:-DEU_WF_WIN_MIN=1024 > gpurun_out/r04/sweep5_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0:$N 1:0:$I 1:0:-DEU_WF_WIN_MIN=1024 > gpurun_out/r04/sweep5_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 64 64 1:0:$N 1:0 > gpurun_out/r04/sweep5_room64.txt 2>&1 || exit 1
echo done
