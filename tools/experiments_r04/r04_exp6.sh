set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
O="-DEU_WF_EQUAL_WIN=0,-DEU_SHADE_TAKE_CHUNKS=0"
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0:$O 2:1000:$O 1:0:-DEU_WF_EQUAL_WIN=0 1:0:-DEU_SHADE_TAKE_CHUNKS=0 1:0 2:1000 3:1000 \
   1:0:-DEU_WF_DEAL_SHADE=1,-DEU_WF_WIN_MIN=1024 2:1000:-DEU_WF_DEAL_SHADE=1,-DEU_WF_WIN_MIN=1024 1:0:-DEU_WF_DEAL_SHADE=1,-DEU_WF_WIN_MIN=512 2:1000:-DEU_WF_DEAL_SHADE=1,-DEU_WF_WIN_MIN=512 \
   1:0:-DEU_WF_DEAL_ISECT=1 2:1000:-DEU_WF_DEAL_ISECT=1 1:0:-DEU_WF_WIN=1024 2:1000:-DEU_WF_WIN=1024 > gpurun_out/r04/sweep6_room.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 > gpurun_out/r04/wgprof6_room.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --streams 2 --permille 1000 > gpurun_out/r04/wgprof6_room_s2.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0:$O 1:0 2:1000 > gpurun_out/r04/sweep6_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0:$O 1:0 2:1000 > gpurun_out/r04/sweep6_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 1:0:$O 1:0 > gpurun_out/r04/sweep6_4dc.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 64 64 1:0:$O 1:0 > gpurun_out/r04/sweep6_room64.txt 2>&1 || exit 1
echo done
