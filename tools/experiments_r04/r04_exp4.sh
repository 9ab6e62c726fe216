set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
S="-DEU_WF_DYNAMIC=0"
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0:$S 2:1000:$S 1:0:$S,-DEU_SHADE_PREFETCH=0 2:1000:$S,-DEU_SHADE_PREFETCH=0 \
   2:1000:$S:200 2:1000:$S:400 2:1000:$S:0:2 2:1000:$S:200:2 3:1000:$S:300 3:1000:$S:300:2 4:1000:$S:400 4:1000:$S:400:2 4:500:$S:400:2 \
   1:0:-DEU_WF_WIN_MIN=2048 2:1000:-DEU_WF_WIN_MIN=2048 1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_STATIC_PCT=25 \
   1:0:$S,-DEU_WF_SPREAD=8 2:1000:$S,-DEU_WF_SPREAD=8 1:0:-DEU_WF_WIN_MIN=2048,-DEU_WF_SPREAD=8 \
   1:0:-DEU_WF_WIN_MIN=1024 2:1000:-DEU_WF_WIN_MIN=1024 > gpurun_out/r04/sweep4_room.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --jit-flags=-DEU_WF_DYNAMIC=0 > gpurun_out/r04/wgprof4_room_static.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --streams 2 --permille 1000 --jit-flags=-DEU_WF_DYNAMIC=0 > gpurun_out/r04/wgprof4_room_static_s2.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --jit-flags=-DEU_WF_WIN_MIN=2048 > gpurun_out/r04/wgprof4_room_dyn_isect.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0:$S 1:0:$S,-DEU_SHADE_PREFETCH=0 1:0:-DEU_WF_WIN_MIN=2048 1:0:$S,-DEU_WF_SPREAD=8 2:1000:$S:300:2 > gpurun_out/r04/sweep4_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0:$S 1:0:$S,-DEU_SHADE_PREFETCH=0 1:0:-DEU_WF_WIN_MIN=2048 2:1000:$S:300:2 > gpurun_out/r04/sweep4_4df.txt 2>&1 || exit 1
echo done
