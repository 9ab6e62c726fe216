set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 2:1000 \
   1:0:-DEU_WF_WIN=1024 2:1000:-DEU_WF_WIN=1024 3:1000:-DEU_WF_WIN=1024 4:1000:-DEU_WF_WIN=1024 \
   1:0:-DEU_WF_WIN=512 2:1000:-DEU_WF_WIN=512 3:1000:-DEU_WF_WIN=512 4:1000:-DEU_WF_WIN=512 \
   2:1000:-DEU_WF_WIN=256 3:1000:-DEU_WF_WIN=256 \
   2:1000:-DEU_WF_WIN=1024,-DEU_WF_DEAL_SHADE=1,-DEU_WF_WIN_MIN=512 3:1000:-DEU_WF_WIN=1024,-DEU_WF_DEAL_SHADE=1,-DEU_WF_WIN_MIN=512 \
   2:1000:-DEU_WF_WIN=1024,-DEU_WF_DEAL_ISECT=1 3:1000:-DEU_WF_WIN=1024,-DEU_WF_DEAL_ISECT=1 \
   2:750:-DEU_WF_WIN=1024 3:667:-DEU_WF_WIN=1024 4:500:-DEU_WF_WIN=1024 > gpurun_out/r04/sweep7_room.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 2:1000 1:0:-DEU_WF_WIN=1024 2:1000:-DEU_WF_WIN=1024 1:0:-DEU_WF_WIN=512 2:1000:-DEU_WF_WIN=512 3:1000:-DEU_WF_WIN=512 > gpurun_out/r04/sweep7_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 2:1000 1:0:-DEU_WF_WIN=512 2:1000:-DEU_WF_WIN=512 > gpurun_out/r04/sweep7_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 1:0 2:1000 1:0:-DEU_WF_WIN=512 2:1000:-DEU_WF_WIN=512 > gpurun_out/r04/sweep7_4dc.txt 2>&1 || exit 1
echo done
