set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 2:1000 3:1000 4:1000 1:0:-DEU_WF_SPREAD=1 2:1000:-DEU_WF_SPREAD=1 3:1000:-DEU_WF_SPREAD=1 4:1000:-DEU_WF_SPREAD=1 \
   2:1000:-DEU_WF_WIN=1024 3:1000:-DEU_WF_WIN=1024 3:1000:-DEU_WF_WIN=1024,-DEU_WF_SPREAD=1 2:1000::2 3:1000::2 > gpurun_out/r04/sweep12_room.txt 2>&1 || exit 1
GPU_MAX_HW_QUEUES=8 python tools/band_sweep.py 3d_room.json 8 1920 1080 3:1000 4:1000 > gpurun_out/r04/sweep12_room_q8.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 2:1000 3:1000 1:0:-DEU_WF_SPREAD=1 > gpurun_out/r04/sweep12_hall.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 2:1000 1:0:-DEU_WF_SPREAD=1 > gpurun_out/r04/sweep12_4df.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 1:0 2:1000 3:1000 1:0:-DEU_WF_SPREAD=1 > gpurun_out/r04/sweep12_4dc.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 10 1920 1080 1:0 2:1000 3:1000 > gpurun_out/r04/sweep12_room10.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 7680 4320 2:1000 3:1000 > gpurun_out/r04/sweep12_room8k.txt 2>&1 || exit 1
echo done
