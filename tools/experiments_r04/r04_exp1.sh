set -x
mkdir -p gpurun_out/r04
export GPU_MAX_HW_QUEUES=8
python tools/wg_profile.py 3d_room.json 8 > gpurun_out/r04/wgprof_room_s1.txt 2>&1 || exit 1
python tools/wg_profile.py 3d_room.json 8 --jit-flags=-DEU_WF_INTERLEAVE=1 > gpurun_out/r04/wgprof_room_s1_il.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 2:1000 2:0 3:0 4:1000 4:500 4:250 6:333 6:500 8:1000 8:500 8:250 8:125 > gpurun_out/r04/sweep_room_q8.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0:-DEU_WF_INTERLEAVE=1 2:1000:-DEU_WF_INTERLEAVE=1 4:500:-DEU_WF_INTERLEAVE=1 8:250:-DEU_WF_INTERLEAVE=1 8:500:-DEU_WF_INTERLEAVE=1 > gpurun_out/r04/sweep_room_q8_il.txt 2>&1 || exit 1
GPU_MAX_HW_QUEUES=4 python tools/band_sweep.py 3d_room.json 8 1920 1080 1:0 2:1000 3:667 4:500 8:250 > gpurun_out/r04/sweep_room_q4.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 1:0 2:1000 4:500 8:250 8:500 4:500:-DEU_WF_INTERLEAVE=1 > gpurun_out/r04/sweep_hall_q8.txt 2>&1 || exit 1
python tools/band_sweep.py 4d_frame.json 8 1920 1080 1:0 2:1000 4:500 8:250 > gpurun_out/r04/sweep_4df_q8.txt 2>&1 || exit 1
python tools/band_sweep.py 3d_room.json 8 64 64 1:0 1:0:-DEU_WF_INTERLEAVE=1 > gpurun_out/r04/sweep_room_64.txt 2>&1 || exit 1
echo done
