set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 0:0::256 0:0::512 0:0::1024 0:0 1:0 1:0::256 1:0::512 > gpurun_out/r04/sweep18_room.txt 2>&1
python tools/band_sweep.py 3d_room.json 8 640 360 0:0 0:0::256 0:0::512 > gpurun_out/r04/sweep18_room360.txt 2>&1
python tools/band_sweep.py 4d_cylinders.json 8 1920 1080 0:0 0:0::256 0:0::512 > gpurun_out/r04/sweep18_4dc.txt 2>&1
echo done
