set -x
mkdir -p gpurun_out/r04
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r04/pytest_gpu_final.txt 2>&1
tail -4 gpurun_out/r04/pytest_gpu_final.txt
for s in 3d_hallways.json:12 4d_frame.json:8 4d_cylinders.json:8; do sc=${s%%:*}; d=${s##*:}; python tools/band_sweep.py $sc $d 1920 1080 1:0 2:0 1:0 2:0 0:0 >> gpurun_out/r04/sweep_final_light.txt 2>&1; done
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 3:0 2:0 >> gpurun_out/r04/sweep_final_room.txt 2>&1
N=2 bash tools/smoke_multirank.sh > gpurun_out/r04/multirank_final_n2.txt 2>&1
echo done
