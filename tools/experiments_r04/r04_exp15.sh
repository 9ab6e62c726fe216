set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 > gpurun_out/r04/sweep15_a.txt 2>&1
EU_SWEEP_DUMMIES=8 python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 > gpurun_out/r04/sweep15_b.txt 2>&1
EU_SWEEP_DUMMIES=8 GPU_MAX_HW_QUEUES=8 python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 > gpurun_out/r04/sweep15_c.txt 2>&1
EU_SWEEP_DUMMIES=3 python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 > gpurun_out/r04/sweep15_d.txt 2>&1
python bench.py > gpurun_out/r04/bench15.json 2> gpurun_out/r04/bench15.err || { tail -5 gpurun_out/r04/bench15.err; exit 1; }
N=2 bash tools/smoke_multirank.sh > gpurun_out/r04/multirank_n2.txt 2>&1
N=4 bash tools/smoke_multirank.sh > gpurun_out/r04/multirank_n4.txt 2>&1
N=6 bash tools/smoke_multirank.sh > gpurun_out/r04/multirank_n6.txt 2>&1
echo done
