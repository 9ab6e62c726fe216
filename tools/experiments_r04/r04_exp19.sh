set -x
mkdir -p gpurun_out/r04
python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 0:0:-DEU_SHADE_FAST_PREP=0 0:0 0:0:-DEU_SHADE_FAST_PREP=0 1:0 1:0:-DEU_SHADE_FAST_PREP=0 > gpurun_out/r04/sweep19_room.txt 2>&1
python tools/band_sweep.py 3d_hallways.json 12 1920 1080 0:0 0:0:-DEU_SHADE_FAST_PREP=0 0:0 0:0:-DEU_SHADE_FAST_PREP=0 > gpurun_out/r04/sweep19_hall.txt 2>&1
python bench.py --no-other-configs --no-cpu-baseline --no-alone > gpurun_out/r04/bench19_fast.json 2>/dev/null
python bench.py --no-other-configs --no-cpu-baseline --no-alone --jit-flags=-DEU_SHADE_FAST_PREP=0 > gpurun_out/r04/bench19_old.json 2>/dev/null
python bench.py --no-other-configs --no-cpu-baseline --no-alone > gpurun_out/r04/bench19_fast2.json 2>/dev/null
echo done
