set -x
mkdir -p gpurun_out/r04
EU_SWEEP_SPECIALIZE=off python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 1:0 1:0::2 2:0 2:0::2 3:0::2 > gpurun_out/r04/sweep16_interp_room.txt 2>&1
EU_SWEEP_SPECIALIZE=off python tools/band_sweep.py 3d_hallways.json 12 1920 1080 0:0 1:0::2 > gpurun_out/r04/sweep16_interp_hall.txt 2>&1
EU_SWEEP_SPECIALIZE=off python tools/band_sweep.py 4d_frame.json 8 1920 1080 0:0 1:0::2 > gpurun_out/r04/sweep16_interp_4df.txt 2>&1
python bench.py --no-other-configs --no-cpu-baseline --renderer-flags 2 > gpurun_out/r04/bench16_nofuse.json 2>/dev/null
python bench.py --no-other-configs --no-cpu-baseline > gpurun_out/r04/bench16_fuse.json 2>/dev/null
python bench.py --no-other-configs --no-cpu-baseline --streams 2 > gpurun_out/r04/bench16_fuse_s2.json 2>/dev/null
echo done
