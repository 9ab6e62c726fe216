#!/bin/bash
# Out-of-suite random-scene hunt of round 3 (GPU box, from the repo root): interpreter kernels f64 / f32 / deep, then specialised kernels.
# (No pipes behind the hunts: a run that stays silent for seven minutes is taken to be hung.)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
out=gpurun_out/r03_random_scene_hunt_slab.txt
: > $out
python tools/random_scene_hunt.py ${HUNT_LO:-140000} ${HUNT_HI:-144000} >> $out 2>&1
HUNT_F32=1 python tools/random_scene_hunt.py 145000 146000 >> $out 2>&1
HUNT_W=72 HUNT_H=40 HUNT_DEPTH=10 python tools/random_scene_hunt.py 146000 146500 >> $out 2>&1
HUNT_SPECIALIZE=sync python tools/random_scene_hunt.py 147000 147100 >> $out 2>&1
HUNT_SPECIALIZE=sync HUNT_F32=1 python tools/random_scene_hunt.py 148000 148030 >> $out 2>&1
grep -v "^at seed\|amdgpu.ids" $out
