#!/bin/bash
# Out-of-suite random-scene hunt of round 3 (GPU box, from the repo root): interpreter kernels f64 / f32 / deep, then specialised kernels.
# (No pipes behind the hunts: a run that stays silent for seven minutes is taken to be hung.)  Usage: [HUNT_BASE=140000] tools/r3_hunt.sh [scale]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B=${HUNT_BASE:-140000}; S=${1:-1}
out=gpurun_out/r03_random_scene_hunt_slab.txt
: > $out
python tools/random_scene_hunt.py $B $((B + 4000 * S)) >> $out 2>&1
HUNT_F32=1 python tools/random_scene_hunt.py $((B + 5000 * S)) $((B + 6000 * S)) >> $out 2>&1
HUNT_W=72 HUNT_H=40 HUNT_DEPTH=10 python tools/random_scene_hunt.py $((B + 6000 * S)) $((B + 6500 * S)) >> $out 2>&1
HUNT_SPECIALIZE=sync python tools/random_scene_hunt.py $((B + 7000 * S)) $((B + 7100 * S)) >> $out 2>&1
HUNT_SPECIALIZE=sync HUNT_F32=1 python tools/random_scene_hunt.py $((B + 8000 * S)) $((B + 8030 * S)) >> $out 2>&1
grep -v "^at seed\|amdgpu.ids" $out
