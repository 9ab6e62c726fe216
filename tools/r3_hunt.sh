#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
python tools/random_scene_hunt.py 100000 106000
HUNT_F32=1 python tools/random_scene_hunt.py 110000 112000
HUNT_W=72 HUNT_H=40 HUNT_DEPTH=10 python tools/random_scene_hunt.py 120000 121000
HUNT_SPECIALIZE=sync python tools/random_scene_hunt.py 130000 130200
HUNT_SPECIALIZE=sync HUNT_F32=1 python tools/random_scene_hunt.py 131000 131060
} > gpurun_out/r03_random_scene_hunt.txt 2>&1
tail -30 gpurun_out/r03_random_scene_hunt.txt | grep -v "^at seed"
