#!/bin/bash
# Out-of-suite random-scene hunt of round 4 (GPU box, from the repo root): the fused kernels, nodes finished at delivery, band pipelines of tiny frames,
# the two-kernel pipeline, f32, deep frames; interpreter kernels, then specialised kernels (a compilation per scene).  Parts: HUNT_PART=1|2|3|4.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
B=${HUNT_BASE:-500000}
out=gpurun_out/r04_random_scene_hunt_part${HUNT_PART:-1}.txt
: > $out
if [ "${HUNT_PART:-1}" = "1" ]; then
python tools/random_scene_hunt.py $B $((B + 4000)) >> $out 2>&1
HUNT_STREAMS=3 python tools/random_scene_hunt.py $((B + 4000)) $((B + 6000)) >> $out 2>&1
HUNT_FLAGS=2 python tools/random_scene_hunt.py $((B + 6000)) $((B + 7000)) >> $out 2>&1
HUNT_F32=1 python tools/random_scene_hunt.py $((B + 7000)) $((B + 8000)) >> $out 2>&1
HUNT_W=72 HUNT_H=40 HUNT_DEPTH=10 HUNT_STREAMS=2 python tools/random_scene_hunt.py $((B + 8000)) $((B + 8500)) >> $out 2>&1
elif [ "${HUNT_PART:-1}" = "2" ]; then
HUNT_SPECIALIZE=sync python tools/random_scene_hunt.py $((B + 10000)) $((B + 10060)) >> $out 2>&1
elif [ "${HUNT_PART:-1}" = "4" ]; then      # the generator's second half of the round: mixed kernels (budgets), fused and not
HUNT_SPECIALIZE=sync HUNT_MIXED=1 python tools/random_scene_hunt.py $((B + 13000)) $((B + 13024)) >> $out 2>&1
HUNT_SPECIALIZE=sync HUNT_MIXED=1 HUNT_FLAGS=2 python tools/random_scene_hunt.py $((B + 13100)) $((B + 13108)) >> $out 2>&1
else
HUNT_SPECIALIZE=sync HUNT_STREAMS=3 python tools/random_scene_hunt.py $((B + 11000)) $((B + 11030)) >> $out 2>&1
HUNT_SPECIALIZE=sync HUNT_F32=1 python tools/random_scene_hunt.py $((B + 12000)) $((B + 12020)) >> $out 2>&1
fi
grep -v "^at seed\|amdgpu.ids" $out
