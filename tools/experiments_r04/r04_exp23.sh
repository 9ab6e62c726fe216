#!/bin/bash
# Where the fused shade kernel's wave time goes, section by section, and the same weighted with the active lanes (3d_room, one band).
cd "$(dirname "$0")/../.."
for l in "" 1; do
  EU_PROFILE_JIT=1 EU_PROFILE_KERNEL=shade EU_PROFILE_LANES=$l python tools/shape_profile.py 3d_room.json 8 > gpurun_out/r04_shade_sections${l:+_lanes}.txt 2>&1
done
cat gpurun_out/r04_shade_sections.txt gpurun_out/r04_shade_sections_lanes.txt
