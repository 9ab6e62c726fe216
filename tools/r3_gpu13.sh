#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for n in 1 2 3 4 5 6; do timeout -k 10 200 python tools/frames_in_flight.py $n 3d_room.json 8 sync 1 2>/dev/null | tail -1; done
for n in 2 3; do timeout -k 10 200 python tools/frames_in_flight.py $n 3d_room.json 8 sync 2 2>/dev/null | tail -1; done
for sc in "3d_hallways.json 12" "4d_frame.json 8"; do set -- $sc; for n in 1 3 4; do timeout -k 10 200 python tools/frames_in_flight.py $n $1 $2 sync 1 2>/dev/null | tail -1; done; done
