#!/bin/bash
# BASELINE.json configs 1-3 (+ extras) on one GPU; one line each (diagnostic helper)
for cfg in "3d_room.json 8" "3d_hallways.json 12" "4d_frame.json 8" "4d_cylinders.json 8" "3d_room.json 10"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --scene $1 --max-depth $2 --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('%-20s depth %-3d %9.1f Mray/s %7.2f ms  rays %d  panic-events %d' % (d['config']['scene'], d['config']['max_depth'], d['value'], d['ms_per_step'], d['config']['rays_per_frame'], d['config']['would_panic_events']))"
done
