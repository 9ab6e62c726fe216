#!/bin/bash
# the fly-through (eu_sequence_*: camera motion, asynchronous read-back) with 4 / 8 / 12 slots
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
for n in 4 8 12 4; do timeout -k 10 300 python bench.py --animate --steps 300 --warmup 20 --slots $n 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots', $n, round(d['value']), d.get('fps') or d['config'].get('fps'))"; done > gpurun_out/r04/animate_slots.txt 2>&1
cat gpurun_out/r04/animate_slots.txt
