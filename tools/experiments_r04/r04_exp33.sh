#!/bin/bash
# A/B in one call on one box: the library of the last commit (resolve passes scan a byte per ray slot; eu_jit_fshade a natural 127 VGPRs) against the working
# tree's (lists of two-child nodes; eu_jit_fshade compiled for four waves: 128 VGPRs, 3 spilled).  variants_ab/head/libeuclider_amd.so = `git archive HEAD` built.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r04
out=gpurun_out/r04/ab_trans_list.txt; : > $out
for rep in 1 2; do
  for lib in variants_ab/head/libeuclider_amd.so euclider_amd/libeuclider_amd.so; do
    echo "== $lib" >> $out
    EU_LIB_PATH=$PWD/$lib python tools/band_sweep.py 3d_room.json 8 1920 1080 0:0 0:0 2>&1 | grep -v "^GPU_MAX\|amdgpu.ids" >> $out
    EU_LIB_PATH=$PWD/$lib timeout -k 10 300 python bench.py --no-other-configs --no-cpu-baseline --no-alone --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench pipelined', d['value'], d['ms_per_step'])" >> $out
  done
done
cat $out
