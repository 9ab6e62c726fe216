#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - > gpurun_out/r3_diag.log 2>&1 <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from euclider_amd import Parser
from oracle.scene_loader import load_scene_file
for scene, w, h, depth in (("4d_frame.json", 320, 180, 8), ("4d_frame.json", 320, 180, 1), ("4d_room.json", 160, 90, 10), ("4d_cylinders.json", 160, 90, 8), ("4d_fresnel.json", 128, 128, 10)):
    path = os.path.join("scenes", scene)
    orgb, ohit, ost = load_scene_file(path).render(w, h, max_depth=depth, want_hit_t=True)
    for mode in (dict(specialize="off"), dict(specialize="sync"), dict(kernel="stack")):
        env = Parser().parse_file(path).configure(**mode)
        env.camera.max_depth = depth
        img = env.render((w, h), want_hit_t=True)
        d = np.argwhere(img.data != orgb)
        both_nan = np.isnan(img.hit_t) & np.isnan(ohit)
        hd = int((img.hit_t[~both_nan] != ohit[~both_nan]).sum())
        print(scene, depth, mode, "diff bytes", len(d), "hit_t diffs", hd, "stats eq", img.stats == ost, d[:4].tolist())
        env.close()
PY
cat gpurun_out/r3_diag.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_low_precision.py -q -m gpu > gpurun_out/r3_t2.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/r3_t2.log
