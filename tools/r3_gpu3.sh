#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python - > gpurun_out/r3_diag2.log 2>&1 <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from euclider_amd import Parser
from oracle.scene_loader import load_scene_file
for scene, w, h, depth in (("4d_room.json", 160, 90, 2), ("4d_room.json", 160, 90, 3), ("4d_room.json", 160, 90, 10), ("4d_frame.json", 320, 180, 2), ("4d_frame.json", 320, 180, 3)):
    path = os.path.join("scenes", scene)
    orgb, ohit, ost = load_scene_file(path).render(w, h, max_depth=depth, want_hit_t=True)
    for mode in (dict(specialize="off"), dict(specialize="off", shade_scene_global=True)):
        env = Parser().parse_file(path).configure(**mode)
        env.camera.max_depth = depth
        for rep in range(3):
            img = env.render((w, h), want_hit_t=True)
            d = np.argwhere(img.data != orgb)
            print(scene, depth, mode, "rep", rep, "diff bytes", len(d), img.stats, "oracle", ost, d[:3].tolist())
        env.close()
PY
cat gpurun_out/r3_diag2.log
