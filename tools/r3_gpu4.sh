#!/bin/bash
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
cat > /tmp/diag.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, '.')
from euclider_amd import Parser
from oracle.scene_loader import load_scene_file
new = hasattr(Parser().parse_file("scenes/3d_fresnel.json"), "configure")
for scene, w, h, depth in (("4d_room.json", 160, 90, 3), ("4d_frame.json", 320, 180, 3)):
    path = os.path.join("scenes", scene)
    orgb, ohit, ost = load_scene_file(path).render(w, h, max_depth=depth, want_hit_t=True)
    env = Parser().parse_file(path)
    if new: env.configure(specialize="off")
    env.camera.max_depth = depth
    for rep in range(3):
        img = env.render((w, h), want_hit_t=True)
        d = np.argwhere(img.data != orgb)
        print(os.environ.get("EU_LIB_PATH", "default"), scene, depth, "rep", rep, "diff bytes", len(d), img.stats == ost)
    env.close()
PY
(python /tmp/diag.py; EU_LIB_PATH=variants/waves2.so python /tmp/diag.py; cd variants/old_tree && make -C oracle >/dev/null 2>&1; python /tmp/diag.py) > gpurun_out/r3_diag3.log 2>&1
cat gpurun_out/r3_diag3.log
