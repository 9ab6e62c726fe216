"""Generates the golden fixtures of this directory with the ORACLE (oracle/), i.e. these are
self-generated regression vectors: the Rust reference cannot be built in this pipeline (no cargo/rustc),
so no reference-generated images exist.  Run from the repository root:  python tests/golden/make_golden.py
Regenerated in round 3: acos / asin / sin / cos of the oracle (and of the product) became correctly rounded (oracle/eo_math.h), which moves
the last bit of ~1-2 % of the bytes of 3d_room / 3d_hallways -- towards what a build of the reference on glibc produces
(tests/test_oracle_libm.py).
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.scene_loader import load_scene_file  # noqa: E402

CASES = [("3d_fresnel.json", 64, 64, 4), ("3d_room.json", 96, 54, 6), ("3d_hallways.json", 96, 54, 12),
         ("4d_frame.json", 96, 54, 8), ("4d_cylinders.json", 64, 36, 6), ("3d_fresnel_2.json", 48, 48, 6)]
index = {}
for scene, w, h, depth in CASES:
    osc = load_scene_file(os.path.join(ROOT, "scenes", scene))
    rgb, hit, st = osc.render(w, h, max_depth=depth, want_hit_t=True)
    name = "%s_%dx%d_d%d" % (scene.replace(".json", ""), w, h, depth)
    np.savez_compressed(os.path.join(os.path.dirname(__file__), name + ".npz"), rgb=rgb, hit_t=hit)
    index[name] = {"scene": scene, "width": w, "height": h, "max_depth": depth, "stats": st,
                   "rgb_crc32": zlib.crc32(rgb.tobytes()) & 0xffffffff}
json.dump(index, open(os.path.join(os.path.dirname(__file__), "index.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(index, indent=1))
