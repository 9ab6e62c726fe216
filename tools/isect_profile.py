"""Diagnostic: cycles of the intersect kernel per entity slot (needs a -DEU_PROFILE_ISECT build).
Usage: python tools/isect_profile.py [scene] [w] [h] [depth]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euclider_amd import Parser, _capi  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "3d_room.json"
w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 8
env = Parser().parse_file(os.path.join(ROOT, "scenes", scene))
env.camera.max_depth = depth
for _ in range(2):
    img = env.render((w, h))
ph = (C.c_uint64 * 16)()
_capi.lib().eu_renderer_debug_phases(env.renderer(0), ph)
tot = float(sum(ph)) or 1.0
print(scene, w, h, depth, "kernel_ms", env.kernel_ms(), "rays", img.stats["rays"])
for i in range(16):
    name = "entity %d" % i if i < 14 else ("load ray" if i == 14 else "store hit")
    print("  %-12s %6.2f%%  (%d cycles)" % (name, 100.0 * ph[i] / tot, ph[i]))
