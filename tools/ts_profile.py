"""Diagnostic: where the stream kernel's workgroups spend their time (needs a -DEU_TS_PROFILE build, EU_LIB_PATH).
Usage: python tools/ts_profile.py [scene] [w] [h] [depth]"""
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
for _ in range(3):
    img = env.render((w, h))
ph = (C.c_uint64 * 16)()
_capi.lib().eu_renderer_debug_phases(env.renderer(0), ph)
names = ["generate", "intersect", "sort + scene copy", "shade", "schedule", "resolve", "start-up"]
tot = float(sum(ph[:7])) or 1.0
wgs = ph[11] or 1
print(scene, w, h, depth, "kernel_ms", env.kernel_ms(), "rays", img.stats["rays"], "workgroups", wgs)
print("  mean clocks per workgroup %.0f (100 MHz s_memtime: %.3f ms), longest %.3f ms" % (tot / wgs, tot / wgs / 1e5, ph[12] / 1e5))
for i, n in enumerate(names):
    print("  %-20s %6.2f%%" % (n, 100.0 * ph[i] / tot))
print("  generate steps %d, chunks %d, rays per chunk %.1f" % (ph[8], ph[9], ph[10] / max(1, ph[9])))
