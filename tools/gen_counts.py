"""Diagnostic: rays per generation of one band (buffer set 0) for a scene / frame."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euclider_amd import Parser, _capi  # noqa: E402

for scene, depth in (("3d_room.json", 8), ("3d_hallways.json", 12), ("4d_frame.json", 8), ("4d_cylinders.json", 8)):
    env = Parser().parse_file(os.path.join(ROOT, "scenes", scene))
    env.camera.max_depth = depth
    img = env.render((1920, 1080))
    out = (C.c_uint64 * 17)()
    _capi.lib().eu_renderer_debug_generations(env.renderer(0), out)
    print(scene, "rays", img.stats["rays"], "band-0 generations:", list(out)[:depth + 1])
    env.close()
