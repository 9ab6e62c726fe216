"""Diagnostic: frame pipeline time for tiny frames = the fixed cost of its ~27 launches."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euclider_amd import Parser  # noqa: E402

env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_room.json"))
env.camera.max_depth = 8
for w, h in ((64, 64), (256, 256), (640, 360), (960, 540), (1920, 1080)):
    for _ in range(5):
        img = env.render((w, h))
    ms = env.kernel_ms_history(4)
    print("%4dx%-4d rays %8d kernel_ms %s" % (w, h, img.stats["rays"], " ".join("%.3f" % m for m in ms)))
env.close()
