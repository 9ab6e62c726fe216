"""Diagnostic: frame pipeline time for tiny frames = the fixed cost of its launches.  Usage: python tools/fixed_cost.py [sync|off] [scene depth [wavefront|stack]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euclider_amd import Parser  # noqa: E402

spec = sys.argv[1] if len(sys.argv) > 1 else "sync"
scene = sys.argv[2] if len(sys.argv) > 2 else "3d_room.json"
kernel = sys.argv[4] if len(sys.argv) > 4 else None
env = Parser().parse_file(os.path.join(ROOT, "scenes", scene)).configure(specialize=spec, kernel=kernel)
env.camera.max_depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
print(scene, "depth", env.camera.max_depth, "specialize", spec, "kernel", kernel or "default")
for w, h in ((64, 64), (128, 96), (256, 256), (640, 360), (960, 540), (1920, 1080)):
    for _ in range(5):
        img = env.render((w, h))
    ms = env.kernel_ms_history(4)
    print("%4dx%-4d rays %8d kernel_ms %s" % (w, h, img.stats["rays"], " ".join("%.3f" % m for m in ms)))
env.close()
