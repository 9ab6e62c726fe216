"""Diagnostic: time + checksum of one configuration with the library EU_LIB_PATH selects (A/B builds).
Usage: [EU_LIB_PATH=variants/x.so] [EU_KERNEL=...] python tools/ab_frame.py scene depth [w h]"""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from euclider_amd import Parser  # noqa: E402

scene, depth = sys.argv[1], int(sys.argv[2])
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
env = Parser().parse_file(os.path.join(ROOT, "scenes", scene))
env.camera.max_depth = depth
img = env.render((W, H))
dev = torch.device("cuda", 0)
rgba = torch.zeros((H, W), dtype=torch.int32, device=dev)
frame = env.frame(W, H, time=0.0, rows=(0, H))
st = torch.cuda.current_stream(dev).cuda_stream
for _ in range(3):
    env.render_device(frame, rgba.data_ptr(), None, st, device=0)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    env.render_device(frame, rgba.data_ptr(), None, st, device=0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print("%-18s d%-2d %dx%d lib %-28s kernel %-9s %7.3f ms %8.1f Mray/s rays %d sha %s" % (
    scene, depth, W, H, os.path.basename(os.environ.get("EU_LIB_PATH", "default")), os.environ.get("EU_KERNEL", "wavefront"),
    dt * 1e3, img.stats["rays"] / dt / 1e6, img.stats["rays"], hashlib.sha1(img.data.tobytes()).hexdigest()[:12]))
