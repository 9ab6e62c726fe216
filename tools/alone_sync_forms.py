"""Diagnostic: ONE frame at a time (launch, pack, wait -- repeat) with the two ways a host can wait for it: hipDeviceSynchronize (what
bench.py's one_frame_alone uses: torch.cuda.synchronize) and hipStreamSynchronize on the stream the frame was submitted to.
Usage: python tools/alone_sync_forms.py [scene depth w h]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from euclider_amd import Parser  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "3d_room.json"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
dev = torch.device("cuda", 0)
env = Parser().parse_file(os.path.join(ROOT, "scenes", scene)).configure(specialize="sync")
env.camera.max_depth = depth
frame = env.frame(W, H, time=0.0, rows=(0, H))
rgba = torch.zeros((H, W), dtype=torch.int32, device=dev)
rgb = torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream(dev)
raw = st.cuda_stream
for form in ("device", "stream", "device", "stream"):
    ts = []
    for k in range(4 + 25):
        t1 = time.perf_counter()
        env.render_device(frame, rgba.data_ptr(), None, raw, device=0)
        env.pack_rgb_device(rgba.data_ptr(), rgb.data_ptr(), H * W, raw, device=0)
        if form == "device":
            torch.cuda.synchronize(dev)
        else:
            st.synchronize()
        if k >= 4:
            ts.append((time.perf_counter() - t1) * 1e3)
    ts.sort()
    kms = env.kernel_ms_history(8)
    rays = env.stats()["rays"]
    print("%-7s wall ms min %.4f median %.4f max %.4f   kernel_ms %.4f   %.0f Mray/s" % (form, ts[0], ts[len(ts) // 2], ts[-1], sum(kms) / len(kms), rays / ts[len(ts) // 2] / 1e3))
env.close()
