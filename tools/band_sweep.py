"""Diagnostic: ONE frame at a time (launch, wait, repeat) for several band-pipeline configurations of one renderer.
Usage: python tools/band_sweep.py scene depth w h  streams:permille[:jitflags[:renderer_flags]] ...   (permille 0 = library default)
Prints wall ms per frame (host launch + device, synchronised after every frame), HIP-event ms, Mray/s, checksum."""
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from euclider_amd import Parser  # noqa: E402

scene, depth, W, H = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = torch.device("cuda", 0)
print("GPU_MAX_HW_QUEUES", os.environ.get("GPU_MAX_HW_QUEUES", "(unset: 4)"))
# EU_SWEEP_DUMMIES=n: n other renderers (one band each, a frame traced on a torch stream of its own) stay alive during the sweep -- how a
# process full of streams changes what a renderer's band streams get from the runtime's hardware queues
dummies = []
for k in range(int(os.environ.get("EU_SWEEP_DUMMIES", "0"))):
    e = Parser().parse_file(os.path.join(ROOT, "scenes", scene)).configure(specialize="sync", streams=1)
    e.camera.max_depth = depth
    st_k = torch.cuda.Stream(dev)
    buf = torch.zeros((H, W), dtype=torch.int32, device=dev)
    e.render_device(e.frame(W, H, time=0.0, rows=(0, H)), buf.data_ptr(), None, st_k.cuda_stream, device=0)
    dummies.append((e, st_k, buf))
torch.cuda.synchronize()
if dummies:
    print("dummy renderers alive:", len(dummies))
for spec in sys.argv[5:]:
    parts = spec.split(":")
    streams, permille = int(parts[0]), int(parts[1])
    flags = (parts[2].replace(",", " ") or None) if len(parts) > 2 else None
    rflags = int(parts[3]) if len(parts) > 3 else 0
    env = Parser().parse_file(os.path.join(ROOT, "scenes", scene)).configure(specialize=os.environ.get("EU_SWEEP_SPECIALIZE", "sync"), streams=streams, band_grid_permille=permille, jit_flags=flags, flags=rflags)
    env.camera.max_depth = depth
    img = env.render((W, H))
    rgba = torch.zeros((H, W), dtype=torch.int32, device=dev)
    rgb = torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device=dev)
    frame = env.frame(W, H, time=0.0, rows=(0, H))
    st = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(3):
        env.render_device(frame, rgba.data_ptr(), None, st, device=0)
    torch.cuda.synchronize()
    K = 15
    best = 1e9
    t0 = time.perf_counter()
    for _ in range(K):
        t1 = time.perf_counter()
        env.render_device(frame, rgba.data_ptr(), None, st, device=0)
        env.pack_rgb_device(rgba.data_ptr(), rgb.data_ptr(), H * W, st, device=0)
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t1)
    dt = (time.perf_counter() - t0) / K
    kms = env.kernel_ms_history(K)
    sha = hashlib.sha1(rgb[:H * W * 3].cpu().numpy().tobytes()).hexdigest()[:10]
    print("%-16s d%-2d %dx%d streams %d permille %4d rflags %d flags %-28s wall %.3f ms (best %.3f, host issue %.3f)  events %.3f ms  %7.0f Mray/s  sha %s %s" % (
        scene, depth, W, H, streams, permille, rflags, flags or "-", dt * 1e3, best * 1e3, (t2 - t1) * 1e3, sum(kms) / len(kms), img.stats["rays"] / dt / 1e6, sha,
        hashlib.sha1(img.data.tobytes()).hexdigest()[:10]), flush=True)
    env.close()
    del rgba, rgb
