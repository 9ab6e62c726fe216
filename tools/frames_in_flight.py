import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from euclider_amd import Parser
dev = torch.device("cuda", 0)
n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 2
scene = sys.argv[2] if len(sys.argv) > 2 else "3d_room.json"
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 8
spec = sys.argv[4] if len(sys.argv) > 4 else "sync"
n_streams = int(sys.argv[5]) if len(sys.argv) > 5 else 0
envs = [Parser().parse_file("scenes/" + scene).configure(specialize=spec, streams=n_streams) for _ in range(n_env)]
for e in envs: e.camera.max_depth = depth
W, H = 1920, 1080
streams = [torch.cuda.Stream(dev) for _ in range(n_env)]
rgba = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(n_env)]
rgb = [torch.empty((H * W * 3 + 16,), dtype=torch.uint8, device=dev) for _ in range(n_env)]
frames = [e.frame(W, H, time=0.0, rows=(0, H)) for e in envs]
def step(k):
    j = k % n_env
    envs[j].render_device(frames[j], rgba[j].data_ptr(), None, streams[j].cuda_stream, device=0)
    envs[j].pack_rgb_device(rgba[j].data_ptr(), rgb[j].data_ptr(), H * W, streams[j].cuda_stream, device=0)
for k in range(6): step(k)
torch.cuda.synchronize()
K = 40
t0 = time.perf_counter()
for k in range(K): step(k)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
rays = envs[0].stats()["rays"]
print(scene, depth, "specialize", spec, "renderers in flight", n_env, "band streams", n_streams or "default", "ms/frame %.3f" % (dt / K * 1e3), "Mray/s %.1f" % (rays * K / dt / 1e6))
assert torch.equal(rgb[0][:H*W*3], rgb[-1][:H*W*3])
