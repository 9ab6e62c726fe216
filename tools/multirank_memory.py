"""Diagnostic: device memory one rank of `bench.py --gpus N` needs (the weak-scaled frame's strips of rank 0, `slots` renderers in flight),
measured on one GPU without starting N ranks.  Usage: python tools/multirank_memory.py [N] [slots]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from euclider_amd import Parser  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
slots = int(sys.argv[2]) if len(sys.argv) > 2 else 8
s = math.sqrt(n)
W, H = 8 * int(round(1920 * s / 8.0)), 8 * int(round(1080 * s / 8.0))
dev = torch.device("cuda", 0)
torch.cuda.init()
free0, total = torch.cuda.mem_get_info(dev)
envs = []
for k in range(slots):
    e = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_room.json")).configure(specialize="sync", streams=1)
    e.camera.max_depth = 8
    fr = e.frame(W, H, time=0.0, rows=(0, H), strips=(0, n))
    rows = e.local_rows(fr)
    rgba = torch.zeros((rows, W), dtype=torch.int32, device=dev)
    e.render_device(fr, rgba.data_ptr(), None, torch.cuda.current_stream(dev).cuda_stream, device=0)
    torch.cuda.synchronize(dev)
    envs.append((e, rgba))
    free, _ = torch.cuda.mem_get_info(dev)
    print("N=%d frame %dx%d, rank 0: %d rows; %d renderer(s) in flight: %.2f GB in use of %.0f GB" % (n, W, H, rows, k + 1, (free0 - free) / 2**30, total / 2**30), flush=True)
for e, _ in envs:
    e.close()
