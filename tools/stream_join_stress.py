"""Diagnostic: N renderers, each on its own non-blocking stream (the pattern of eu_render_multi and eu_sequence_*), 1080p frames
traced as two concurrent bands inside every renderer; after every round all images are compared with a reference frame.
Usage: python tools/stream_join_stress.py [n_renderers] [rounds] [low_precision]"""
import os
import sys

sys.path.insert(0, os.getcwd())
import torch
from euclider_amd import Parser

n_env = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lp = len(sys.argv) > 3 and sys.argv[3] == "1"
dev = torch.device("cuda", 0)
W, H = 1920, 360
envs = [Parser(low_precision=lp).parse_file("scenes/3d_room.json") for _ in range(n_env)]
for e in envs:
    e.camera.max_depth = 8
ref = torch.from_numpy(envs[0].render((W, H)).data.copy()).to(dev).reshape(-1)
streams = [torch.cuda.Stream(dev) for _ in range(n_env)]
rgba = [torch.zeros((H, W), dtype=torch.int32, device=dev) for _ in range(n_env)]
rgb = [torch.zeros((H * W * 3 + 16,), dtype=torch.uint8, device=dev) for _ in range(n_env)]
frames = [e.frame(W, H, time=0.0, rows=(0, H)) for e in envs]
bad = 0
for it in range(rounds):
    for j in range(n_env):
        rgba[j].zero_(); rgb[j].zero_()
    torch.cuda.synchronize()
    for j in range(n_env):
        envs[j].render_device(frames[j], rgba[j].data_ptr(), None, streams[j].cuda_stream, device=0)
        envs[j].pack_rgb_device(rgba[j].data_ptr(), rgb[j].data_ptr(), H * W, streams[j].cuda_stream, device=0)
    torch.cuda.synchronize()
    for j in range(n_env):
        if not torch.equal(rgb[j][:H * W * 3], ref):
            bad += 1
            d = (rgb[j][:H * W * 3] != ref).reshape(H, W * 3).any(dim=1).nonzero().flatten()
            print("round", it, "renderer", j, "differs in", int(d.numel()), "rows, first", d[:4].tolist(), flush=True)
print("renderers", n_env, "rounds", rounds, "low_precision", lp, "EU_WF_STREAMS", os.environ.get("EU_WF_STREAMS"), "mismatching images:", bad)
