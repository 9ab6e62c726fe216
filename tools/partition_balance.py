"""Per-rank ray counts of BASELINE config 5 (3d_room 7680x4320 depth 8, 8-row strips dealt round-robin) for N = 2 / 4 / 8, from strip
renders on ONE GPU (eu_frame.strip_*: exactly what rank k of N traces), and the bytes each rank sends in the gather.
Writes profiles/r03_partition_balance.json.  Usage (GPU box): python tools/partition_balance.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402  (device buffers only)
from euclider_amd import Parser  # noqa: E402

W, H, DEPTH = 7680, 4320, 8
dev = torch.device("cuda", 0)
env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_room.json")).configure(specialize="sync")
env.camera.max_depth = DEPTH
stream = torch.cuda.current_stream(dev).cuda_stream
out = {"workload": "3d_room.json %dx%d depth %d, 8-row strips round-robin" % (W, H, DEPTH), "ranks": {}}
for n in (1, 2, 4, 8):
    rays, ms, rows = [], [], []
    for k in range(n):
        frame = env.frame(W, H, time=0.0, rows=(0, H), strips=(k, n) if n > 1 else None)
        lr = env.local_rows(frame)
        rgba = torch.zeros((lr, W), dtype=torch.int32, device=dev)
        env.render_device(frame, rgba.data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(3):
            env.render_device(frame, rgba.data_ptr(), None, stream)
        torch.cuda.synchronize(dev)
        ms.append((time.perf_counter() - t0) / 3 * 1e3)
        rays.append(env.stats()["rays"])
        rows.append(lr)
        del rgba
    mean = sum(rays) / n
    out["ranks"][str(n)] = {"rays_per_rank": rays, "max_over_mean_rays": max(rays) / mean, "trace_ms_per_rank_on_this_gpu": [round(x, 3) for x in ms],
                            "max_over_mean_ms": max(ms) / (sum(ms) / n), "rows_per_rank": rows, "gather_bytes_per_rank": [r * W * 3 for r in rows],
                            "gather_ms_at_153GBs_per_link": max(rows) * W * 3 / 153e9 * 1e3}
    print(n, "ranks: rays max/mean %.4f  ms max/mean %.4f  slowest rank %.2f ms  gather %.3f ms per link" % (
        out["ranks"][str(n)]["max_over_mean_rays"], out["ranks"][str(n)]["max_over_mean_ms"], max(ms), out["ranks"][str(n)]["gather_ms_at_153GBs_per_link"]), flush=True)
env.close()
out["note"] = ("every rank's strips rendered one after the other on one MI355X (the time a rank of an N-GPU job would need for its share); "
               "gather: each non-root rank sends its packed RGB8 strips over its own xGMI link to the root (7 links x ~153 GB/s, MI355X_MICROARCH.md), so the "
               "transfer time is one rank's bytes over one link")
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_partition_balance.json"), "w"), indent=1)
