"""Diagnostic: how much of the chip a frame's launches keep busy (tails, imbalance), from per-workgroup time stamps.
Kernels must carry -DEU_PROFILE_WG.  Usage: python tools/wg_profile.py scene depth [w h] [--streams N] [--permille P] [--jit-flags "..."]
Prints per launch: span, workgroups, mean / max workgroup time, busy share = sum(workgroup time) / (resident slots * span)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from euclider_amd import Parser  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("scene"); ap.add_argument("depth", type=int)
ap.add_argument("w", type=int, nargs="?", default=1920); ap.add_argument("h", type=int, nargs="?", default=1080)
ap.add_argument("--streams", type=int, default=1); ap.add_argument("--permille", type=int, default=0)
ap.add_argument("--jit-flags", default=""); ap.add_argument("--specialize", default="sync")
a = ap.parse_args()
env = Parser().parse_file(os.path.join(ROOT, "scenes", a.scene)).configure(
    specialize=a.specialize, streams=a.streams, band_grid_permille=a.permille, jit_flags=("-DEU_PROFILE_WG " + a.jit_flags).strip())
env.camera.max_depth = a.depth
for _ in range(3):
    img = env.render((a.w, a.h))
env.wg_profile()
img = env.render((a.w, a.h))
rec = env.wg_profile()
ms = env.kernel_ms_history(1)[0]
kind = (rec[:, 0] & 0xff).astype(int); gen = ((rec[:, 0] >> 8) & 0xff).astype(int); band = ((rec[:, 0] >> 16) & 0xff).astype(int)
t0 = rec[:, 1].astype(np.int64); t1 = rec[:, 2].astype(np.int64)
base = t0.min()
names = {0: "intersect", 1: "shade", 2: "resolve"}
print("%s depth %d %dx%d streams %d permille %d flags '%s': frame %.3f ms (events), %d rays, %d workgroup records, clock span %.3f ms" % (
    a.scene, a.depth, a.w, a.h, a.streams, a.permille, a.jit_flags, ms, img.stats["rays"], len(rec), (t1.max() - base) / 1e5))
grid = (rec[:, 3] >> 32).astype(int)
# launches: group by (kind, gen, grid) and by start-time clusters (bands share kind/gen)
tot_busy = 0.0
rows = []
for b in sorted(set(band)):
  for k in (0, 1, 2):
    for g in sorted(set(gen[(kind == k) & (band == b)])):
        m = (kind == k) & (gen == g) & (band == b)
        s0, s1 = t0[m], t1[m]
        span = (s1.max() - s0.min()) / 100.0      # us
        dur = (s1 - s0) / 100.0
        rows.append((s0.min(), "b%d %-9s g%-2d wgs %5d  start %8.1f us  span %7.1f us  wg mean %6.1f  p50 %6.1f  max %6.1f  last-start %6.1f  sum/span %6.1f wg" % (
            b, names[k], g, m.sum(), (s0.min() - base) / 100.0, span, dur.mean(), np.median(dur), dur.max(), (s0.max() - s0.min()) / 100.0, dur.sum() / max(span, 1e-9))))
        tot_busy += dur.sum()
for _, line in sorted(rows):
    print(line)
print("sum of workgroup time %.1f ms-wg; over the frame's clock span: %.0f workgroups busy on average" % (tot_busy / 1e3, tot_busy / ((t1.max() - base) / 100.0)))
env.close()
