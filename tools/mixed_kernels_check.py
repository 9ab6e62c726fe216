"""Diagnostic: the specialised kernels under several budgets of the generator (jit.hpp) against the interpreter kernels, frame by frame.
Usage: python tools/mixed_kernels_check.py [scene depth]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euclider_amd import Parser  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "3d_room.json"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
path = os.path.join(ROOT, "scenes", scene)
a = Parser().parse_file(path).configure(specialize="off")
a.camera.max_depth = depth
ref = a.render((256, 144), time=0.75, want_hit_t=True)
a.close()
print(scene, "depth", depth, "interpreter rays", ref.stats["rays"])
for ops, surf, extra in ((256, 48, ""), (256, 48, "-DEU_JIT_NO_RUNS"), (3, 48, ""), (256, 2, ""), (256, 0, ""), (0, 48, ""), (0, 0, ""), (3, 2, "")):
    flags = "-DEU_JIT_OPS_BUDGET=%d -DEU_JIT_SURFACES_BUDGET=%d %s" % (ops, surf, extra)
    for rflags in (0, 2):
        b = Parser().parse_file(path).configure(specialize="sync", jit_flags=flags.strip(), flags=rflags, cache_dir="/tmp/eu_mixed_check")
        b.camera.max_depth = depth
        img = b.render((256, 144), time=0.75, want_hit_t=True)
        info = b.jit_info()
        bad = int((img.data != ref.data).any(axis=2).sum())
        both_nan = np.isnan(img.hit_t) & np.isnan(ref.hit_t)
        bad_t = int((img.hit_t[~both_nan] != ref.hit_t[~both_nan]).sum())
        print("ops %3d surfaces %2d %-18s rflags %d active %d: %6d pixels differ, %6d primary hit distances differ, stats equal %s  (%.1f s compile)" % (
            ops, surf, extra, rflags, info["active"], bad, bad_t, img.stats == ref.stats, info["compile_ms"] / 1e3))
        b.close()
