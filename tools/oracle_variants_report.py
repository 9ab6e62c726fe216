"""Full-size numbers of the oracle's measurement builds for the BASELINE configurations (CPU only):
  - f64 operations of the reference algorithm per ray (libeo_oracle_flops.so)      -> profiles/r03_oracle_flops.json
  - RGB bytes / rays that change when glibc's libm replaces eo_math.h (libeo_oracle_libm.so) -> profiles/r03_oracle_libm.json
Usage: python tools/oracle_variants_report.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import scene_loader as sl  # noqa: E402

CONFIGS = [("3d_fresnel.json", 256, 256, 4), ("3d_room.json", 1920, 1080, 8), ("3d_hallways.json", 1920, 1080, 12),
           ("4d_frame.json", 1920, 1080, 8), ("4d_cylinders.json", 1920, 1080, 8), ("3d_room.json", 1920, 1080, 10)]
flops, libm = {}, {}
for scene, w, h, depth in CONFIGS:
    path = os.path.join(ROOT, "scenes", scene)
    key = "%s %dx%d depth %d" % (scene, w, h, depth)
    a, _, sa = sl.load_scene_file(path).render(w, h, max_depth=depth)
    sl.flops_take()
    f, _, sf = sl.load_scene_file(path, variant="flops").render(w, h, max_depth=depth)
    fl = sl.flops_take()
    assert np.array_equal(a, f)
    rays = sa["rays"]
    flops[key] = {"rays": rays, "total": fl, "per_ray": {k: v / rays for k, v in fl.items()}, "per_ray_all": sum(fl.values()) / rays}
    b, _, sb = sl.load_scene_file(path, variant="libm").render(w, h, max_depth=depth)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    libm[key] = {"bytes": int(d.size), "bytes_differing": int((d != 0).sum()), "share": float((d != 0).mean()), "max_abs_diff": int(d.max()),
                 "pixels_differing": int((d.max(axis=2) != 0).sum()), "rays_eo_math": rays, "rays_libm": sb["rays"]}
    print("%-36s rays %9d  flops/ray %7.1f (add/mul %.0f div %.1f sqrt %.1f transc %.1f) | libm: %d bytes differ (%.2f %%), rays %+d" % (
        key, rays, flops[key]["per_ray_all"], fl["add_mul"] / rays, fl["div"] / rays, fl["sqrt"] / rays, fl["transcendental"] / rays,
        libm[key]["bytes_differing"], 100 * libm[key]["share"], sb["rays"] - rays), flush=True)
note = ("counting rule: add/sub/mul = 1 each, div (and fmod) = 1, sqrt = 1, calls of acos/asin/sin/cos/tan/atan/atan2 = 1 each; negation, abs, "
        "floor, compares, selects and integer work are not counted; counted where the oracle (the reference's lazy algorithm) performs them")
json.dump({"note": note, "workloads": flops}, open(os.path.join(ROOT, "profiles", "r03_oracle_flops.json"), "w"), indent=1)
json.dump({"note": "RGB8 frame and ray count of the oracle built with glibc's libm (what Rust's f64 methods call on Linux) versus the shipped oracle (eo_math.h, <= 1 ulp from glibc)",
           "workloads": libm}, open(os.path.join(ROOT, "profiles", "r03_oracle_libm.json"), "w"), indent=1)
