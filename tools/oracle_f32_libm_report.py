"""What `low_precision` is worth (round-3 verdict, item 6): the reference with F = f32 calls the f32 libm (acosf, asinf, sinf, cosf, tanf,
atan2f: util.rs:712-722, surface.rs:225,280, Cargo.toml:18-20); the f32 build here (and its oracle) evaluate the elementary functions in
f64 and round.  Full-size frames of the f32 oracle against the same oracle built on glibc's f32 functions (libeo_oracle_f32_libm.so),
and -- for scale -- against the f64 oracle.  CPU only.  -> profiles/r04_oracle_f32_libm.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import scene_loader as sl  # noqa: E402

CONFIGS = [("3d_fresnel.json", 256, 256, 4), ("3d_room.json", 1920, 1080, 8), ("3d_hallways.json", 1920, 1080, 12), ("4d_frame.json", 1920, 1080, 8)]
out = {}
for scene, w, h, depth in CONFIGS:
    path = os.path.join(ROOT, "scenes", scene)
    key = "%s %dx%d depth %d" % (scene, w, h, depth)
    a, _, sa = sl.load_scene_file(path, variant="f32").render(w, h, max_depth=depth)
    b, _, sb = sl.load_scene_file(path, variant="f32_libm").render(w, h, max_depth=depth)
    c, _, sc = sl.load_scene_file(path).render(w, h, max_depth=depth)
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    e = np.abs(a.astype(np.int16) - c.astype(np.int16))
    out[key] = {"bytes": int(d.size), "f32_vs_f32_libm": {"bytes_differing": int((d != 0).sum()), "share": float((d != 0).mean()), "max_abs_diff": int(d.max()),
                                                          "rays": [int(sa["rays"]), int(sb["rays"])]},
                "f32_vs_f64": {"bytes_differing": int((e != 0).sum()), "share": float((e != 0).mean()), "max_abs_diff": int(e.max()), "rays": [int(sa["rays"]), int(sc["rays"])]}}
    print("%-36s f32 vs f32-libm: %7d bytes differ (%.3f %%, max %d), rays %d / %d | f32 vs f64: %.2f %% (max %d)" % (
        key, out[key]["f32_vs_f32_libm"]["bytes_differing"], 100 * out[key]["f32_vs_f32_libm"]["share"], out[key]["f32_vs_f32_libm"]["max_abs_diff"],
        sa["rays"], sb["rays"], 100 * out[key]["f32_vs_f64"]["share"], out[key]["f32_vs_f64"]["max_abs_diff"]), flush=True)
json.dump({"note": "RGB8 frames of libeo_oracle_f32.so (elementary functions evaluated in f64 by eo_math.h and rounded to f32 -- the checker of libeuclider_amd_f32.so) "
                   "against libeo_oracle_f32_libm.so (glibc's acosf/asinf/sinf/cosf/tanf/atanf/atan2f: what the reference's low_precision binary calls), and against the f64 oracle",
           "workloads": out}, open(os.path.join(ROOT, "profiles", "r04_oracle_f32_libm.json"), "w"), indent=1)
