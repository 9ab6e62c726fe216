"""Golden checksums of full-size ORACLE frames for the workloads bench.py times besides the headline (CPU only, test infrastructure):
sha256 of the RGB8 frame and the ray count, written to tests/golden/frame_sha.json.  bench.py compares the frame its timed run left
in HBM with these, so that every line of `other_configs` carries a comparison with the oracle, not only with itself.
Usage: python tools/make_frame_sha.py"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import scene_loader as sl  # noqa: E402

CONFIGS = [("3d_hallways.json", 1920, 1080, 12, ""), ("4d_frame.json", 1920, 1080, 8, ""), ("4d_cylinders.json", 1920, 1080, 8, ""),
           ("3d_room.json", 1920, 1080, 10, ""), ("3d_room.json", 1920, 1080, 8, "f32"), ("3d_room.json", 1920, 1080, 8, ""),
           ("3d_room.json", 7680, 4320, 8, "")]
out_path = os.path.join(ROOT, "tests", "golden", "frame_sha.json")
out = json.load(open(out_path)) if os.path.exists(out_path) else {}
for scene, w, h, depth, variant in CONFIGS:
    key = "%s %dx%d depth %d%s" % (scene, w, h, depth, " f32" if variant else "")
    if key in out:
        continue
    t = time.time()
    rgb, _, st = sl.load_scene_file(os.path.join(ROOT, "scenes", scene), variant=variant).render(w, h, max_depth=depth, threads=os.cpu_count() or 1)
    out[key] = {"sha256": hashlib.sha256(rgb.tobytes()).hexdigest(), "rays": int(st["rays"]), "bytes": int(rgb.size),
                "made_by": "oracle/ (libeo_oracle%s.so), tools/make_frame_sha.py" % ("_f32" if variant else "")}
    print("%-40s rays %10d  %.1f s  %s" % (key, st["rays"], time.time() - t, out[key]["sha256"][:16]), flush=True)
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
