"""Diagnostic: where eval_shape (or, EU_PROFILE_KERNEL=shade, the shade kernel) spends a wave's time.  The interpreter kernels need a
-DEU_PROFILE_SHAPE / -DEU_PROFILE_SHADE_WAVE build of the library (tools/build_variant.sh, EU_LIB_PATH=...); with EU_PROFILE_JIT=1 the
scene-specialised kernels are compiled with that flag instead (any build of the library).  The stamps cost time themselves: read the
shares as a ranking, not as a budget.
Usage: [EU_PROFILE_JIT=1] [EU_PROFILE_KERNEL=shade] python tools/shape_profile.py [scene] [depth] [w] [h]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from euclider_amd import Parser, _capi  # noqa: E402

scene = sys.argv[1] if len(sys.argv) > 1 else "4d_frame.json"
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
h = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
env = Parser().parse_file(os.path.join(ROOT, "scenes", scene))
if os.environ.get("EU_PROFILE_JIT"):
    env.configure(specialize="sync", jit_flags=("-DEU_PROFILE_SHADE_WAVE" if os.environ.get("EU_PROFILE_KERNEL") == "shade" else "-DEU_PROFILE_SHAPE") +
                  (" -DEU_PROFILE_SHAPE_LANES" if os.environ.get("EU_PROFILE_LANES") else ""))      # EU_PROFILE_LANES=1: every share weighted with the lanes active at its stamp (x / 64)
env.camera.max_depth = depth
img = env.render((w, h))
ph = (C.c_uint64 * 16)()
_capi.lib().eu_renderer_debug_phases(env.renderer(0), ph)
tot = float(sum(ph[:12])) or 1.0
print(scene, w, h, depth, "kernel_ms", env.kernel_ms(), "rays", img.stats["rays"])
names = ["chain: bound test, set-up", "chain: matrices (t_k, IN, LT)", "chain: merge cascade", "chain: push to the hit stack", "leaf ops",
         "composite merges (incl. inside tests)", "entity loop between the ops (entity record, bounding-sphere test, root op, result selection)", "top-level Union chain: first element directly",
         "kernel prologue (scene header, queue prefix)", "per batch: ray out of the prefetch registers, next ray located and requested", "per batch: end of the entity loop, result store", "-"]
if os.environ.get("EU_PROFILE_KERNEL") == "shade":      # -DEU_PROFILE_SHADE_WAVE build: sections of eu_wf_shade_kernel
    names = ["(live test)", "ray + hit loaded, hit point, normal, angle (HitCtx)", "surface record, reflection ratio (Fresnel)", "transmission set-up outside the parts below",
             "reflection direction, node record, delivery of a finished colour", "children that only sample the background", "children appended to the next queue", "-",
             "surface colour program", "to_pixel of the surface colour", "threshold direction (Snell: rotation)", "child origin, material_at of an exiting ray", "fused part: trace_closest of a batch of own children", "barrier before the fused part", "fused part: waiting for the last batch",
             "window sort, batch loop, kernel prologue / epilogue"]
    tot = float(sum(ph[:16])) or 1.0
    for i in range(16):
        if ph[i]:
            print("  %-78s %6.2f%%  %10.1f M" % (names[i], 100.0 * ph[i] / tot, ph[i] / 1e6))
    sys.exit(0)
for i in range(11):
    print("  %-78s %6.2f%%  %10.1f M" % (names[i], 100.0 * ph[i] / tot, ph[i] / 1e6))
