import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from random_scenes import random_scene
from euclider_amd import Parser
from euclider_amd.environment import EuError
from oracle.scene_loader import OracleScene, default_texture_loader
ROOT = os.getcwd()
lo, hi = int(sys.argv[1]), int(sys.argv[2])
W, H, DEPTH = int(os.environ.get("HUNT_W", 40)), int(os.environ.get("HUNT_H", 30)), int(os.environ.get("HUNT_DEPTH", 5))
F32 = os.environ.get("HUNT_F32") == "1"      # the low_precision pair: libeuclider_amd_f32.so against libeo_oracle_f32.so
SPEC = os.environ.get("HUNT_SPECIALIZE", "off")      # "sync": kernels specialised for every scene (a hiprtc compilation each)
STREAMS = int(os.environ.get("HUNT_STREAMS", "0"))    # > 1: the frame is cut into that many concurrent band pipelines (interleaved 8-row groups) however small it is
RFLAGS = int(os.environ.get("HUNT_FLAGS", "0"))       # eu_renderer_opts.flags (2: the two-kernel pipeline instead of the fused kernels)
MIXED = os.environ.get("HUNT_MIXED") == "1"            # specialised kernels with budgets of half the scene's shape operations and one or two surfaces: the rest is traced from the flat scene inside them (jit.hpp)
not_specialised = 0
bad = []; undefined = 0; skipped = 0
for seed in range(lo, hi):
    if seed % (10 if SPEC == "sync" else 200) == 0: print("at seed", seed, "bad so far", len(bad), flush=True)
    text, dim = random_scene(seed)
    try:
        osc = OracleScene(text, default_texture_loader([ROOT]), variant="f32" if F32 else "")
        orgb, ohit, ost = osc.render(W, H, max_depth=DEPTH, time_ms=100 * (seed % 7), want_hit_t=True, threads=8)
        env = Parser(texture_dirs=[ROOT], low_precision=F32).parse(text)
        jf = ("-DEU_JIT_OPS_BUDGET=%d -DEU_JIT_SURFACES_BUDGET=%d" % (max(1, env.info.n_shape_ops // 2), 1 + seed % 2)) if MIXED else None
        env.configure(specialize=SPEC, streams=STREAMS, split_pixels=64 if STREAMS > 1 else 0, flags=RFLAGS, jit_flags=jf)
    except Exception as e:
        skipped += 1; continue
    env.camera.max_depth = DEPTH
    try:
        img = env.render((W, H), time=(100 * (seed % 7)) / 1000.0, want_hit_t=True)
    except EuError as e:
        env.close(); skipped += 1; continue
    if SPEC == "sync" and not env.jit_info()["active"]: not_specialised += 1
    env.close()
    if osc.last_spins:
        undefined += 1
        continue
    if not np.array_equal(img.data, orgb) or img.stats != ost:
        bad.append((seed, int((img.data != orgb).sum()), img.stats, ost))
        continue
    gh = img.hit_t.astype(ohit.dtype)
    nn = ~(np.isnan(gh) & np.isnan(ohit))
    if not np.array_equal(gh[nn], ohit[nn]): bad.append((seed, "hit_t"))
print("streams", STREAMS or "default", "flags", RFLAGS, "specialize", SPEC + (" (mixed kernels)" if MIXED else ""), "(scenes left to the interpreter kernels: %d)" % not_specialised, "f32" if F32 else "f64", "%dx%d depth %d" % (W, H, DEPTH), "seeds", lo, hi, "bad", len(bad), "undefined", undefined, "skipped", skipped)
for b in bad[:20]: print(b)
