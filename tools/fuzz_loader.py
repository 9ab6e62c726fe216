"""Diagnostic: mutate random scene files and feed them to the C++ loader (crash safety); with `render`, also trace the accepted
ones on the GPU and compare with the oracle where it accepts them too.  Usage: python tools/fuzz_loader.py [render] [n]"""
import sys, os, json, random, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from random_scenes import random_scene
from euclider_amd import Parser, ParserError
rng = random.Random(1)
def mutate(o, depth=0):
    if isinstance(o, dict):
        o = dict(o)
        if o and rng.random() < 0.01:
            k = rng.choice(list(o.keys()))
            choice = rng.random()
            if choice < 0.3: o.pop(k)
            elif choice < 0.6: o[k + rng.choice(["x", "::new", "_3", ""])] = o.pop(k)
            else: o[k] = rng.choice([None, 1, "str", [], {}, [1, 2, 3], True, -1e308, float("nan") if False else 1e400])
        return {k: mutate(v, depth + 1) for k, v in o.items()}
    if isinstance(o, list):
        o = [mutate(v, depth + 1) for v in o]
        if o and rng.random() < 0.01:
            c = rng.random()
            if c < 0.4: o.pop(rng.randrange(len(o)))
            elif c < 0.7: o.append(rng.choice([0, "x", {}, []]))
            else: o.insert(0, o[-1])
        return o
    if isinstance(o, (int, float)) and rng.random() < 0.004:
        return rng.choice([0, -1, 1e300, -1e-300, "nan", None, 2**70])
    if isinstance(o, str) and rng.random() < 0.004:
        return rng.choice(["", "Union", "x * ", "((", "sqrt(", "./nope.png", "xyzw", "é"])
    return o
acc = rej = 0
RENDER = len(sys.argv) > 1 and sys.argv[1] == "render"
only_product = render_fail = compared = 0
mismatch = []
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3000):
    if it % 200 == 0: print("iteration", it, flush=True)
    text, dim = random_scene(rng.randrange(10**6))
    obj = mutate(json.loads(text))
    try:
        t = json.dumps(obj)
    except Exception:
        continue
    if rng.random() < 0.05:      # raw text damage
        p = rng.randrange(len(t)); t = t[:p] + rng.choice(["", "}", "[", "\"", ",", "\\", "\x00"]) + t[p + rng.randrange(3):]
    try:
        env = Parser(texture_dirs=[ROOT]).parse(t)
        acc += 1
        if RENDER:
            import numpy as np
            from oracle.scene_loader import OracleScene, default_texture_loader
            try:
                osc = OracleScene(t, default_texture_loader([ROOT]))
            except Exception:
                osc = None; only_product += 1
            env.camera.max_depth = 4
            try:
                img = env.render((16, 12))
            except Exception as e:
                img = None; render_fail += 1
            if osc is not None and img is not None:
                orgb, _, ost = osc.render(16, 12, max_depth=4, threads=4)
                if not osc.last_spins:
                    compared += 1
                    if not np.array_equal(img.data, orgb) or img.stats != ost:
                        mismatch.append(it)
        env.close()
    except ParserError:
        rej += 1
    except ValueError as e:      # embedded NUL etc. rejected by ctypes before the call
        rej += 1
print("accepted", acc, "rejected", rej, "only-product", only_product, "render-fail", render_fail, "compared", compared, "mismatch", mismatch[:10])
