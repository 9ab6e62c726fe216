"""Compile the scene-specialised kernels of the shipped scenes into euclider_amd/jit_cache (no GPU needed): the code objects travel
with the library, so the first renderer of such a scene does not wait for hiprtc.  Usage: python tools/jit_precompile.py [scene.json ...]"""
import glob
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = os.path.join(ROOT, "euclider_amd", "jit_cache")


def one(job):
    path, low = job
    from euclider_amd import Parser
    env = Parser(low_precision=low).parse_file(path)
    t = time.time()
    info = env.jit_precompile(CACHE)
    env.close()
    return os.path.basename(path), low, info, time.time() - t


def main(paths=None, workers=None):
    paths = paths or sorted(glob.glob(os.path.join(ROOT, "scenes", "*.json")))
    os.makedirs(CACHE, exist_ok=True)
    jobs = [(p, low) for p in paths for low in (False, True)]
    keep = set()
    with ProcessPoolExecutor(max_workers=workers or min(8, os.cpu_count() or 1)) as ex:
        for name, low, info, dt in ex.map(one, jobs):
            keep.add(info["key"])
            print("%-22s %s  %s  %6.1f s%s" % (name, "f32" if low else "f64", info["key"], dt, "  (cached)" if info["from_cache"] else ""))
    if len(paths) >= 10:      # a full run: code objects of earlier builds of the device headers are of no use any more
        for f in os.listdir(CACHE):
            if f.endswith(".hsaco") and f[:-6].split("_")[0] not in keep and f[:-6] not in keep:
                os.remove(os.path.join(CACHE, f))


if __name__ == "__main__":
    main(sys.argv[1:] or None)
