"""Sanitizer builds on the CPU (GPU sanitizers are not available on the pool): the product's host side -- JSON loader, flattener,
generator of the scene-specialised kernels, camera arithmetic -- and the oracle's restatement under AddressSanitizer +
UndefinedBehaviorSanitizer (`make -C euclider_amd/csrc asan`, `make -C oracle asan`).  Each runs in a child process with the
sanitizer runtime preloaded: loader tests' inputs, 400 fuzzed scene files, the generator on every shipped scene; the reference's
known answers and a small render on the oracle."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    p = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(p) or not os.path.exists(p):
        pytest.skip("no %s in this image" % name)
    return p


def _run(code, extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    env["LD_PRELOAD"] = _runtime("libasan.so") + ":" + _runtime("libubsan.so")
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    cp = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    report = cp.stdout[-3000:] + cp.stderr[-6000:]
    assert cp.returncode == 0, report
    assert "AddressSanitizer" not in cp.stderr and "runtime error" not in cp.stderr, report
    return cp.stdout


def test_product_host_side_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "euclider_amd", "csrc"), "asan"], stdout=subprocess.DEVNULL)
    code = r'''
import glob, json, os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
sys.argv = ["fuzz_loader.py", "load", "400"]
from euclider_amd import Parser, ParserError
n = 0
for path in sorted(glob.glob("scenes/*.json")):
    for low in (False,):
        env = Parser().parse_file(path)
        src, key = env.jit_source()          # the generator walks the whole flat scene
        assert "eu_jit_fshade" in src and len(key) == 32
        env.close(); n += 1
for bad in ("", "{", "[]", '{"Universe3": []}', '{"Nope": {}}', '{"Universe3": {"camera": 1, "entities": [], "background": 2}}'):
    try:
        Parser().parse(bad).close()
    except ParserError:
        pass
__file__ = os.path.abspath("tools/fuzz_loader.py")
exec(open("tools/fuzz_loader.py").read())
print("scenes", n, "fuzz accepted", acc, "rejected", rej)
'''
    out = _run(code, {"EU_LIB_PATH": os.path.join(ROOT, "euclider_amd", "libeuclider_host_asan.so"), "EU_LIB_HOST_ONLY": "1"})
    assert "scenes 10" in out


def test_oracle_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    code = r'''
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import scene_loader as sl
for scene, depth in (("3d_room.json", 6), ("3d_hallways.json", 8), ("4d_frame.json", 4), ("4d_cylinders.json", 3), ("3d_fresnel_2.json", 6)):
    a, _, sa = sl.load_scene_file(os.path.join("scenes", scene), variant="asan").render(48, 27, max_depth=depth, want_hit_t=True)
    b, _, sb = sl.load_scene_file(os.path.join("scenes", scene)).render(48, 27, max_depth=depth, want_hit_t=True)
    assert np.array_equal(a, b) and sa == sb, scene
from random_scenes import random_scene
from oracle.scene_loader import OracleScene, default_texture_loader, ParserError
n = 0
for seed in range(0, 60):
    text, dim = random_scene(seed)
    try:
        OracleScene(text, default_texture_loader(["."]), variant="asan").render(24, 18, max_depth=4)
        n += 1
    except ParserError:
        pass
print("random scenes rendered", n)
'''
    out = _run(code, {})
    assert "random scenes rendered" in out
