"""The C ABI from compiled C code (examples/eu_render.c): builds with gcc against include/euclider_amd.h and the in-tree
library; without a GPU it must fail loudly (no CPU fallback); on the GPU its frames equal the Python mirror's."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "eu_render")
    lib_dir = os.path.join(ROOT, "euclider_amd")
    subprocess.check_call(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "eu_render.c"),
                           "-L" + lib_dir, "-leuclider_amd", "-Wl,-rpath," + lib_dir, "-o", exe])
    return exe


def test_c_example_builds_and_has_no_cpu_fallback(tmp_path):
    from euclider_amd import _capi
    exe = build(tmp_path)
    if _capi.lib().eu_device_count() > 0:
        pytest.skip("a GPU is present: covered by the gpu test")
    p = subprocess.run([exe, os.path.join(ROOT, "scenes", "3d_fresnel.json"), "32", "32", "4", str(tmp_path / "o.ppm")],
                       capture_output=True, text=True)
    assert p.returncode == 1 and "no usable HIP device" in p.stderr


@pytest.mark.gpu
def test_c_example_matches_python_mirror(tmp_path):
    from euclider_amd import FrameSequence, Parser, SimulationContext
    exe = build(tmp_path)
    out = str(tmp_path / "o.ppm")
    frames = 12
    p = subprocess.run([exe, os.path.join(ROOT, "scenes", "3d_hallways.json"), "160", "90", "8", out, str(frames)],
                       capture_output=True, text=True, cwd=ROOT)
    assert p.returncode == 0, p.stderr
    raw = open(out, "rb").read()
    m = re.match(rb"P6\n(\d+) (\d+)\n255\n", raw)
    w, h = int(m.group(1)), int(m.group(2))
    img = np.frombuffer(raw[m.end():], dtype=np.uint8).reshape(h, w, 3)
    # the same walk through the Python mirror; textures: no loader either (procedural grid substituted on both sides)
    from euclider_amd import _capi
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_hallways.json"))
    env.camera.max_depth = 8
    ctx = SimulationContext(pressed_keys=["W"])
    last = None
    with FrameSequence(env, (160, 90), slots=2) as seq:
        for k in range(frames):
            if k > 0:
                env.update(0.016, ctx)
            if seq.in_flight == 2:
                last = seq.next()
            seq.submit((160, 90), time=k * 0.016)
        while seq.in_flight:
            last = seq.next()
    assert np.array_equal(img, last.data)
    cam = [float(x) for x in re.search(r"camera at \((.*)\)", p.stdout).group(1).split(",")]
    assert cam[:3] == list(env.camera.location)[:3]
    env.close()
