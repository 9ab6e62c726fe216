"""Scene-specialised kernels on the GPU: what eu_renderer_jit_info reports, the cache, the fall-back when the compilation fails, and
parity of the specialised kernels with the interpreter kernels on inputs the parametrised suite does not reach."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.interpreter_only]      # (chooses its kernel paths itself)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")


def test_jit_info_and_cache(tmp_path):
    from euclider_amd import Parser
    text = open(os.path.join(SCENES, "3d_fresnel.json")).read().replace("1.458", "1.4375")      # a scene no cache knows yet
    cache = str(tmp_path / "cache")
    a = Parser(texture_dirs=[ROOT]).parse(text).configure(specialize="sync", cache_dir=cache)
    img_a = a.render((96, 96))
    info = a.jit_info()
    assert info["requested"] and info["active"] and not info["from_cache"] and info["compile_ms"] > 0
    assert os.listdir(cache) == [info["key"] + ".hsaco"]
    a.close()
    b = Parser(texture_dirs=[ROOT]).parse(text).configure(specialize="sync", cache_dir=cache)
    img_b = b.render((96, 96))
    assert b.jit_info()["from_cache"] and b.jit_info()["active"]
    b.close()
    c = Parser(texture_dirs=[ROOT]).parse(text).configure(specialize="off")
    img_c = c.render((96, 96))
    assert not c.jit_info()["requested"] and not c.jit_info()["active"]
    c.close()
    assert np.array_equal(img_a.data, img_c.data) and np.array_equal(img_b.data, img_c.data) and img_a.stats == img_c.stats


def test_failed_compilation_falls_back_to_the_interpreter(tmp_path):
    from euclider_amd import Parser
    path = os.path.join(SCENES, "3d_room.json")
    good = Parser().parse_file(path).configure(specialize="off")
    good.camera.max_depth = 5
    ref = good.render((160, 90))
    good.close()
    bad = Parser().parse_file(path).configure(specialize="sync", cache_dir=str(tmp_path), jit_flags="-DEU_TRACE_WAVEFRONT_H=1")      # (the kernels' header skips itself: the compilation fails)
    bad.camera.max_depth = 5
    img = bad.render((160, 90))
    info = bad.jit_info()
    bad.close()
    assert info["requested"] and not info["active"]
    assert "error" in info["log"].lower()          # eu_renderer_jit_log: why the interpreter kernels are in charge (the compiler's words)
    assert np.array_equal(img.data, ref.data) and img.stats == ref.stats


def test_opts_struct_of_an_older_caller():
    """struct_size smaller than the library's eu_renderer_opts: the fields beyond it keep their defaults."""
    from euclider_amd import Parser, _capi
    env = Parser().parse_file(os.path.join(SCENES, "3d_fresnel.json"))
    L = _capi.lib()
    opts = _capi.RendererOpts()
    opts.struct_size = 16                      # struct_size, kernel, specialize, streams only
    opts.kernel = _capi.EU_KERNEL_STACK
    opts.ray_factor = -1.0                     # would be refused if it were read
    out, err = C.c_void_p(), C.create_string_buffer(256)
    assert L.eu_renderer_create_opts(env._scene, 0, C.byref(opts), C.byref(out), err, len(err)) == _capi.EU_OK, err.value
    L.eu_renderer_destroy(out)
    opts.struct_size = C.sizeof(_capi.RendererOpts)
    assert L.eu_renderer_create_opts(env._scene, 0, C.byref(opts), C.byref(out), err, len(err)) == _capi.EU_ERR_INVALID_ARGUMENT
    env.close()


@pytest.mark.parametrize("scene,depth", [("3d_room.json", 8), ("3d_hallways.json", 12), ("4d_frame.json", 8)])
def test_sequence_and_trace_screen_point_on_specialised_kernels(scene, depth):
    """The frame sequence (renderer clones per slot) and the single-pixel path with specialised kernels against the interpreter's."""
    from euclider_amd import FrameSequence, Parser
    path = os.path.join(SCENES, scene)
    a = Parser().parse_file(path).configure(specialize="off")
    a.camera.max_depth = depth
    ref = a.render((192, 108))
    pts = [a.trace_screen_point(0.0, depth, x, y, 192, 108) for (x, y) in ((0, 0), (95, 54), (191, 107))]
    a.close()
    b = Parser().parse_file(path).configure(specialize="sync")
    b.camera.max_depth = depth
    assert b.jit_info()["active"], "the specialised kernels did not build: this test would compare the interpreter with itself"
    with FrameSequence(b, (192, 108), slots=3) as seq:
        for _ in range(3):
            seq.submit((192, 108))
        frames = [seq.next() for _ in range(3)]
    assert all(np.array_equal(f.data, ref.data) and f.stats == ref.stats for f in frames)
    assert [b.trace_screen_point(0.0, depth, x, y, 192, 108) for (x, y) in ((0, 0), (95, 54), (191, 107))] == pts
    b.close()


@pytest.mark.parametrize("scene,depth,flags", [
    ("3d_room.json", 8, "-DEU_JIT_OPS_BUDGET=3 -DEU_JIT_SURFACES_BUDGET=2"),          # entities 0-2 with code of their own, 3-7 from the flat scene; two surfaces of five
    ("3d_hallways.json", 12, "-DEU_JIT_OPS_BUDGET=0 -DEU_JIT_SURFACES_BUDGET=0"),     # nothing but the kernels' structure is specialised (LinearSpace expressions stay arithmetic)
    ("4d_cylinders.json", 5, "-DEU_JIT_OPS_BUDGET=80 -DEU_JIT_SURFACES_BUDGET=4"),    # 15 + 29 + 29 operations fit; the private hit stack
    ("3d_room.json", 8, "-DEU_JIT_OPS_BUDGET=256 -DEU_JIT_SURFACES_BUDGET=1"),        # every shape, one surface
])
def test_mixed_kernels_of_scenes_beyond_the_budgets(scene, depth, flags, tmp_path):
    """jit.hpp's budgets, forced small: entities and surfaces without code of their own are traced and shaded by the interpreter's routines
    inside the specialised kernels.  Frames, counters and single pixels equal the interpreter kernels'."""
    from euclider_amd import Parser
    path = os.path.join(SCENES, scene)
    a = Parser().parse_file(path).configure(specialize="off")
    a.camera.max_depth = depth
    ref = a.render((256, 144), time=0.75)
    pts = [a.trace_screen_point(0.75, depth, x, y, 256, 144) for (x, y) in ((3, 3), (128, 72), (200, 40))]
    a.close()
    for rflags in ((0, 2) if "BUDGET=3 " in flags else (0,)):          # the fused kernels and, once, the two-kernel pipeline (EU_RENDERER_NO_FUSE)
        b = Parser().parse_file(path).configure(specialize="sync", jit_flags=flags, cache_dir=str(tmp_path), flags=rflags)
        b.camera.max_depth = depth
        img = b.render((256, 144), time=0.75)
        info = b.jit_info()
        assert info["active"] and not info["from_cache"], info
        assert np.array_equal(img.data, ref.data) and img.stats == ref.stats
        assert [b.trace_screen_point(0.75, depth, x, y, 256, 144) for (x, y) in ((3, 3), (128, 72), (200, 40))] == pts
        b.close()


def test_asynchronous_specialisation(tmp_path):
    """EU_SPECIALIZE_ASYNC: the renderer is usable at once (interpreter kernels), switches to the specialised kernels at a frame boundary
    when the worker thread's compilation is done, and every frame on either side of the switch is the same."""
    import time
    from euclider_amd import Parser
    text = open(os.path.join(SCENES, "3d_fresnel_2.json")).read().replace("1.458", "1.4453125")      # a scene no cache knows yet
    cache = str(tmp_path / "cache")
    ref_env = Parser(texture_dirs=[ROOT]).parse(text).configure(specialize="off")
    ref = ref_env.render((128, 128))
    ref_env.close()
    env = Parser(texture_dirs=[ROOT]).parse(text).configure(specialize="async", cache_dir=cache)
    t0 = time.time()
    first = env.render((128, 128))
    created_and_first_frame_s = time.time() - t0
    info = env.jit_info()
    assert info["requested"]
    frames_before = frames_after = 0
    while time.time() - t0 < 180:
        img = env.render((128, 128))
        assert np.array_equal(img.data, ref.data) and img.stats == ref.stats
        if env.jit_info()["active"]:
            frames_after += 1
            if frames_after >= 3:
                break
        else:
            frames_before += 1
            time.sleep(0.05)
    assert np.array_equal(first.data, ref.data)
    assert env.jit_info()["active"], "the worker thread's compilation never arrived"
    assert frames_before >= 1 or created_and_first_frame_s < 1.0      # the compilation did not block the first frames
    env.close()
    again = Parser(texture_dirs=[ROOT]).parse(text).configure(specialize="async", cache_dir=cache)      # now cached: specialised from the first frame on
    img = again.render((128, 128))
    assert again.jit_info()["active"] and again.jit_info()["from_cache"] and np.array_equal(img.data, ref.data)
    again.close()
    gone = Parser(texture_dirs=[ROOT]).parse(text.replace("1.4453125", "1.44921875")).configure(specialize="async", cache_dir=cache)
    gone.render((64, 64))
    gone.close()          # destroyed while its compilation is queued or running: nothing may crash, here or at interpreter exit


@pytest.mark.parametrize("scene,depth", [("4d_frame.json", 6), ("3d_room.json", 5)])
def test_full_hit_stack_retraces_the_frame(scene, depth, tmp_path):
    """The wavefront kernels reserve two hit-stack entries for an Intersection chain (a line meets a convex solid's boundary twice);
    should rounding noise ever let a third hit through, the lane reports a full stack, the frame counts as overflowed and the stack
    kernel -- whose stack has the strict size -- traces it again.  -DEU_TEST_HS_FULL makes every box that is hit report that:
    the frame must still be the interpreter's, and the slow path must be visible (eu_renderer_retraces)."""
    from euclider_amd import Parser
    path = os.path.join(SCENES, scene)
    def place(env):      # (4d_frame's own camera sits inside all of its boxes: one hit each)
        env.camera.max_depth = depth
        if scene.startswith("4d"):
            for k, x in enumerate((-10.0, 0.5, 0.25, 0.0)):
                env.camera.location[k] = x
    good = Parser().parse_file(path).configure(specialize="off")
    place(good)
    ref = good.render((160, 90), want_hit_t=True)
    assert good.retraces() == 0
    good.close()
    env = Parser().parse_file(path).configure(specialize="sync", cache_dir=str(tmp_path), jit_flags="-DEU_TEST_HS_FULL")
    place(env)
    img = env.render((160, 90), want_hit_t=True)
    assert env.jit_info()["active"] and env.retraces() == 1
    env.close()
    assert np.array_equal(img.data, ref.data) and img.stats == ref.stats
    both_nan = np.isnan(img.hit_t) & np.isnan(ref.hit_t)
    assert np.array_equal(img.hit_t[~both_nan], ref.hit_t[~both_nan])


def test_full_hit_stack_on_the_sequence_and_single_pixel_paths(tmp_path):
    """Round 4 (ADVICE): a full hit stack has its own counter (EuDevCounters::hs_full) and every entry point that hands out pixels
    falls back to the stack kernel by itself: eu_sequence_next traces the frame again when it is collected, eu_trace_screen_point
    traces the pixel again -- neither returns a colour made of a dropped hit -- and the plainly asynchronous eu_render_device reports
    what happened (EU_ERR_CAPACITY with a text that names the hit stack, not the ray queues)."""
    import torch
    from euclider_amd import FrameSequence, Parser, _capi
    from euclider_amd.environment import EuError
    path = os.path.join(SCENES, "3d_room.json")
    good = Parser().parse_file(path).configure(specialize="off")
    good.camera.max_depth = 5
    ref = good.render((160, 90))
    grid = [(x, y) for y in range(5, 90, 10) for x in range(5, 160, 10)]          # (some of these pixels look at the glass block: a box inside a CSG tree)
    pts = [good.trace_screen_point(0.0, 5, x, y, 160, 90) for (x, y) in grid]
    good.close()
    env = Parser().parse_file(path).configure(specialize="sync", cache_dir=str(tmp_path), jit_flags="-DEU_TEST_HS_FULL")
    env.camera.max_depth = 5
    assert env.jit_info()["active"]
    with FrameSequence(env, (160, 90), slots=2) as seq:
        seq.submit((160, 90))
        seq.submit((160, 90))
        frames = [seq.next(), seq.next()]
    assert all(np.array_equal(f.data, ref.data) and f.stats == ref.stats for f in frames)
    assert env.retraces() == 2
    assert [env.trace_screen_point(0.0, 5, x, y, 160, 90) for (x, y) in grid] == pts
    assert env.retraces() > 2           # (a pixel whose rays meet no such box needs no second trace; those that do got one)
    rgba = torch.zeros((90, 160), dtype=torch.int32, device="cuda:0")
    env.render_device(env.frame(160, 90, time=0.0, rows=(0, 90)), rgba.data_ptr(), None, torch.cuda.current_stream().cuda_stream, device=0)
    with pytest.raises(EuError) as ei:
        env.stats(device=0)
    assert ei.value.code == _capi.EU_ERR_CAPACITY and "hit stack" in str(ei.value)
    env.close()
