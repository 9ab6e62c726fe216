"""GPU parity tests: the HIP path (through the C ABI) against the oracle on the same inputs.

Bar: RGB8 bit-exact, ray/background counters identical, primary hit distance identical
(tolerance 1e-5 in BASELINE.json; the observed difference is required to be exactly 0 here
unless noted).  Sizes are chosen so the oracle finishes in seconds.
"""
import os
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")

# (scene, width, height, max_depth) -- BASELINE.json configs at reduced size + every other shipped scene
CASES = [
    ("3d_fresnel.json", 256, 256, 4),       # config 0 at full size
    ("3d_room.json", 320, 180, 8),          # config 1 (16:9, reduced)
    ("3d_hallways.json", 320, 180, 12),     # config 2
    ("4d_frame.json", 320, 180, 8),         # config 3
    ("4d_cylinders.json", 160, 90, 8),
    ("3d_frame.json", 160, 90, 10),
    ("3d_fresnel_2.json", 128, 128, 10),
    ("3d_photo.json", 160, 90, 10),
    ("4d_fresnel.json", 128, 128, 10),
    ("4d_room.json", 160, 90, 10),
    ("3d_room.json", 127, 63, 10),          # odd sizes: centre-pixel rays are axis aligned
]


def render_both(scene, w, h, depth, time_ms=0, crosshair=False):
    from euclider_amd import Parser, SimulationContext
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, scene)
    env = Parser().parse_file(path)
    env.camera.max_depth = depth
    img = env.render((w, h), time=time_ms / 1000.0, context=SimulationContext(1, crosshair), want_hit_t=True)
    osc = load_scene_file(path)
    orgb, ohit, ost = osc.render(w, h, max_depth=depth, time_ms=time_ms, want_hit_t=True, debug_crosshair=crosshair)
    env.close()
    return img, orgb, ohit, ost


@pytest.mark.parametrize("scene,w,h,depth", CASES)
def test_scene_parity(scene, w, h, depth):
    img, orgb, ohit, ost = render_both(scene, w, h, depth)
    diff = np.argwhere(img.data != orgb)
    assert diff.size == 0, "%d differing bytes, first at %s: gpu %s oracle %s" % (
        len(diff), diff[0], img.data[tuple(diff[0][:2])], orgb[tuple(diff[0][:2])])
    assert img.stats["rays"] == ost["rays"]
    assert img.stats["bg_samples"] == ost["bg_samples"]
    assert img.stats["nan_pixels"] == ost["nan_pixels"]
    assert img.stats["errors"] == ost["errors"]
    both_nan = np.isnan(img.hit_t) & np.isnan(ohit)
    assert np.all(both_nan | (np.abs(img.hit_t - ohit) <= 1e-5))
    assert np.array_equal(img.hit_t[~both_nan], ohit[~both_nan])


def test_time_and_crosshair():
    img, orgb, ohit, ost = render_both("3d_room.json", 160, 90, 6, time_ms=12345, crosshair=True)
    assert np.array_equal(img.data, orgb)
    assert img.stats["rays"] == ost["rays"]


def test_row_tiles_match_full_frame():
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(SCENES, "3d_room.json"))
    env.camera.max_depth = 6
    full = env.render((160, 90)).data
    parts = [env.render((160, 90), rows=(r0, r1)).data for r0, r1 in [(0, 23), (23, 45), (45, 90)]]
    env.close()
    assert np.array_equal(np.concatenate(parts, axis=0), full)


@pytest.mark.interpreter_only
def test_device_math_matches_oracle(oracle_lib):
    import ctypes as C
    from euclider_amd import _capi
    rng = np.random.default_rng(7)
    n = 1 << 18
    cases = {0: rng.uniform(-1, 1, n), 1: rng.uniform(-1, 1, n), 2: rng.uniform(-20, 20, n), 3: rng.uniform(-20, 20, n),
             4: rng.uniform(-1.5, 1.5, n), 5: rng.uniform(-5, 5, n), 6: rng.uniform(0, 1e6, n) ** 2,
             7: rng.uniform(-1e3, 1e3, n), 8: rng.uniform(-5000, 5000, n)}
    y = rng.uniform(-5, 5, n)
    y8 = np.full(n, 1024.0)
    for fn, x in cases.items():
        x = np.ascontiguousarray(x)
        yy = y8 if fn == 8 else y
        out_d = np.zeros(n)
        out_o = np.zeros(n)
        rc = _capi.lib().eu_selftest_math(0, fn, x.ctypes.data, yy.ctypes.data, out_d.ctypes.data, n)
        assert rc == 0
        oracle_lib.eo_test_math(fn, x.ctypes.data, yy.ctypes.data, out_o.ctypes.data, n)
        assert np.array_equal(out_d.view(np.uint64), out_o.view(np.uint64)), "fn %d differs" % fn


def test_trace_screen_point_unquantised():
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(SCENES, "3d_fresnel.json"))
    img = env.render((64, 64)).data
    for (x, y) in [(0, 0), (31, 33), (63, 5)]:
        rgb = env.trace_screen_point(0.0, env.max_depth(), x, y, 64, 64)
        q = [int(min(max(c, 0.0), 1.0) * 255.0) for c in rgb]
        assert q == list(img[y, x])
    env.close()


@pytest.mark.parametrize("world,H", [(2, 90), (3, 100), (8, 64)])
def test_strip_partition_matches_full_frame(world, H):
    """The multi-GPU partition (8-row strips round-robin, eu_frame.strip_*) reassembles to the full frame."""
    from euclider_amd import Parser, partition
    env = Parser().parse_file(os.path.join(SCENES, "3d_room.json"))
    env.camera.max_depth = 6
    W = 160
    full = env.render((W, H)).data
    max_rows = max(partition.local_rows(H, r, world) for r in range(world))
    bufs = []
    for r in range(world):
        part = env.render((W, H), strips=(r, world)).data
        assert part.shape[0] == partition.local_rows(H, r, world)
        pad = np.zeros((max_rows, W, 3), dtype=np.uint8)
        pad[:part.shape[0]] = part
        bufs.append(pad)
    env.close()
    cat = np.concatenate(bufs, axis=0)
    perm = partition.gather_permutation(H, world, max_rows)
    assert np.array_equal(cat[perm], full)


def test_full_size_properties_config1():
    """BASELINE.json configs[1] at its full size (3d_room, 1920x1080, depth 8), through size-independent properties:
    idempotence, partition invariance (pixels and ray counts), and direct parity on sampled rows."""
    from euclider_amd import Parser
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, "3d_room.json")
    env = Parser().parse_file(path)
    env.camera.max_depth = 8
    W, H = 1920, 1080
    a = env.render((W, H))
    b = env.render((W, H))
    assert np.array_equal(a.data, b.data) and a.stats == b.stats                       # idempotent
    halves = [env.render((W, H), rows=(0, 536)), env.render((W, H), rows=(536, H))]
    assert np.array_equal(np.concatenate([h.data for h in halves], 0), a.data)         # row tiles
    assert sum(h.stats["rays"] for h in halves) == a.stats["rays"]
    strips = [env.render((W, H), strips=(r, 2)) for r in range(2)]
    assert sum(s.stats["rays"] for s in strips) == a.stats["rays"]                      # strip partition: a checksum of checksums
    env.close()
    osc = load_scene_file(path)                                                         # the WHOLE frame against the oracle
    orgb, _, ost = osc.render(W, H, max_depth=8)
    assert int((orgb != a.data).sum()) == 0
    assert a.stats["rays"] == ost["rays"] and a.stats["bg_samples"] == ost["bg_samples"]
    assert a.stats["nan_pixels"] == ost["nan_pixels"] and a.stats["errors"] == ost["errors"]


@pytest.mark.parametrize("scene,depth", [("3d_hallways.json", 12), ("4d_frame.json", 8), ("4d_cylinders.json", 8), ("3d_room.json", 10)])
def test_full_size_whole_frame_other_configs(scene, depth):
    """configs[2] and configs[3] at 1920x1080, the README's "cylinder hypercube" and 3d_room at the reference's own depth
    (d3/entity/camera.rs:50): the whole frame and the counters against the oracle + partition invariance."""
    from euclider_amd import Parser
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, scene)
    env = Parser().parse_file(path)
    env.camera.max_depth = depth
    W, H = 1920, 1080
    a = env.render((W, H))
    parts = [env.render((W, H), rows=(0, 400)), env.render((W, H), rows=(400, H))]
    assert np.array_equal(np.concatenate([p.data for p in parts], 0), a.data)
    assert sum(p.stats["rays"] for p in parts) == a.stats["rays"]
    env.close()
    osc = load_scene_file(path)
    orgb, _, ost = osc.render(W, H, max_depth=depth)
    assert int((orgb != a.data).sum()) == 0
    assert a.stats == {k: ost[k] for k in a.stats}


def test_edge_cases():
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(SCENES, "3d_fresnel.json"))
    assert env.render((16, 16), rows=(5, 5)).data.shape == (0, 16, 3)                   # empty row range
    one = env.render((1, 1))                                                            # 1x1 frame (odd size: axis-aligned ray)
    assert one.data.shape == (1, 1, 3)
    env.camera.max_depth = 0                                                            # depth 0: background only
    bg = env.render((32, 32))
    assert bg.stats["rays"] == 0 and bg.stats["bg_samples"] == 32 * 32
    env.camera.max_depth = 16                                                           # compiled maximum
    deep = env.render((32, 32))
    assert deep.stats["rays"] > 32 * 32
    env.close()


@pytest.mark.interpreter_only
def test_stack_kernel_variant_matches_wavefront():
    """The persistent stack-based kernel (eu_renderer_opts.kernel = EU_KERNEL_STACK; the re-trace path of overflowing frames) and the
    wavefront pipeline give the same frame."""
    from euclider_amd import Parser
    path = os.path.join(SCENES, "3d_room.json")
    a = Parser().parse_file(path)
    a.camera.max_depth = 6
    wf = a.render((200, 120), want_hit_t=True)
    a.close()
    b = Parser().parse_file(path).configure(kernel="stack")
    b.camera.max_depth = 6
    mk = b.render((200, 120), want_hit_t=True)
    b.close()
    assert np.array_equal(wf.data, mk.data)
    assert np.array_equal(wf.hit_t, mk.hit_t, equal_nan=True)
    assert wf.stats["rays"] == mk.stats["rays"] and wf.stats["bg_samples"] == mk.stats["bg_samples"]


def test_banded_wavefront_matches_single_pass():
    """Frames larger than the wavefront band size are traced in several passes; the result must not change."""
    from euclider_amd import Parser
    path = os.path.join(SCENES, "3d_room.json")
    a = Parser().parse_file(path)
    a.camera.max_depth = 6
    one = a.render((320, 200), want_hit_t=True)
    a.close()
    b = Parser().parse_file(path).configure(band_pixels=320 * 24)          # 24-row bands -> 9 passes
    b.camera.max_depth = 6
    many = b.render((320, 200), want_hit_t=True)
    strips = b.render((320, 200), strips=(1, 3))
    b.close()
    assert np.array_equal(one.data, many.data) and np.array_equal(one.hit_t, many.hit_t, equal_nan=True)
    assert one.stats == many.stats
    assert strips.data.shape[0] > 0


@pytest.mark.interpreter_only
def test_shade_kernel_global_scene_variant():
    """Scenes too large for the shade kernel's LDS copy are read from global memory (eu_wf_shade_kernel<D, false>); none of
    the shipped scenes is that large, so the variant is forced here and must give the same frame."""
    from euclider_amd import Parser
    for scene, depth in (("3d_room.json", 6), ("4d_room.json", 5), ("3d_hallways.json", 8)):
        path = os.path.join(SCENES, scene)
        a = Parser().parse_file(path).configure(specialize="off")
        a.camera.max_depth = depth
        lds = a.render((192, 108))
        a.close()
        b = Parser().parse_file(path).configure(specialize="off", shade_scene_global=True)
        b.camera.max_depth = depth
        glb = b.render((192, 108))
        b.close()
        assert np.array_equal(lds.data, glb.data) and lds.stats == glb.stats, scene


@pytest.mark.interpreter_only
def test_error_codes():
    """Bad arguments come back as EU_ERR_* codes, never as a crash (include/euclider_amd.h conventions)."""
    import ctypes as C
    from euclider_amd import Parser, _capi
    L = _capi.lib()
    env3 = Parser().parse_file(os.path.join(SCENES, "3d_fresnel.json"))
    env4 = Parser().parse_file(os.path.join(SCENES, "4d_fresnel.json"))
    r3 = env3.renderer(0)
    out = np.zeros((16, 16, 3), dtype=np.uint8)
    fr = env3.frame(16, 16)
    assert L.eu_render(r3, C.byref(env4.camera), C.byref(fr), out.ctypes.data, None, None) == _capi.EU_ERR_INVALID_ARGUMENT   # 4-D camera, 3-D scene
    cam = _capi.Camera.from_buffer_copy(env3.camera)
    cam.max_depth = 17
    assert L.eu_render(r3, C.byref(cam), C.byref(fr), out.ctypes.data, None, None) == _capi.EU_ERR_CAPACITY                  # deeper than the compiled 16
    bad = env3.frame(16, 16)
    bad.row_end = 17
    assert L.eu_render(r3, C.byref(env3.camera), C.byref(bad), out.ctypes.data, None, None) == _capi.EU_ERR_INVALID_ARGUMENT
    assert L.eu_render(r3, C.byref(env3.camera), C.byref(fr), None, None, None) == _capi.EU_ERR_INVALID_ARGUMENT
    rgb = (C.c_double * 3)()
    assert L.eu_trace_screen_point(r3, C.byref(env3.camera), C.byref(fr), 16, 0, rgb) == _capi.EU_ERR_INVALID_ARGUMENT        # x out of range
    assert L.eu_render(r3, C.byref(env3.camera), C.byref(fr), out.ctypes.data, None, None) == _capi.EU_OK                     # and the renderer still works
    env3.close()
    env4.close()


def test_8k_frame_on_one_gpu_in_bands():
    """BASELINE.json configs[4]'s frame (7680x4320, 3d_room, depth 8) on ONE GPU: traced in bands of <= 4 Mpixel.  Checked
    through row tiles rendered on their own (pixels and ray counts) and sampled rows against the oracle."""
    from euclider_amd import Parser
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, "3d_room.json")
    env = Parser().parse_file(path)
    env.camera.max_depth = 8
    W, H = 7680, 4320
    full = env.render((W, H))
    assert full.data.shape == (H, W, 3)
    tiles = [(0, 8), (2152, 2176), (4312, 4320)]
    for r0, r1 in tiles:
        t = env.render((W, H), rows=(r0, r1))
        assert np.array_equal(t.data, full.data[r0:r1]), (r0, r1)
    halves = [env.render((W, H), rows=(0, 2160)), env.render((W, H), rows=(2160, H))]
    assert sum(h.stats["rays"] for h in halves) == full.stats["rays"]
    env.close()
    osc = load_scene_file(path)                                                         # the whole 8K frame against the oracle (~10 s of CPU)
    orgb, _, ost = osc.render(W, H, max_depth=8)
    assert int((orgb != full.data).sum()) == 0
    assert full.stats["rays"] == ost["rays"] and full.stats["bg_samples"] == ost["bg_samples"]


def test_queue_overflow_falls_back_to_the_stack_kernel():
    """Wavefront pipeline: with queues far too small for the frame (eu_renderer_opts.ray_factor), the asynchronous path reports
    EU_ERR_CAPACITY and the synchronous eu_render still returns the right frame (traced again by the stack-based kernel)."""
    from euclider_amd import Parser, _capi
    path = os.path.join(SCENES, "3d_room.json")
    a = Parser().parse_file(path)
    a.camera.max_depth = 4
    good = a.render((1920, 1080))
    assert a.retraces() == 0
    a.close()
    b = Parser().parse_file(path).configure(kernel="wavefront", ray_factor=0.05, streams=1)
    b.camera.max_depth = 4
    assert b.retraces() == 0
    fell_back = b.render((1920, 1080))
    assert np.array_equal(fell_back.data, good.data) and fell_back.stats == good.stats
    assert b.retraces() == 1          # the slow path is visible to the caller
    multi = b.render_multi((1920, 1080), [0, 0])          # eu_render_multi re-traces an overflowing device's strips the same way
    assert np.array_equal(multi.data, good.data) and multi.stats == good.stats
    from euclider_amd import FrameSequence
    before = b.retraces()
    with FrameSequence(b, (1920, 1080), slots=2) as seq:          # round 4: a frame of a sequence that overflowed is traced again when it is collected
        seq.submit((1920, 1080))
        seq.submit((1920, 1080))
        for _ in range(2):
            img = seq.next()
            assert np.array_equal(img.data, good.data) and img.stats == good.stats
    assert b.retraces() == before + 2
    # the plainly asynchronous call cannot retry: eu_renderer_stats reports the overflow and says what it was
    import torch
    from euclider_amd.environment import EuError
    rgba = torch.zeros((1080, 1920), dtype=torch.int32, device="cuda:0")
    b.render_device(b.frame(1920, 1080, time=0.0, rows=(0, 1080)), rgba.data_ptr(), None, torch.cuda.current_stream().cuda_stream, device=0)
    with pytest.raises(EuError) as ei:
        b.stats(device=0)
    assert ei.value.code == _capi.EU_ERR_CAPACITY and "queue" in str(ei.value)
    b.close()


@pytest.mark.parametrize("ranks,W,H,rows", [(2, 160, 90, None), (3, 200, 100, None), (8, 96, 64, None), (2, 160, 90, (13, 77))])
def test_render_multi_on_one_device(ranks, W, H, rows):
    """eu_render_multi (config 5's path through the C ABI) with several renderers on the ONE device of this box: every
    renderer traces its own strips with the HIP path, the packed strips are gathered into the root's buffer and the row
    order is restored by the kernel -- the product's multi-GPU code, end to end, against the single-renderer frame and
    (for one case) the oracle."""
    from euclider_amd import Parser
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, "3d_room.json")
    env = Parser().parse_file(path)
    env.camera.max_depth = 6
    one = env.render((W, H), rows=rows)
    multi = env.render_multi((W, H), [0] * ranks, rows=rows)
    again = env.render_multi((W, H), [0] * ranks, rows=rows)
    env.close()
    assert np.array_equal(multi.data, one.data) and multi.stats == one.stats
    assert np.array_equal(again.data, one.data)
    if ranks == 2 and rows is None:
        orgb, _, ost = load_scene_file(path).render(W, H, max_depth=6)
        assert np.array_equal(multi.data, orgb) and multi.stats["rays"] == ost["rays"]


@pytest.mark.parametrize("ray_factor", [None, 0.05])
def test_render_multi_two_frames_in_flight(ray_factor):
    """eu_render_multi_begin / _end with three renderers on the one device: frame k's pack, transfer and row restore run on the copy
    streams while frame k + 1 is traced (both sets of buffers in use at once), frames with different cameras, times and row windows
    come back in order and equal the single-renderer frames -- also when every frame overflows its ray queues and is traced again by
    the stack kernel at collection time, behind the next frame's work (ray_factor 0.05)."""
    from euclider_amd import Parser, _capi
    from euclider_amd.environment import EuError
    path = os.path.join(SCENES, "3d_room.json")
    ref = Parser().parse_file(path)
    ref.camera.max_depth = 4
    env = Parser().parse_file(path)
    if ray_factor is not None:
        env.configure(kernel="wavefront", ray_factor=ray_factor, streams=1)
    env.camera.max_depth = 4
    frames = [((320, 180), 0.0, None, 0.0), ((320, 180), 0.5, None, 0.4), ((256, 144), 1.0, (8, 120), -0.3), ((320, 180), 1.5, None, 0.0), ((64, 64), 2.0, None, 0.1)]
    if ray_factor is not None:
        frames = [((1920, 1080), t, r, dx) for (_, t, r, dx) in frames[:3]]
    devs = [0, 0, 0]

    def pose(e, dx):
        e.camera.location[1] = dx

    want = []
    for dims, t, rows, dx in frames:
        pose(ref, dx)
        want.append(ref.render(dims, time=t, rows=rows))
    ref.close()
    got = []
    pose(env, frames[0][3]); env.render_multi_begin(frames[0][0], devs, time=frames[0][1], rows=frames[0][2])
    for dims, t, rows, dx in frames[1:]:
        pose(env, dx); env.render_multi_begin(dims, devs, time=t, rows=rows)      # two in flight
        if len(got) == 0:
            with pytest.raises(EuError) as ei:                                   # a third is refused, and so is the one-call form
                env.render_multi_begin(dims, devs)
            assert ei.value.code == _capi.EU_ERR_INVALID_ARGUMENT
            with pytest.raises(EuError):
                env.render_multi(dims, devs)
        got.append(env.render_multi_end(devs))
    got.append(env.render_multi_end(devs))
    with pytest.raises(EuError):
        env.render_multi_end(devs)
    for w, g in zip(want, got):
        assert g.data.shape == w.data.shape and np.array_equal(g.data, w.data) and g.stats == w.stats
    pose(env, frames[0][3])
    assert np.array_equal(env.render_multi(frames[0][0], devs, time=frames[0][1], rows=frames[0][2]).data, want[0].data)      # the one-call form still works afterwards
    env.close()


@pytest.mark.parametrize("scene,depth,W,H", [("3d_room.json", 8, 1000, 563), ("3d_hallways.json", 12, 960, 540), ("4d_frame.json", 6, 640, 360),
                                             ("4d_cylinders.json", 5, 320, 180), ("3d_photo.json", 6, 333, 187)])
def test_pipeline_forms_agree(scene, depth, W, H):
    """Round 4: how a frame is cut and launched does not show in it.  One to four concurrent band pipelines (interleaved 8-row groups,
    frames large enough to be split), the fused kernel (shade a generation, intersect the rays just queued) against the two-kernel
    pipeline, shade windows spread over the queue, batches and windows dealt through counters: every form gives the oracle's frame,
    counters included."""
    from euclider_amd import Parser, _capi
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, scene)
    orgb, _, ost = load_scene_file(path).render(W, H, max_depth=depth)
    forms = [dict(streams=1), dict(streams=2, split_pixels=4096), dict(streams=3, split_pixels=4096), dict(streams=4, split_pixels=4096),
             dict(streams=1, flags=_capi.EU_RENDERER_NO_FUSE), dict(streams=3, split_pixels=4096, flags=_capi.EU_RENDERER_NO_FUSE),
             dict(streams=2, split_pixels=4096, band_grid_permille=400)]
    from euclider_amd import environment
    if environment.DEFAULT_RENDERER_OPTS.get("specialize") == "sync" and scene in ("3d_room.json", "4d_frame.json"):      # the tuning flags reach the specialised kernels only (a compilation each)
        forms += [dict(streams=2, split_pixels=4096, jit_flags="-DEU_WF_SPREAD=1"), dict(streams=1, jit_flags="-DEU_WF_DEAL_ISECT=1 -DEU_WF_DEAL_SHADE=1 -DEU_WF_WIN_MIN=256"),
                  dict(streams=2, split_pixels=4096, flags=_capi.EU_RENDERER_NO_FUSE, jit_flags="-DEU_WF_DEAL_ISECT=1 -DEU_WF_DEAL_SHADE=1 -DEU_WF_WIN=512"),
                  dict(streams=1, jit_flags="-DEU_WF_EQUAL_WIN=0 -DEU_SHADE_TAKE_CHUNKS=0")]
    for form in forms:
        env = Parser().parse_file(path).configure(**form)
        env.camera.max_depth = depth
        img = env.render((W, H))
        env.close()
        assert np.array_equal(img.data, orgb), form
        assert img.stats["rays"] == ost["rays"] and img.stats["bg_samples"] == ost["bg_samples"], form


def test_renderer_error_text():
    """eu_renderer_error: what the most recent failing call on a renderer had to say."""
    from euclider_amd import Parser, _capi
    env = Parser().parse_file(os.path.join(SCENES, "3d_fresnel.json"))
    L, r = env._L, env.renderer()
    assert (L.eu_renderer_error(r) or b"") == b""
    cam = _capi.Camera.from_buffer_copy(env.camera)
    cam.dim = 4
    fr = env.frame(8, 8, time=0.0, rows=(0, 8))
    out = np.zeros((8, 8, 3), dtype=np.uint8)
    assert L.eu_render(r, C.byref(cam), C.byref(fr), out.ctypes.data, None, None) == _capi.EU_ERR_INVALID_ARGUMENT
    assert b"dimension" in L.eu_renderer_error(r)
    env.close()


def test_band_count_follows_the_frames():
    """A renderer whose caller named no number of bands starts from the scene's flag (4d_cylinders' surfaces reflect: three bands) and
    then goes by the rays per pixel its frames turn out to have (one ray per pixel: one band from the third frame on) -- the switch, the
    buffers cut anew and every frame on either side of it give the oracle's picture."""
    from euclider_amd import Parser
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, "4d_cylinders.json")
    orgb, _, ost = load_scene_file(path).render(1024, 576, max_depth=4)
    env = Parser().parse_file(path)
    env.camera.max_depth = 4
    for k in range(5):
        img = env.render((1024, 576))
        assert np.array_equal(img.data, orgb) and img.stats["rays"] == ost["rays"], k
    env.close()
