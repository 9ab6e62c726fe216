"""Differential test on random scenes: GPU (through the C ABI) == oracle, bit for bit, on CSG / material / surface
combinations the shipped scenes do not contain (SymmetricDifference, nested Complements, cylinders with height, hyperplanes
from vectors, LinearSpace around glass, every blend function, nearest-neighbour textures, camera inside solids, ...)."""
import os

import numpy as np
import pytest

from random_scenes import random_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_path_is_jit():
    from euclider_amd import environment
    return environment.DEFAULT_RENDERER_OPTS.get("specialize") == "sync"


def run_case(seed, w, h, depth, time_ms=0, n_entities=None, min_flat_bytes=0, low_precision=False):
    from euclider_amd import Parser
    from euclider_amd.environment import EuError
    from oracle.scene_loader import OracleScene, default_texture_loader
    text, dim = random_scene(seed, n_entities=n_entities)
    from oracle.scene_loader import ParserError as OracleParserError
    from euclider_amd import ParserError
    try:
        osc = OracleScene(text, default_texture_loader([ROOT]), variant="f32" if low_precision else "")
    except OracleParserError:      # a degenerate draw (e.g. a zero normal): both loaders must refuse it
        with pytest.raises(ParserError):
            Parser(texture_dirs=[ROOT], low_precision=low_precision).parse(text).close()
        pytest.skip("scene %d is refused by both loaders" % seed)
    orgb, ohit, ost = osc.render(w, h, max_depth=depth, time_ms=time_ms, want_hit_t=True)
    try:
        env = Parser(texture_dirs=[ROOT], low_precision=low_precision).parse(text)
    except Exception as e:          # a capacity the kernels were compiled for (reported, never silent)
        pytest.skip("scene %d rejected by the product loader: %s" % (seed, e))
    assert env.info.flat_bytes >= min_flat_bytes
    env.camera.max_depth = depth
    mixed = None
    if kernel_path_is_jit():
        # the generator's budgets (jit.hpp): a scene beyond them gets mixed kernels -- here with small budgets, the straight-line code of 256
        # random shape operations takes minutes to compile; one in 32 of the other scenes is put on the mixed path too (a mixed module
        # holds the interpreter's routines as well: ~25 s of hiprtc each)
        if env.info.n_shape_ops > 256 or env.info.n_entities > 48:
            mixed = "-DEU_JIT_OPS_BUDGET=24 -DEU_JIT_SURFACES_BUDGET=6"
        elif seed % 32 == 4:
            mixed = "-DEU_JIT_OPS_BUDGET=%d -DEU_JIT_SURFACES_BUDGET=%d" % (max(1, env.info.n_shape_ops // 2), 1 + seed // 32 % 2)
        if mixed:
            env.configure(jit_flags=mixed)
    try:
        img = env.render((w, h), time=time_ms / 1000.0, want_hit_t=True)
    except EuError as e:
        env.close()
        if e.code == -5:
            pytest.skip("scene %d exceeds a compiled capacity" % seed)
        raise
    if kernel_path_is_jit():
        assert env.jit_info()["active"], "specialised kernels of scene %d (%s): %s" % (seed, mixed, env.jit_info()["log"][:2000])
    env.close()
    if osc.last_spins:
        # a CSG stream the reference never finishes computing (shape.rs:390-392 under an outer operation that keeps asking, e.g.
        # Intersection(Complement(box, missed cylinder), X) takes the Complement's never-advancing element for ever): its render
        # hangs, so no frame is defined.  (The GPU either flags the entity in eu_stats.errors or, when the ray misses the entity's
        # bounding sphere, never evaluates it.)
        return
    diff = np.argwhere(img.data != orgb)
    assert diff.size == 0, "seed %d: %d differing bytes, first at %s: gpu %s oracle %s" % (
        seed, len(diff), diff[0], img.data[tuple(diff[0][:2])], orgb[tuple(diff[0][:2])])
    assert img.stats == ost, (seed, img.stats, ost)
    gh = img.hit_t.astype(np.float32) if low_precision else img.hit_t
    both_nan = np.isnan(gh) & np.isnan(ohit)
    assert np.array_equal(gh[~both_nan], ohit[~both_nan]), seed


@pytest.mark.interpreter_only
@pytest.mark.parametrize("chunk", range(20))
def test_random_scene_hunt_f64(chunk):
    """2 000 random scenes per run (seeds 40000 + 100 * chunk ...), small frames: the hunt that used to live in tools/ only."""
    for seed in range(40000 + 100 * chunk, 40000 + 100 * (chunk + 1)):
        try:
            run_case(seed, 40, 30, 5, time_ms=250 * (seed % 5))
        except pytest.skip.Exception:
            continue


@pytest.mark.interpreter_only
@pytest.mark.parametrize("chunk", range(6))
def test_random_scene_hunt_f32(chunk):
    """600 more on the F = f32 pair (libeuclider_amd_f32.so against libeo_oracle_f32.so)."""
    for seed in range(60000 + 100 * chunk, 60000 + 100 * (chunk + 1)):
        try:
            run_case(seed, 40, 30, 5, low_precision=True)
        except pytest.skip.Exception:
            continue


@pytest.mark.parametrize("seed", range(0, 120))
def test_random_scene_parity(seed):
    run_case(seed, 48, 36, 5, time_ms=250 * (seed % 5))


@pytest.mark.parametrize("seed", range(20000, 20040))
def test_random_scene_parity_general_expressions(seed):
    """Seeds >= 20000 also draw LinearSpace expressions with functions, powers, remainders and several variables."""
    run_case(seed, 48, 36, 5)


@pytest.mark.parametrize("seed", range(1000, 1012))
def test_random_scene_parity_deeper_and_larger(seed):
    run_case(seed, 160, 90, 8)


@pytest.mark.interpreter_only
@pytest.mark.parametrize("seed", range(300, 340))
def test_random_scene_trace_path(seed):
    """Camera motion (Universe::trace_path_unknown) through random scenes: GPU == oracle, bit for bit."""
    import random
    from euclider_amd import Parser
    from euclider_amd.environment import EuError
    from oracle.scene_loader import OracleScene, default_texture_loader
    text, dim = random_scene(seed)
    osc = OracleScene(text, default_texture_loader([ROOT]))
    try:
        env = Parser(texture_dirs=[ROOT]).parse(text)
    except Exception as e:
        pytest.skip("scene %d rejected by the product loader: %s" % (seed, e))
    rng = random.Random(seed)
    base = list(env.camera.location)[:dim]
    for k in range(12):
        loc = [b + rng.uniform(-4.0, 14.0 if i == 0 else 4.0) for i, b in enumerate(base)]
        d = [rng.gauss(0.0, 1.0) for _ in range(dim)]
        n = sum(x * x for x in d) ** 0.5
        d = [x / n for x in d]
        dist = rng.choice([0.5, 3.0, 12.0, 40.0])
        try:
            o = osc.trace_path_unknown(dist, loc, d)
        except RuntimeError:
            continue                        # step cap in the oracle: the reference would overflow its stack
        try:
            g = env.trace_path_unknown(dist, loc, d)
        except EuError as e:
            assert e.code in (-5, -8), (seed, k, e.code)
            continue
        assert (g is None) == (o is None), (seed, k)
        if g is not None:
            assert g == o, (seed, k, loc, d, dist, g, o)
    env.close()


@pytest.mark.parametrize("seed,n", [(500, 60), (501, 150), (504, 300), (505, 300)])
def test_random_scene_parity_many_entities(seed, n):
    """Scenes far larger than the shipped ones: the flat scene no longer fits the shade kernel's LDS copy (global variant), the
    sort keys of entities >= 31 share a bucket, hundreds of bounds and materials."""
    run_case(seed, 64, 48, 4, n_entities=n, min_flat_bytes=45 * 1024 if n >= 150 else 0)


def test_guarded_subtrees_parity():
    """Capped cylinders and spheres far apart inside Unions and a Complement: the loader guards the bounded subtrees (EU_SH_SKIP,
    tests/test_loader.py::test_bounded_subtrees_get_guard_ops), waves whose rays all miss a guard's sphere jump over the subtree, and
    containment tests of points outside it are answered by the sphere.  Reflective surfaces send the rays round the scene, the
    camera stands between the solids.  Must equal the oracle, which knows nothing of guards."""
    import json
    from euclider_amd import Parser
    from oracle.scene_loader import OracleScene, default_texture_loader

    def cyl(c, d, h=4):
        return {"Cylinder3::new_with_height": [{"Point3::new": c}, {"Vector3::new": d}, 0.5, h]}

    def sph(c, r=1.0):
        return {"Sphere3::new": [{"Point3::new": c}, r]}

    def of(shapes, op):
        return {"ComposableShape3::of": [shapes, {"SetOperation": [op]}]}

    def entity(shape, ratio):
        return {"Entity3Impl::new_with_surface": [shape, {"Vacuum3::new": []}, {"ComposableSurface3": {
            "reflection_ratio": {"reflection_ratio_uniform_3": [ratio]},
            "reflection_direction": {"reflection_direction_specular_3": []},
            "threshold_direction": {"threshold_direction_identity_3": []},
            "surface_color": {"surface_color_illumination_directional_3": [{"Vector3::new": [0.3, -0.5, -1]}, {"Rgba::new": [1, 0.8, 0.6, 1]}, {"Rgba::new": [0.1, 0.1, 0.3, 1]}]}}}]}

    shapes = [
        of([cyl([6, 3, 0], [1, 0, 0]), cyl([6, -3, 1], [0, 0, 1]), sph([9, 0, -2]), cyl([5, 0, 3], [0, 1, 0], 6)], "Union"),
        of([of([sph([-6, 2, 0], 2.0), sph([-6, -3, 0], 1.5)], "Union"), of([cyl([-6, 2, 0], [0, 0, 1], 6), sph([-6, -3, 1], 1.0)], "Union")], "Complement"),
        of([of([sph([0, 7, 0], 1.5), cyl([0, 7, 0], [1, 1, 0], 5)], "SymmetricDifference"), sph([0, -7, 0], 2.0)], "Union"),
    ]
    text = json.dumps({"Universe3": {"camera": {"FreeCamera3": []},
                                     "entities": [entity(s, r) for s, r in zip(shapes, (0.5, 0.3, 0.0))] + [{"Void3::new_with_vacuum": []}],
                                     "background": {"MappedTextureImpl3::new": [{"uv_sphere_3": [{"Point3::new": [0, 0, 0]}]},
                                                                               {"texture_image_nearest_neighbor": ["./resources/simple.png"]}]}}})
    env = Parser(texture_dirs=[ROOT]).parse(text)
    assert env.info.n_shape_ops > 30          # guards included
    env.camera.max_depth = 6
    osc = OracleScene(text, default_texture_loader([ROOT]))
    for fwd in ([1.0, 0.0, 0.0], [-1.0, 0.2, 0.1], [0.1, 1.0, 0.0]):
        import math
        n = math.sqrt(sum(x * x for x in fwd))
        for k in range(3):
            env.camera.forward[k] = fwd[k] / n
        # left = up x forward with up = z, kept orthonormal enough for a picture; the oracle gets the same pose
        left = [-env.camera.forward[1], env.camera.forward[0], 0.0]
        ln = math.sqrt(sum(x * x for x in left)) or 1.0
        for k in range(3):
            env.camera.left[k] = left[k] / ln
        up = [env.camera.forward[1] * env.camera.left[2] - env.camera.forward[2] * env.camera.left[1],
              env.camera.forward[2] * env.camera.left[0] - env.camera.forward[0] * env.camera.left[2],
              env.camera.forward[0] * env.camera.left[1] - env.camera.forward[1] * env.camera.left[0]]
        for k in range(3):
            env.camera.up[k] = up[k]
        ocam = osc.camera()
        for fld in ("location", "forward", "up", "left"):
            for k in range(3):
                getattr(ocam, fld)[k] = getattr(env.camera, fld)[k]
        img = env.render((160, 90), want_hit_t=True)
        orgb, ohit, ost = osc.render(160, 90, max_depth=6, want_hit_t=True, camera=ocam)
        assert np.array_equal(img.data, orgb) and img.stats == ost
        both_nan = np.isnan(img.hit_t) & np.isnan(ohit)
        assert np.array_equal(img.hit_t[~both_nan], ohit[~both_nan])
    env.close()


@pytest.mark.parametrize("inner,outer", [("Complement", "Union"), ("Intersection", "SymmetricDifference"), ("SymmetricDifference", "Complement")])
def test_congruent_entities_parity(inner, outer):
    """A row of seven entities with the same shape program (a carved box with a capped cylinder and a sphere) at different places, with three
    kinds of surface and a different solid between them: the generator writes ONE body for the first five and ONE for the last two, each in a
    loop over its entities with a stride on every parameter address (jit.cpp: find_runs) -- hits on the boxes' faces take their normals
    from memory there, not from a table of constants.  Must equal the oracle, which knows nothing of runs, from a camera that looks along
    the row and from one inside the third box."""
    import json
    from euclider_amd import Parser
    from oracle.scene_loader import OracleScene, default_texture_loader

    def of(shapes, op):
        return {"ComposableShape3::of": [shapes, {"SetOperation": [op]}]}

    def column(x, y):
        box = {"HalfSpace3::cuboid": [{"Point3::new": [x, y, 0.25]}, {"Vector3::new": [2.5, 2, 3]}]}
        hole = {"Sphere3::new": [{"Point3::new": [x, y + 0.5, 0.5]}, 1.1]}
        post = {"Cylinder3::new_with_height": [{"Point3::new": [x + 0.25, y, 0]}, {"Vector3::new": [0.1, 0.2, 1]}, 0.4, 5]}
        return of([of([box, hole], inner), post], outer)

    def entity(shape, kind):
        surf = [
            {"reflection_ratio": {"reflection_ratio_fresnel_3": [1.458, 1]}, "reflection_direction": {"reflection_direction_specular_3": []},
             "threshold_direction": {"threshold_direction_snell_3": [1.458]}, "surface_color": {"surface_color_uniform_3": [{"Rgba::new": [0.1, 0.3, 0.2, 0.25]}]}},
            {"reflection_ratio": {"reflection_ratio_uniform_3": [0.4]}, "reflection_direction": {"reflection_direction_specular_3": []},
             "threshold_direction": {"threshold_direction_identity_3": []},
             "surface_color": {"surface_color_illumination_global_3": [{"Rgba::new": [1, 0.9, 0.5, 1]}, {"Rgba::new": [0.1, 0, 0.2, 1]}]}},
            {"reflection_ratio": {"reflection_ratio_uniform_3": [0]}, "reflection_direction": {"reflection_direction_specular_3": []},
             "threshold_direction": {"threshold_direction_identity_3": []},
             "surface_color": {"surface_color_illumination_directional_3": [{"Vector3::new": [0.3, -0.5, -1]}, {"Rgba::new": [0.5, 0.8, 1, 1]}, {"Rgba::new": [0.2, 0.1, 0.1, 1]}]}},
        ][kind]
        return {"Entity3Impl::new_with_surface": [shape, {"Vacuum3::new": []}, {"ComposableSurface3": surf}]}

    cols = [column(6.0 + 4.5 * k, -6.0 + 2.25 * k) for k in range(7)]
    ents = [entity(c, k % 3) for k, c in enumerate(cols[:5])]
    ents.append(entity({"Sphere3::new": [{"Point3::new": [14, 9, 2]}, 2.0]}, 1))
    ents += [entity(c, (k + 1) % 3) for k, c in enumerate(cols[5:])]
    text = json.dumps({"Universe3": {"camera": {"FreeCamera3": []}, "entities": ents + [{"Void3::new_with_vacuum": []}],
                                     "background": {"MappedTextureImpl3::new": [{"uv_sphere_3": [{"Point3::new": [0, 0, 0]}]},
                                                                               {"texture_image_nearest_neighbor": ["./resources/simple.png"]}]}}})
    env = Parser(texture_dirs=[ROOT]).parse(text)
    src, _ = env.jit_source()
    assert "ge < 5u" in src and "ge < 2u" in src and src.count("/* entity ") == 1          # two loops and the sphere between them
    env.camera.max_depth = 6
    osc = OracleScene(text, default_texture_loader([ROOT]))
    for loc in ([0.0, 0.0, 0.0], [15.0, -1.6, 0.4]):
        ocam = osc.camera()
        for k in range(3):
            env.camera.location[k] = loc[k]
            ocam.location[k] = loc[k]
        img = env.render((200, 112), want_hit_t=True)
        orgb, ohit, ost = osc.render(200, 112, max_depth=6, want_hit_t=True, camera=ocam)
        assert np.array_equal(img.data, orgb) and img.stats == ost
        both_nan = np.isnan(img.hit_t) & np.isnan(ohit)
        assert np.array_equal(img.hit_t[~both_nan], ohit[~both_nan])
    if kernel_path_is_jit():
        assert env.jit_info()["active"]
    env.close()


@pytest.mark.parametrize("low_precision", [False, True])
def test_box_with_faces_through_coordinate_planes(low_precision):
    """A cuboid with faces in the planes x = 0, y = 0, z = 0 (half-space constants +-0: EU_SH_CHAIN_BOX0) as glass, alone and inside a
    Complement; camera origins with coordinates that are exactly +0 and -0 (the one case where the device's one-product form of the
    dot product would differ from the reference's sum, so those waves go the generic way).  Must equal the oracle bit for bit."""
    import json
    import math
    from euclider_amd import Parser
    from oracle.scene_loader import OracleScene, default_texture_loader

    def glass(shape):
        return {"Entity3Impl::new_with_surface": [shape, {"Vacuum3::new": []}, {"ComposableSurface3": {
            "reflection_ratio": {"reflection_ratio_fresnel_3": [1.458, 1]},
            "reflection_direction": {"reflection_direction_specular_3": []},
            "threshold_direction": {"threshold_direction_snell_3": [1.458]},
            "surface_color": {"surface_color_uniform_3": [{"Rgba::new": [0.2, 0.4, 0.1, 0.3]}]}}}]}

    box = {"HalfSpace3::cuboid": [{"Point3::new": [2, 2, 2]}, {"Vector3::new": [4, 4, 4]}]}
    carved = {"ComposableShape3::of": [[{"HalfSpace3::cuboid": [{"Point3::new": [2, -3, -2]}, {"Vector3::new": [4, 2, 4]}]},
                                        {"Sphere3::new": [{"Point3::new": [2, -3, -2]}, 1.2]}], {"SetOperation": ["Complement"]}]}
    text = json.dumps({"Universe3": {"camera": {"FreeCamera3": []}, "entities": [glass(box), glass(carved), {"Void3::new_with_vacuum": []}],
                                     "background": {"MappedTextureImpl3::new": [{"uv_sphere_3": [{"Point3::new": [0, 0, 0]}]},
                                                                               {"texture_image_nearest_neighbor": ["./resources/simple.png"]}]}}})
    env = Parser(texture_dirs=[ROOT], low_precision=low_precision).parse(text)
    osc = OracleScene(text, default_texture_loader([ROOT]), variant="f32" if low_precision else "")
    env.camera.max_depth = 6
    for loc, fwd in (([-3.0, -0.0, 0.0], [1.0, 0.3, 0.2]), ([6.0, 0.0, -0.0], [-1.0, 0.1, 0.0]), ([-0.0, -6.0, -0.0], [0.2, 1.0, 0.3]),
                     ([2.0, 2.0, -0.0], [0.0, 0.0, 1.0]), ([-0.0, -0.0, -0.0], [1.0, 1.0, 1.0])):
        n = math.sqrt(sum(x * x for x in fwd))
        f = [x / n for x in fwd]
        left = [-f[1], f[0], 0.0] if abs(f[2]) < 0.99 else [0.0, 1.0, 0.0]
        ln = math.sqrt(sum(x * x for x in left))
        left = [x / ln for x in left]
        up = [f[1] * left[2] - f[2] * left[1], f[2] * left[0] - f[0] * left[2], f[0] * left[1] - f[1] * left[0]]
        ocam = osc.camera()
        for k in range(3):
            for cam in (env.camera, ocam):
                cam.location[k] = loc[k]; cam.forward[k] = f[k]; cam.left[k] = left[k]; cam.up[k] = up[k]
        img = env.render((128, 72), want_hit_t=True)
        orgb, ohit, ost = osc.render(128, 72, max_depth=6, want_hit_t=True, camera=ocam)
        assert np.array_equal(img.data, orgb), (loc, int((img.data != orgb).sum()))
        assert img.stats == ost
        both_nan = np.isnan(img.hit_t) & np.isnan(ohit)
        assert np.array_equal(img.hit_t[~both_nan], ohit[~both_nan])
    env.close()


@pytest.mark.parametrize("low_precision", [False, True])
@pytest.mark.parametrize("dim", [3, 4])
def test_box_rays_with_tied_plane_hits(dim, low_precision):
    """Axis-aligned cameras on the symmetry axes of cuboids / hypercuboids, outside and at the centre, odd and even frames: whole
    diagonals of pixels whose rays meet two face planes at exactly the same t, pass through edges and corners, or run inside a
    face plane.  Those are the rays the closed-form slab answer for boxes (chain_slab) must hand back to the cascade (its
    all-pairs separation test), so the frames must still equal the oracle's bit for bit."""
    import json
    from euclider_amd import Parser
    from oracle.scene_loader import OracleScene, default_texture_loader

    if dim == 3:
        def glass(shape):
            return {"Entity3Impl::new_with_surface": [shape, {"Vacuum3::new": []}, {"ComposableSurface3": {
                "reflection_ratio": {"reflection_ratio_fresnel_3": [1.458, 1]},
                "reflection_direction": {"reflection_direction_specular_3": []},
                "threshold_direction": {"threshold_direction_snell_3": [1.458]},
                "surface_color": {"surface_color_uniform_3": [{"Rgba::new": [0.2, 0.4, 0.1, 0.3]}]}}}]}
        text = json.dumps({"Universe3": {"camera": {"FreeCamera3": []}, "entities": [
            glass({"HalfSpace3::cuboid": [{"Point3::new": [5, 0, 0]}, {"Vector3::new": [2, 2, 2]}]}),
            glass({"ComposableShape3::of": [[{"HalfSpace3::cuboid": [{"Point3::new": [-6, 0, 0]}, {"Vector3::new": [4, 4, 4]}]},
                                             {"HalfSpace3::cuboid": [{"Point3::new": [-6, 0, 0]}, {"Vector3::new": [6, 2, 2]}]}],
                                            {"SetOperation": ["Complement"]}]}),
            {"Void3::new_with_vacuum": []}],
            "background": {"MappedTextureImpl3::new": [{"uv_sphere_3": [{"Point3::new": [0, 0, 0]}]},
                                                      {"texture_image_nearest_neighbor": ["./resources/simple.png"]}]}}})
        poses = [([0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]), ([0, 0, 0], [-1, 0, 0], [0, -1, 0], [0, 0, 1]),
                 ([5, 0, 0], [0, 1, 0], [-1, 0, 0], [0, 0, 1]), ([5, 1, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]),
                 ([5, 0, 8], [0, 0, -1], [0, 1, 0], [1, 0, 0])]
    else:
        text = open(os.path.join(ROOT, "scenes", "4d_frame.json")).read()
        poses = [([-10, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0]), ([0, 0, 0, 0], [0, 1, 0, 0], [-1, 0, 0, 0], [0, 0, 1, 0]),
                 ([0, 0, 9, 0], [0, 0, -1, 0], [0, 1, 0, 0], [1, 0, 0, 0]), ([1.5, 1.5, -7, 0], [0, 0, 1, 0], [0, 1, 0, 0], [-1, 0, 0, 0])]
    env = Parser(texture_dirs=[ROOT], low_precision=low_precision).parse(text)
    osc = OracleScene(text, default_texture_loader([ROOT]), variant="f32" if low_precision else "")
    env.camera.max_depth = 5
    for loc, fwd, left, up in poses:
        ocam = osc.camera()
        for k in range(dim):
            for cam in (env.camera, ocam):
                cam.location[k] = loc[k]; cam.forward[k] = fwd[k]; cam.left[k] = left[k]; cam.up[k] = up[k]
        for w, h in ((65, 65), (64, 48)):
            img = env.render((w, h), want_hit_t=True)
            orgb, ohit, ost = osc.render(w, h, max_depth=5, want_hit_t=True, camera=ocam)
            assert np.array_equal(img.data, orgb), (loc, fwd, w, h, int((img.data != orgb).sum()))
            assert img.stats == ost
            both_nan = np.isnan(img.hit_t) & np.isnan(ohit)
            gh = img.hit_t.astype(np.float32) if low_precision else img.hit_t
            assert np.array_equal(gh[~both_nan], ohit[~both_nan])
    env.close()
