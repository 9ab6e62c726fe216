"""Scene loader (product, C++ behind the C ABI): the reference's scene format and error taxonomy.

Mirrors the intent of the reference's loader tests (/root/reference/src/scene.rs:1503-1802: positional
`{"item":[42]}` form for every primitive, nested constructors) on the real registry, and adds the keyed
form the shipped scenes use, aliases and every ParserError variant (scene.rs:524-552).
"""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest

from euclider_amd import Parser, ParserError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BG = {"MappedTextureImpl3::new": [{"uv_sphere_3": [{"Point3::new": [0, 0, 0]}]}, {"texture_image_linear": ["nope.png"]}]}
SURF = {"ComposableSurface3": {"reflection_ratio": {"reflection_ratio_uniform_3": [0]},
                               "reflection_direction": {"reflection_direction_specular_3": []},
                               "threshold_direction": {"threshold_direction_identity_3": []},
                               "surface_color": {"surface_color_uniform_3": [{"Rgba::new": [1, 0, 0, 1]}]}}}


def universe(entities, camera=None):
    return json.dumps({"Universe3": {"camera": camera or {"PitchYawCamera3": []}, "entities": entities, "background": BG}})


def entity(shape):
    return {"Entity3Impl::new": [shape, {"Vacuum3::new": []}, SURF]}


def flat_words(env):
    from euclider_amd import _capi
    n = C.c_size_t()
    ptr = _capi.lib().eu_scene_flat(env._scene, C.byref(n))
    return np.frombuffer(C.string_at(ptr, n.value), dtype=np.uint64).copy()


def test_all_shipped_scenes_load():
    for path in sorted(glob.glob(os.path.join(ROOT, "scenes", "*.json"))):
        env = Parser().parse_file(path)
        assert env.info.dim in (3, 4)
        assert env.info.n_entities >= 2
        assert env.camera.max_depth == 10 and env.camera.fov_deg == 90       # d3/entity/camera.rs:49-50
        env.close()


def test_positional_and_keyed_forms_are_equivalent():
    a = Parser().parse(universe([entity({"Sphere3::new": [{"Point3::new": [10, 0, 0]}, 3]})]))
    b = Parser().parse(universe([entity({"Sphere3": {"radius": 3, "center": {"Point3": {"z": 0, "y": 0, "x": 10}}}})]))
    assert np.array_equal(flat_words(a), flat_words(b))
    a.close(); b.close()


def test_extra_positional_fields_are_ignored():            # scene.rs:490-504: the iterator is simply not exhausted
    env = Parser().parse(universe([entity({"Sphere3::new": [{"Point3::new": [1, 2, 3, 99]}, 3, "ignored"]})]))
    env.close()


def test_camera_location_constructor():
    env = Parser().parse(universe([{"Void3::new_with_vacuum": []}], {"FreeCamera3::new_with_location": [{"Point3": [1, 2, 3]}]}))
    assert list(env.camera.location)[:3] == [1.0, 2.0, 3.0]
    assert list(env.camera.forward)[:3] == [1.0, 0.0, 0.0] and list(env.camera.up)[:3] == [0.0, 0.0, 1.0]
    env.close()


def test_cuboid_is_collapsed_to_a_half_space_chain():
    env = Parser().parse(universe([entity({"HalfSpace3::cuboid": [{"Point3::new": [16, 0, -1]}, {"Vector3::new": [3, 3, 6]}]})]))
    assert env.info.n_leaves == 6 and env.info.n_shape_ops == 1           # six half-spaces, ONE chain op
    env.close()


def _flat_ops(env):
    """(kind, first, param) of every shape op of the flattened scene (flat_scene.h), header flags."""
    import ctypes as C
    import struct
    from euclider_amd import _capi
    L = _capi.lib()
    L.eu_scene_flat.restype = C.c_void_p
    n = C.c_size_t()
    raw = C.string_at(L.eu_scene_flat(env._scene, C.byref(n)), n.value)
    h = struct.unpack("<32I", raw[:128])
    n_ops, off_ops, flags = h[4], h[5], h[27]
    words = struct.unpack("<%dQ" % n_ops, raw[off_ops * 8:(off_ops + n_ops) * 8])
    return [(w & 0xff, (w >> 16) & 0xffff, w >> 32) for w in words], flags


def test_bounded_subtrees_get_guard_ops():
    """A capped cylinder (Cylinder::new_with_height = cylinder with two half-spaces, shape.rs:906-927) inside a Union is bounded:
    the loader recognises it and puts a guard op (EU_SH_SKIP = 24) in front of its five ops; `first` of the guard is the index of the
    subtree's root op.  An entity's own root never gets one (the entity record carries that bound)."""
    cyl = lambda c, d: {"Cylinder3::new_with_height": [{"Point3::new": c}, {"Vector3::new": d}, 0.5, 4]}
    union = {"ComposableShape3::of": [[cyl([0, 8, 0], [1, 0, 0]), cyl([0, -8, 0], [0, 0, 1]), {"Sphere3::new": [{"Point3::new": [9, 0, 0]}, 1]}],
                                      {"SetOperation": ["Union"]}]}
    env = Parser().parse(universe([entity(union)]))
    ops, flags = _flat_ops(env)
    kinds = [k for k, _, _ in ops]
    assert kinds.count(24) == 2 and flags & 2
    for i, (k, first, _) in enumerate(ops):
        if k == 24:
            assert first == i + 5 and ops[first][0] == 9          # cylinder, half-space, Intersection, half-space, Intersection (root)
            assert ops[first][1] == i                              # the guarded subtree starts at its guard
    assert ops[-1][0] == 8 and kinds[0] == 24                      # the entity's root Union itself is not guarded
    env.close()
    lone = Parser().parse(universe([entity(cyl([0, 0, 0], [0, 0, 1]))]))
    ops, flags = _flat_ops(lone)
    assert 24 not in [k for k, _, _ in ops] and not (flags & 2)    # entity root: bound in the entity record instead
    lone.close()


def test_box_chain_kinds():
    """EU_SH_CHAIN_BOX (18) for a cuboid whose half-space constants are all non-zero, EU_SH_CHAIN_BOX0 (19) when a face lies in a
    coordinate plane (constant +-0), an ordinary Intersection chain (17) when a normal is not +-e_k."""
    def kinds(shape):
        env = Parser().parse(universe([entity(shape)]))
        ops, _ = _flat_ops(env)
        env.close()
        return [k for k, _, _ in ops]
    assert kinds({"HalfSpace3::cuboid": [{"Point3::new": [16, 0, -1]}, {"Vector3::new": [3, 3, 6]}]}) == [18]
    assert kinds({"HalfSpace3::cuboid": [{"Point3::new": [14, -6, -2]}, {"Vector3::new": [4, 4, 4]}]}) == [19]      # 3d_room's glass block: a face at z = 0
    assert kinds({"HalfSpace3::cuboid": [{"Point3::new": [2, 2, 2]}, {"Vector3::new": [4, 4, 4]}]}) == [19]


@pytest.mark.parametrize("text,kind", [
    ("{ not json", "SyntaxError"),
    (json.dumps({"Universe3": {}, "Universe4": {}}), "InvalidConstructor"),
    (json.dumps({"Universe3": 5}), "InvalidConstructor"),
    (json.dumps({"NoSuchThing": []}), "NoDeserializer"),
    (json.dumps({"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}), "TypeMismatch"),                     # root must be an Environment
    (universe([entity({"Sphere3::new": [{"Vector3::new": [0, 0, 0]}, 1]})]), "TypeMismatch"),            # Vector3 where Point3 is expected
    (universe([entity({"Sphere3::new": [{"Point3::new": [0, 0, 0]}, "big"]})]), "TypeMismatch"),
    (universe([entity({"Sphere3::new": [{"Point3::new": [0, 0]}, 1]})]), "MissingField"),
    (universe([entity({"Sphere3": {"center": {"Point3::new": [0, 0, 0]}}})]), "MissingField"),
    (universe([entity({"ComposableShape3::of": [[{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}], {"SetOperation": ["Union"]}]})]), "CustomError"),
    (universe([entity({"ComposableShape3::of": [[{"VoidShape3": []}, {"VoidShape3": []}], {"SetOperation": ["Nope"]}]})]), "CustomError"),
    (universe([entity({"HalfSpace3::new": [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, 1]})]), "CustomError"),   # not a Hyperplane3
    (universe([entity({"Hyperplane3::new": [{"Vector3::new": [0, 0, 0]}, 1]})]), "CustomError"),          # zero normal
    (universe([entity({"Cylinder3::new": [{"Point3::new": [0, 0, 0]}, {"Vector3::new": [0, 0, 1]}, -1]})]), "CustomError"),
])
def test_parser_errors(text, kind):
    with pytest.raises(ParserError) as e:
        Parser().parse(text)
    assert e.value.kind == kind


def test_linear_space_expressions_compile():
    mat = {"LinearSpace3": {"legend": "xyz", "transformations": [{"ComponentTransformation3": {"expressions": [
        {"ComponentTransformationExpr": {"expression": "x * 4", "inverse_expression": "x / 4"}},
        {"ComponentTransformationExpr": {"expression": "y", "inverse_expression": "y"}},
        {"ComponentTransformationExpr": {"expression": "-(z + 1) ^ 2 / max(x, 2)", "inverse_expression": "sqrt(abs(z))"}}]}}]}}
    ent = {"Entity3Impl::new": [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, mat, SURF]}
    env = Parser().parse(universe([ent]))
    assert env.info.n_materials >= 1
    env.close()
    bad = json.loads(json.dumps(mat).replace('"y"', '"q"', 1))
    with pytest.raises(ParserError):
        Parser().parse(universe([{"Entity3Impl::new": [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, bad, SURF]}]))


def test_missing_texture_is_substituted_by_the_procedural_grid():
    from euclider_amd.textures import procedural_uv_grid
    g = procedural_uv_grid()
    assert g.shape == (512, 1024, 4) and g[0, 0].tolist() == [255, 255, 255, 255] and g[1, 1].tolist() == [0, 0, 64, 255]
    # the oracle's loader uses the same generator
    from oracle.scene_loader import procedural_uv_grid as oracle_grid
    assert np.array_equal(g, oracle_grid())


def test_scene_beyond_the_flat_format_is_refused():
    """The flat records hold 16-bit indices: a scene with more shape operations than that is refused, not wrapped around."""
    leaf = {"Sphere3::new": [{"Point3::new": [5, 0, 0]}, 1]}
    ents = [entity(leaf) for _ in range(0x10000)]
    text = json.dumps({"Universe3": {"camera": {"PitchYawCamera3::new": []}, "entities": ents,
                                     "background": {"MappedTextureImpl3::new": [{"uv_sphere_3": [{"Point3::new": [0, 0, 0]}]},
                                                                                {"texture_image_linear": ["./resources/none.jpg"]}]}}})
    with pytest.raises(ParserError) as ei:
        Parser().parse(text)
    assert "too many shape nodes" in str(ei.value) or "16-bit" in str(ei.value)


# ---------------------------------------------------------------- the reference's own loader tests, one by one
# /root/reference/src/scene.rs:1503-1802 registers a one-field constructor "item" on an EMPTY parser and parses `{"item": [ 42 ]}`
# for every primitive type.  The product's registry is fixed (the reference's own 100 constructors), so each test is mirrored
# with the registered constructor that takes a field of that type, in the same positional form; what is read back is the value
# as it reaches the flattened scene.  No counterpart exists for f32 (the `low_precision` feature, not built), and for u64, u16,
# usize, i32, i64, i16, i8, isize and bool: no registered constructor has a field of those types (scene.rs:620-1408).
def _params_of(env):
    """the leaf-parameter doubles of the flattened scene"""
    w = flat_words(env)
    hdr = w[:16].view(np.uint32)
    n_params, off_params = int(hdr[28]), int(hdr[29])
    return w[off_params:off_params + n_params].view(np.float64)


def test_reference_parse_float_and_f64():          # scene.rs:1503-1508, 1561-1576: {"item": [ 42 ]} -> 42.0 (an integer literal is a valid F)
    env = Parser().parse(universe([entity({"Sphere3::new": [{"Point3::new": [42, 0.5, -7e-1]}, 3]})]))
    p = _params_of(env)
    assert p[0] == 42.0 and p[1] == 0.5 and p[2] == -0.7 and p[3] == 3.0 and p[4] == 9.0
    env.close()


def test_reference_parse_str():                    # scene.rs:1510-1525: a &str field (SetOperation's name, a texture's path)
    two = [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, {"Sphere3::new": [{"Point3::new": [1, 0, 0]}, 1]}]
    for name, kind in (("Union", 8), ("Intersection", 9), ("Complement", 10), ("SymmetricDifference", 11)):
        env = Parser().parse(universe([entity({"ComposableShape3::of": [two, {"SetOperation": [name]}]})]))
        w = flat_words(env)
        hdr = w[:16].view(np.uint32)
        ops = w[int(hdr[5]):int(hdr[5]) + int(hdr[4])]
        assert int(ops[-1]) & 0xff == kind          # the root op of the entity's shape program (flat_scene.h: EU_SH_UNION = 8 ...)
        env.close()
    with pytest.raises(ParserError) as e:           # a number where a &str is expected
        Parser().parse(universe([entity({"ComposableShape3::of": [two, {"SetOperation": [42]}]})]))
    assert e.value.kind == "TypeMismatch"


def test_reference_parse_string():                 # scene.rs:1527-1542: a String field (LinearSpace3's legend)
    def mat(legend):
        return {"LinearSpace3::new": [legend, [{"ComponentTransformation3::new": [[
            {"ComponentTransformationExpr::new": ["a * 2", "a / 2"]}, {"ComponentTransformationExpr::new": ["b", "b"]},
            {"ComponentTransformationExpr::new": ["c", "c"]}]]}]]}
    env = Parser().parse(universe([{"Entity3Impl::new": [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, mat("abc"), SURF]}]))
    env.close()
    with pytest.raises(ParserError):                # the legend names the variables: "xyz" does not define a, b, c
        Parser().parse(universe([{"Entity3Impl::new": [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, mat("xyz"), SURF]}]))


def test_reference_parse_u32_and_u8():             # scene.rs:1578-1593 (u32: the Perlin seed), 1629-1644 (u8: Rgba::new_u8)
    surf = json.loads(json.dumps(SURF))
    surf["ComposableSurface3"]["surface_color"] = {"surface_color_blend_3": [
        {"surface_color_perlin_hue_seed_3": [42, 1.5, 0.25]},
        {"surface_color_uniform_3": [{"Rgba::new_u8": [255, 51, 0, 102]}]}, {"blend_function_over": []}]}
    ent = {"Entity3Impl::new": [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, {"Vacuum3::new": []}, surf]}
    env = Parser().parse(universe([ent]))
    w = flat_words(env)
    hdr = w[:16].view(np.uint32)
    off_color, n_color = int(hdr[17]), int(hdr[16])
    doubles = w[off_color:off_color + 16 * n_color].view(np.float64)
    assert any(np.allclose(doubles[k:k + 4], [1.0, 0.2, 0.0, 0.4], rtol=0, atol=0) for k in range(len(doubles) - 3))     # u8 / 255 (scene.rs:656-661)
    env.close()
    for bad in (256, -1, 1.5):                      # not a u8
        s2 = json.loads(json.dumps(surf).replace("255", json.dumps(bad), 1))
        with pytest.raises(ParserError) as e:
            Parser().parse(universe([{"Entity3Impl::new": [{"Sphere3::new": [{"Point3::new": [0, 0, 0]}, 1]}, {"Vacuum3::new": []}, s2]}]))
        assert e.value.kind == "TypeMismatch"


def test_reference_parse_vec_and_constructor():    # scene.rs:1765-1780 (Vec<T>), 1782-1802 (a constructor inside a constructor)
    ents = [entity({"Sphere3::new": [{"Point3::new": [10 * k, 0, 0]}, 1]}) for k in range(1, 4)] + [{"Void3::new_with_vacuum": []}]
    env = Parser().parse(universe(ents))
    assert env.info.n_entities == 4 and env.info.n_leaves == 4      # three spheres + the void's shape
    p = _params_of(env)
    assert [p[5 * k] for k in range(3)] == [10.0, 20.0, 30.0]
    env.close()
    with pytest.raises(ParserError) as e:           # a scalar where a Vec is expected
        Parser().parse(json.dumps({"Universe3": {"camera": {"PitchYawCamera3": []}, "entities": 5, "background": BG}}))
    assert e.value.kind in ("TypeMismatch", "InvalidConstructor")
