"""The reference's own known-answer tests, run against the oracle (CPU restatement).

Vectors: /root/reference/src/universe/entity/shape.rs:1048-1148 (4 shape KATs, 2-D),
/root/reference/src/util.rs:947-958 (combine_palette_color), :960-969 (remainder),
:1007-1037 (angle_between, f32 in the reference; checked here in f64 to 2 ulps of f32).
"""
import ctypes as C
import math
import struct

import pytest

from oracle.scene_loader import Intersection, dvec


def ulps(a, b):
    ia = struct.unpack("<q", struct.pack("<d", a))[0]
    ib = struct.unpack("<q", struct.pack("<d", b))[0]
    return abs(ia - ib)


def intersect(L, s, shape, loc, dirn, n=8):
    out = (Intersection * n)()
    k = L.eo_test_intersect(s, shape, dvec(loc), dvec(dirn), out, n)
    return [out[i] for i in range(k)]


@pytest.fixture()
def scene2(oracle_lib):
    s = oracle_lib.eo_scene_new(2)
    yield s
    oracle_lib.eo_scene_free(s)


def check_hit(h, loc, dirn, normal, dist):
    assert (h.location[0], h.location[1]) == loc
    assert (h.direction[0], h.direction[1]) == dirn
    assert (h.normal[0], h.normal[1]) == normal
    assert ulps(h.distance, dist) <= 2


def test_intersect_sphere_linear(oracle_lib, scene2):            # shape.rs:1048-1073
    sh = oracle_lib.eo_shape_sphere(scene2, dvec([2.0, 0.0]), 1.0)
    hits = intersect(oracle_lib, scene2, sh, [0.0, 0.0], [1.0, 0.0])
    assert len(hits) == 2
    check_hit(hits[0], (1.0, 0.0), (1.0, 0.0), (-1.0, 0.0), 1.0)
    check_hit(hits[1], (3.0, 0.0), (1.0, 0.0), (1.0, 0.0), 3.0)


def test_intersect_plane_linear(oracle_lib, scene2):             # shape.rs:1075-1095
    sh = oracle_lib.eo_shape_hyperplane_with_point(scene2, dvec([-1.0, 0.0]), dvec([1.0, 0.0]))
    hits = intersect(oracle_lib, scene2, sh, [0.0, 0.0], [1.0, 0.0])
    assert len(hits) == 1
    check_hit(hits[0], (1.0, 0.0), (1.0, 0.0), (-1.0, 0.0), 1.0)


def test_intersect_halfspace_linear(oracle_lib, scene2):         # shape.rs:1097-1120
    pl = oracle_lib.eo_shape_hyperplane_with_point(scene2, dvec([-1.0, 0.0]), dvec([1.0, 0.0]))
    sh = oracle_lib.eo_shape_halfspace_with_point(scene2, pl, dvec([2.0, 0.0]))
    hits = intersect(oracle_lib, scene2, sh, [0.0, 0.0], [1.0, 0.0])
    assert len(hits) == 1
    check_hit(hits[0], (1.0, 0.0), (1.0, 0.0), (-1.0, 0.0), 1.0)


def test_intersect_cylinder_linear(oracle_lib, scene2):          # shape.rs:1122-1148
    sh = oracle_lib.eo_shape_cylinder(scene2, dvec([2.0, 0.0]), dvec([0.0, 1.0]), 1.0)
    hits = intersect(oracle_lib, scene2, sh, [0.0, 0.0], [1.0, 0.0])
    assert len(hits) == 2
    check_hit(hits[0], (1.0, 0.0), (1.0, 0.0), (-1.0, 0.0), 1.0)
    check_hit(hits[1], (3.0, 0.0), (1.0, 0.0), (1.0, 0.0), 3.0)


def test_angle_between(oracle_lib):                              # util.rs:1007-1037
    f32_ulp = 2 * 2.0 ** -23 * math.pi
    a = [1.0, 0.0, 0.0]
    for b, expect in [([0.0, 1.0, 0.0], math.pi / 2), ([1.0, 1.0, 0.0], math.pi / 4),
                      ([-1.0, 1.0, 0.0], 3 * math.pi / 4), ([-1.0, 0.0, 0.0], math.pi)]:
        got = oracle_lib.eo_test_angle_between(3, dvec(a), dvec(b))
        assert abs(got - expect) <= f32_ulp
        assert ulps(got, expect) <= 2


def test_combine_palette_color(oracle_lib):                      # util.rs:947-958
    out = dvec([0, 0, 0, 0])
    ratio = 1.0 / 3.0
    oracle_lib.eo_test_combine_palette_color(dvec([1.0, 0.5, 0.0, 1.0]), dvec([0.0, 1.0, 0.5, 0.5]), ratio, out)
    expect = [1.0 / 3.0, 0.5 / 3.0 + 2.0 / 3.0, 1.0 / 3.0, 2.0 / 3.0]
    for g, e in zip(out, expect):
        assert ulps(g, e) <= 2


def test_remainder(oracle_lib):                                  # util.rs:960-969
    assert [oracle_lib.eo_test_remainder_i(a, 3) for a in range(-3, 4)] == [0, 1, 2, 0, 1, 2, 0]
    assert [oracle_lib.eo_test_remainder_f(float(a), 3.0) for a in range(-3, 4)] == [0, 1, 2, 0, 1, 2, 0]
