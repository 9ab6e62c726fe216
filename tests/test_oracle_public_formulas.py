"""The oracle's colour and rotation arithmetic against PUBLISHED definitions (not against itself).

The blend functions live in a dependency that is absent from /root/reference (palette 0.2.1, blend/blend.rs; call site
/root/reference/src/universe/entity/surface.rs:315-322) and the oracle restates them in premultiplied form.  Here each one is
checked against the W3C "Compositing and Blending Level 1" definitions, which are written the other way round (separable
blend function B(cb, cs) on straight colours plus the general Porter-Duff formula), so an error in the restatement does not
cancel.  The two agree algebraically, not bit for bit: tolerance 1e-12 on values in [0, 1].

One known difference is asserted as such: palette follows the SVG Compositing draft's soft-light, whose middle branch has
"- 3m" where W3C has "+ 3m"; the oracle follows palette (that is what the reference links).

general_rotation (/root/reference/src/util.rs:630-666, used by the Snell threshold direction, surface.rs:284) is checked through the properties of a plane
rotation: it preserves lengths, leaves the orthogonal complement of the plane alone and turns vectors of the plane by the
given angle.
"""
import math
import random

import pytest

from oracle.scene_loader import dvec

TOL = 1e-12


def B(name, cb, cs):
    """W3C separable blend functions on straight colours (backdrop cb, source cs)."""
    def multiply(a, b): return a * b
    def screen(a, b): return a + b - a * b
    def hard_light(cb, cs): return multiply(cb, 2 * cs) if cs <= 0.5 else screen(cb, 2 * cs - 1)
    if name == "multiply": return multiply(cb, cs)
    if name == "screen": return screen(cb, cs)
    if name == "overlay": return hard_light(cs, cb)
    if name == "darken": return min(cb, cs)
    if name == "lighten": return max(cb, cs)
    if name == "dodge": return 0.0 if cb == 0 else (1.0 if cs == 1 else min(1.0, cb / (1 - cs)))
    if name == "burn": return 1.0 if cb == 1 else (0.0 if cs == 0 else 1 - min(1.0, (1 - cb) / cs))
    if name == "hard_light": return hard_light(cb, cs)
    if name == "soft_light":
        if cs <= 0.5:
            return cb - (1 - 2 * cs) * cb * (1 - cb)
        d = ((16 * cb - 12) * cb + 4) * cb if cb <= 0.25 else math.sqrt(cb)
        return cb + (2 * cs - 1) * (d - cb)
    if name == "difference": return abs(cb - cs)
    if name == "exclusion": return cb + cs - 2 * cb * cs
    raise KeyError(name)


def w3c_separable(name, src, dst):
    """co = cs*as*(1-ab) + cb*ab*(1-as) + as*ab*B(cb, cs); ao = as + ab - as*ab; straight result co / ao."""
    sa, da = src[3], dst[3]
    ao = sa + da - sa * da
    out = []
    for cs, cb in zip(src[:3], dst[:3]):
        co = cs * sa * (1 - da) + cb * da * (1 - sa) + sa * da * B(name, cb, cs)
        out.append(co / ao)
    return out + [ao]


def porter_duff(name, src, dst):
    """Fa / Fb of the Porter-Duff operators: co = as*Fa*cs + ab*Fb*cb, ao = as*Fa + ab*Fb."""
    sa, da = src[3], dst[3]
    fa, fb = {"over": (1, 1 - sa), "inside": (da, 0), "outside": (1 - da, 0), "atop": (da, 1 - sa),
              "xor": (1 - da, 1 - sa), "plus": (1, 1)}[name]
    ao = min(1.0, sa * fa + da * fb)
    return [(sa * fa * cs + da * fb * cb) / ao for cs, cb in zip(src[:3], dst[:3])] + [ao]


def oracle_blend(L, name, src, dst):
    out = dvec([0, 0, 0, 0])
    L.eo_test_blend(name.encode(), dvec(src), dvec(dst), out)
    return list(out)


def colours(seed, n):
    r = random.Random(seed)
    for _ in range(n):
        yield ([r.uniform(0.02, 0.98) for _ in range(3)] + [r.uniform(0.05, 0.95)],
               [r.uniform(0.02, 0.98) for _ in range(3)] + [r.uniform(0.05, 0.95)])


@pytest.mark.parametrize("name", ["multiply", "screen", "overlay", "darken", "lighten", "dodge", "burn", "hard_light",
                                  "difference", "exclusion"])
def test_separable_blend_modes_match_w3c(oracle_lib, name):
    for src, dst in colours(sum(map(ord, name)), 400):
        got, want = oracle_blend(oracle_lib, name, src, dst), w3c_separable(name, src, dst)
        assert all(abs(g - w) <= TOL for g, w in zip(got, want)), (name, src, dst, got, want)


@pytest.mark.parametrize("name", ["over", "inside", "outside", "atop", "xor", "plus"])
def test_porter_duff_operators_match_w3c(oracle_lib, name):
    for src, dst in colours(1000 + len(name), 400):
        if name == "plus" and src[3] + dst[3] > 1.0:
            # palette clamps alpha to 1 and leaves the premultiplied sum alone: the straight colour is the plain sum then
            want = [cs * src[3] + cb * dst[3] for cs, cb in zip(src[:3], dst[:3])] + [1.0]
        else:
            want = porter_duff(name, src, dst)
        got = oracle_blend(oracle_lib, name, src, dst)
        assert all(abs(g - w) <= TOL for g, w in zip(got, want)), (name, src, dst, got, want)


def test_soft_light_matches_w3c_outside_the_svg_draft_branch(oracle_lib):
    """Branches cs <= 0.5 and (cs > 0.5, cb > 0.25) agree with W3C; in the branch (cs > 0.5, cb <= 0.25) palette has the SVG
    draft's 16m^3 - 12m^2 - 3m (W3C: + 3m), so the premultiplied results differ by exactly 6*m*Da*(2*Sca - Sa)."""
    n_mid = 0
    for src, dst in colours(77, 1500):
        got, want = oracle_blend(oracle_lib, "soft_light", src, dst), w3c_separable("soft_light", src, dst)
        sa, da = src[3], dst[3]
        ao = sa + da - sa * da
        for k in range(3):
            cs, cb = src[k], dst[k]
            if cs > 0.5 and cb <= 0.25:
                n_mid += 1
                svg_minus_w3c = -6.0 * cb * da * (2 * cs * sa - sa) / ao
                assert abs((got[k] - want[k]) - svg_minus_w3c) <= TOL, (src, dst, k)
            else:
                assert abs(got[k] - want[k]) <= TOL, (src, dst, k)
        assert abs(got[3] - want[3]) <= TOL
    assert n_mid > 100


def test_blend_with_transparent_or_opaque_layers(oracle_lib):
    """Limits every operator must respect: a fully transparent source leaves the backdrop (separable modes, over, atop, xor,
    plus), an opaque source over anything is the source."""
    dst = [0.3, 0.6, 0.9, 0.7]
    clear = [0.5, 0.5, 0.5, 0.0]
    for name in ["over", "atop", "xor", "plus", "multiply", "screen", "overlay", "darken", "lighten", "hard_light",
                 "soft_light", "difference", "exclusion"]:
        got = oracle_blend(oracle_lib, name, clear, dst)
        assert all(abs(g - w) <= TOL for g, w in zip(got, dst)), (name, got)
    src = [0.2, 0.4, 0.8, 1.0]
    assert oracle_blend(oracle_lib, "over", src, dst) == src
    assert oracle_blend(oracle_lib, "inside", src, clear) == [0.0, 0.0, 0.0, 0.0]
    assert oracle_blend(oracle_lib, "outside", src, clear) == src


def rotate(L, dim, a, b, angle, v):
    out = dvec(list(v) + [0.0] * (4 - len(v)))
    L.eo_test_general_rotation(dim, dvec(a), dvec(b), angle, out)
    return list(out)[:dim]


def dot(a, b): return sum(x * y for x, y in zip(a, b))
def norm(a): return math.sqrt(dot(a, a))


@pytest.mark.parametrize("dim", [3, 4])
def test_general_rotation_is_a_plane_rotation(oracle_lib, dim):
    r = random.Random(dim)
    for _ in range(100):
        # an orthonormal pair (a, b) by Gram-Schmidt
        a = [r.gauss(0, 1) for _ in range(dim)]
        na = norm(a)
        a = [x / na for x in a]
        b = [r.gauss(0, 1) for _ in range(dim)]
        k = dot(a, b)
        b = [y - k * x for x, y in zip(a, b)]
        nb = norm(b)
        b = [y / nb for y in b]
        angle = r.uniform(-math.pi, math.pi)
        v = [r.gauss(0, 1) for _ in range(dim)]
        w = rotate(oracle_lib, dim, a, b, angle, v)
        assert abs(norm(w) - norm(v)) <= 1e-12 * max(1.0, norm(v))
        # the part of v outside span(a, b) does not move
        va, vb, wa, wb = dot(v, a), dot(v, b), dot(w, a), dot(w, b)
        rest_v = [x - va * p - vb * q for x, p, q in zip(v, a, b)]
        rest_w = [x - wa * p - wb * q for x, p, q in zip(w, a, b)]
        assert all(abs(x - y) <= 1e-12 for x, y in zip(rest_v, rest_w))
        # inside the plane the coordinates turn by `angle` (either orientation, fixed for the whole function)
        c, s = math.cos(angle), math.sin(angle)
        plus = abs(wa - (c * va - s * vb)) <= 1e-12 and abs(wb - (s * va + c * vb)) <= 1e-12
        minus = abs(wa - (c * va + s * vb)) <= 1e-12 and abs(wb - (-s * va + c * vb)) <= 1e-12
        assert plus or minus, (a, b, angle, v, w)


def test_general_rotation_orientation_is_fixed(oracle_lib):
    """Quarter turn in the (e0, e1) plane: e0 goes to +-e1, the other axes stay."""
    w = rotate(oracle_lib, 3, [1.0, 0.0, 0.0], [0.0, 1.0, 0.0], math.pi / 2, [1.0, 0.0, 0.0])
    assert abs(w[0]) <= 1e-15 and abs(abs(w[1]) - 1.0) <= 1e-15 and w[2] == 0.0
    w4 = rotate(oracle_lib, 4, [1.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0], math.pi / 2, [0.0, 1.0, 0.0, 1.0])
    assert abs(abs(w4[0]) - 1.0) <= 1e-15 and abs(w4[1]) <= 1e-15 and w4[2] == 0.0 and w4[3] == 1.0


def test_general_rotation_singular_plane_is_nan(oracle_lib):
    """The reference completes (self, other) to a basis with the identity's columns 2.. and orthonormalises (util.rs:632-653):
    when e2 (or e3 in 4-D) lies in the plane the leftover column is the zero vector, its normalisation is 0/0 and the whole
    matrix becomes NaN.  The oracle keeps that (a refraction whose normal/direction plane contains the z axis is undefined
    in the reference too), it does not repair it."""
    w = rotate(oracle_lib, 3, [0.0, 0.0, 1.0], [1.0, 0.0, 0.0], 0.25, [0.0, 0.0, 1.0])
    assert all(math.isnan(x) for x in w)
    w4 = rotate(oracle_lib, 4, [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0], 0.25, [0.0, 0.0, 1.0, 0.0])
    assert all(math.isnan(x) for x in w4)
