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


def unit(v):
    n = norm(v)
    return [x / n for x in v]


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


# ---------------------------------------------------------------------------------------------------------------------
# Fresnel / Snell (surface.rs:213-244, 268-288) against the textbook laws, to_pixel (palette RgbPixel for [u8; 4]) against
# the scene files' own colour constants.

N_AXIS = {3: unit([0.3, -0.5, 0.8]), 4: unit([0.3, -0.5, 0.8, 0.2])}      # (a normal along z would hit general_rotation's singular plane)
U_AXIS = {}
for _dim, _n in N_AXIS.items():
    _u = [1.0, 0.4, 0.1, -0.3][:_dim]
    _k = dot(_u, _n)
    U_AXIS[_dim] = unit([a - _k * b for a, b in zip(_u, _n)])


def incident(theta, dim=3):
    """A ray arriving at a surface with normal_closer = n at angle `theta` from the normal, in the plane spanned by (n, u)."""
    n, u = N_AXIS[dim], U_AXIS[dim]
    return [math.sin(theta) * a - math.cos(theta) * b for a, b in zip(u, n)], n


def fresnel_textbook(n1, n2, ti):
    st = n1 / n2 * math.sin(ti)
    if st > 1.0:
        return 1.0
    tt = math.asin(st)
    rs = ((n1 * math.cos(ti) - n2 * math.cos(tt)) / (n1 * math.cos(ti) + n2 * math.cos(tt))) ** 2
    rp = ((n1 * math.cos(tt) - n2 * math.cos(ti)) / (n1 * math.cos(tt) + n2 * math.cos(ti))) ** 2
    return (rs + rp) / 2


@pytest.mark.parametrize("dim", [3, 4])
def test_fresnel_matches_the_textbook(oracle_lib, dim):
    n_in, n_out = 1.458, 1.0                               # fused silica in vacuum (scenes/3d_fresnel.json)
    # normal incidence: ((n1 - n2) / (n1 + n2))^2 = 0.034719... from either side
    d, n = incident(0.0, dim)
    r0 = ((n_in - n_out) / (n_in + n_out)) ** 2
    assert abs(oracle_lib.eo_test_fresnel(dim, n_in, n_out, dvec(d), dvec(n), 0) - r0) <= 1e-12
    assert abs(oracle_lib.eo_test_fresnel(dim, n_in, n_out, dvec(d), dvec(n), 1) - r0) <= 1e-12
    # entering, over the whole range of angles
    for k in range(1, 90):
        ti = math.radians(k)
        d, n = incident(ti, dim)
        got = oracle_lib.eo_test_fresnel(dim, n_in, n_out, dvec(d), dvec(n), 0)
        assert abs(got - fresnel_textbook(n_out, n_in, ti)) <= 1e-9, k
    # Brewster's angle: the p component vanishes, R = Rs / 2
    tb = math.atan(n_in / n_out)
    d, n = incident(tb, dim)
    tt = math.asin(n_out / n_in * math.sin(tb))
    rs = ((n_out * math.cos(tb) - n_in * math.cos(tt)) / (n_out * math.cos(tb) + n_in * math.cos(tt))) ** 2
    assert abs(oracle_lib.eo_test_fresnel(dim, n_in, n_out, dvec(d), dvec(n), 0) - rs / 2) <= 1e-12
    # leaving the glass: total internal reflection beyond asin(1 / 1.458) = 43.3 degrees
    crit = math.degrees(math.asin(n_out / n_in))
    for k in range(1, 90):
        ti = math.radians(k)
        d, n = incident(ti, dim)
        got = oracle_lib.eo_test_fresnel(dim, n_in, n_out, dvec(d), dvec(n), 1)
        if k > crit + 0.5:
            assert got == 1.0, k
        elif k < crit - 0.5:
            assert abs(got - fresnel_textbook(n_in, n_out, ti)) <= 1e-9, k


@pytest.mark.parametrize("dim", [3, 4])
def test_snell_direction_obeys_snells_law(oracle_lib, dim):
    n_glass = 1.458
    for exiting, n1, n2 in ((0, 1.0, n_glass), (1, n_glass, 1.0)):
        for k in range(1, 89):
            ti = math.radians(k)
            if n1 / n2 * math.sin(ti) >= 1.0:
                continue                                   # (total internal reflection: the ratio provider returned 1, no transmission)
            d, n = incident(ti, dim)
            out = dvec([0.0] * 4)
            oracle_lib.eo_test_snell(dim, n_glass, dvec(d), dvec(n), exiting, out)
            o = list(out)[:dim]
            assert abs(norm(o) - 1.0) <= 1e-12              # a rotation of a unit vector
            along_u, along_n = dot(o, U_AXIS[dim]), dot(o, n)
            rest = [x - along_u * a - along_n * b for x, a, b in zip(o, U_AXIS[dim], n)]
            assert all(abs(x) <= 1e-12 for x in rest)       # stays in the plane of incidence
            assert along_n < 0.0 and along_u > 0.0          # goes on through the surface, same side of the normal
            tt = math.atan2(along_u, -along_n)
            assert abs(n1 * math.sin(ti) - n2 * math.sin(tt)) <= 1e-9, (exiting, k)


def test_to_pixel_truncates_and_clamps(oracle_lib):
    """u8 = trunc(clamp(c, 0, 1) * 255): the inverse of Rgba::new_u8 (c = u8 / 255) on every byte value, no rounding up, no
    gamma.  (255 * (k / 255) is k or k - 1ulp-ish in floating point: k / 255 * 255 must not fall below k for the scenes' own
    constants to survive a round trip -- it does not, for any k.)"""
    import ctypes as C
    px = (C.c_uint8 * 4)()
    for k in range(256):
        oracle_lib.eo_test_to_pixel(dvec([k / 255.0, (k + 0.999) / 255.0 if k < 255 else 1.0, -0.5, 7.0]), px)
        assert (px[0], px[2], px[3]) == (k, 0, 255)
        assert px[1] == k                                   # truncation: 0.999 of a step short of k + 1 is still k
    oracle_lib.eo_test_to_pixel(dvec([0.5, 0.25, 0.999, 1.0]), px)
    assert list(px) == [127, 63, 254, 255]                  # 127.5 -> 127 (a rounding to_pixel would give 128)
