/*
 * trace_device.h -- the per-ray trace loop for gfx950, device side.
 *
 * Replaces, for one ray per lane, the reference's
 *   Universe::trace / trace_closest / intersect / material_at   (src/universe/mod.rs:61-184,229-271)
 *   the leaf intersectors and the four CSG iterators             (src/universe/entity/shape.rs:188-1038)
 *   ComposableSurface::get_color and its providers               (src/universe/entity/surface.rs:39-542)
 *   Vacuum / LinearSpace enter+exit                              (src/universe/entity/material.rs:32-163)
 *   camera ray generation                                        (src/universe/d3/entity/camera.rs:155-185,
 *                                                                 src/universe/d4/entity/camera.rs:146-176)
 * (paths relative to /root/reference).
 *
 * How it differs from the reference's shape (it is not a translation):
 *  - no recursion: a lane runs a small state machine (TRACE one segment / RETURN through pending
 *    frames); the reflect/transmit recursion lives in an explicit per-lane frame stack;
 *  - no lazy iterator objects: an entity's CSG tree is a post-order program; every node's hit
 *    stream is produced eagerly into a per-lane hit stack of compact (t, leaf|hit#|flip) records --
 *    location and normal are recomputed from the leaf only for the winning hit;
 *  - the scene is read through wave-uniform addresses from an LDS copy (broadcast reads);
 *  - the arithmetic (operation order, IEEE semantics, NaN behaviour, mid-recursion u8
 *    quantisation) is kept exactly, so results are bit-identical to the CPU restatement.
 */
#ifndef EU_TRACE_DEVICE_H
#define EU_TRACE_DEVICE_H

#include "eu_platform.h"
#include "eu_math.h"
#include "flat_scene.h"
#include "eu_real.h"      /* after the headers whose doubles stay doubles: the elementary functions and the flat record structs */

#define EU_DEV __device__ __forceinline__
#define EU_RPN_INLINE __device__ __noinline__
#define EU_MAX_DEPTH 16
#define EU_PI_C R(3.14159265358979323846264338327950288)
#define EU_FRAC_PI_2_C R(1.57079632679489661923132169163975144)
#define EU_EPS R(1.0e-6) /* nalgebra 0.8.2 approx_epsilon (UNVERIFIED), surface.rs:84,133 */

struct EuDevCamera {
    real location[4], forward[4], up[4], right[4];
    real dist;            /* sqrt(w*w+h*h) / (2 tan(fov/2)), camera.rs:176-179 */
    uint32_t max_depth, pad;
};

struct EuDevFrame {
    uint32_t width, height, row_begin, row_end;
    uint32_t tiles_x, n_tiles, debug_crosshair, single_pixel;
    uint32_t single_x, single_y;
    uint32_t local_rows, strip_count, strip_index, pad;   /* rows in the output buffer; interleaved-strip partition */
    uint32_t band_row0, band_rows, root_base, band_stride;        /* wavefront: the band of local rows traced by this pass; root_base = band_row0 * width.
                                                                   * band_stride > 1: the band is every band_stride-th group of 8 rows, starting with group band_row0 */
    real time_s;          /* time_millis, d3/entity/surface.rs:32 */
};

struct EuDevCounters {      /* device memory, zeroed before each launch */
    unsigned long long next_item;
    unsigned long long rays, bg_samples, nan_pixels, errors;
    unsigned long long phase[16];  /* diagnostic builds only */
    unsigned long long gen_count[EU_MAX_DEPTH + 2];   /* wavefront pipeline: rays queued per generation */
    unsigned long long overflow;                      /* rays / nodes dropped because a queue was full */
    unsigned long long hs_full;                       /* rays whose hit stack (the wavefront kernels' soft bound, EU_CNT_HS_FULL) was full: only the stack kernel can finish such a frame */
};

/* ------------------------------------------------------------------ scene view */
struct EuScene {
    const uint64_t *w;      /* blob (LDS or global) */
    const uint64_t *wrt;    /* the blob in device memory: texel addresses are patched there at upload */
    uint32_t off_ops, off_params, off_entities, n_entities, off_materials, off_transforms, off_code;
    uint32_t off_surfaces, off_color_ops, off_mapped, off_perlin, background, off_bounds;

    EU_DEV void init(const uint64_t *base) {
        w = base; wrt = base;
        const EuFlatHeader *h = (const EuFlatHeader *)base;
        off_ops = h->off_ops; off_params = h->off_params; off_entities = h->off_entities; n_entities = h->n_entities;
        off_materials = h->off_materials; off_transforms = h->off_transforms; off_code = h->off_code;
        off_surfaces = h->off_surfaces; off_color_ops = h->off_color_ops; off_mapped = h->off_mapped;
        off_perlin = h->off_perlin; background = h->background; off_bounds = h->off_bounds;
    }
    EU_DEV uint64_t word(uint32_t i) const { return w[i]; }
    EU_DEV eu_f64 dbl(uint32_t i) const { return __longlong_as_double((long long)w[i]); }
    EU_DEV void op(uint32_t i, uint32_t &kind, uint32_t &first, uint32_t &param, uint32_t &count) const {
        uint64_t x = w[off_ops + i];
        kind = (uint32_t)(x & 0xff); count = (uint32_t)((x >> 8) & 0xff); first = (uint32_t)((x >> 16) & 0xffff); param = (uint32_t)(x >> 32);
    }
    EU_DEV const real *params(uint32_t off) const { return (const real *)(w + off_params) + off; }      /* offsets count elements of F */
    /* EuFlatEntity by value, unpacked from two 64-bit words: there are no sub-dword scalar loads, a 16-bit field read through
     * a pointer becomes a VECTOR load followed by s_waitcnt vmcnt(0) -- a full memory drain per entity of the intersect loop */
    struct EntityView { uint32_t shape_first, shape_root, material; int32_t surface; uint32_t max_hits, bound; };
    EU_DEV EntityView entity(uint32_t e) const {
        const uint64_t a = w[off_entities + 2 * e], b = w[off_entities + 2 * e + 1];
        return EntityView{(uint32_t)(a & 0xffffu), (uint32_t)((a >> 16) & 0xffffu), (uint32_t)((a >> 32) & 0xffffu), (int32_t)(int16_t)(uint16_t)(a >> 48),
                          (uint32_t)b, (uint32_t)(b >> 32)};
    }
    EU_DEV const EuFlatSurface *surface(uint32_t s) const { return (const EuFlatSurface *)(w + off_surfaces + 8 * s); }
    EU_DEV const EuFlatColorOp *color_op(uint32_t c) const { return (const EuFlatColorOp *)(w + off_color_ops + 16 * c); }
    EU_DEV const EuFlatMapped *mapped(uint32_t m) const { return (const EuFlatMapped *)(w + off_mapped + 8 * m); }
    EU_DEV uint64_t texels(uint32_t m) const { return ((const EuFlatMapped *)(wrt + off_mapped + 8 * m))->texels; }
    EU_DEV const real *bounds(uint32_t b, int D) const { return (const real *)(w + off_bounds) + (uint32_t)(D + 2) * b; }
    EU_DEV const uint8_t *perlin(uint32_t p) const { return (const uint8_t *)(w + off_perlin + 64 * p); }
};

/* ------------------------------------------------------------------ vectors (x -> w summation order) */
template <int D> EU_DEV real vdot(const real *a, const real *b) {
    real s = a[0] * b[0];
#pragma unroll
    for (int i = 1; i < D; i++) s = s + a[i] * b[i];
    return s;
}
template <int D> EU_DEV real vnsq(const real *a) { return vdot<D>(a, a); }
template <int D> EU_DEV real vnorm(const real *a) { return sqrt(vnsq<D>(a)); }
template <int D> EU_DEV void vnormalize(const real *a, real *o) {
    real n = vnorm<D>(a);
#pragma unroll
    for (int i = 0; i < D; i++) o[i] = a[i] / n;
}
template <int D> EU_DEV real angle_between(const real *a, const real *b) {   /* util.rs:712-722 */
    real r = eu_acos(vdot<D>(a, b) / (vnorm<D>(a) * vnorm<D>(b)));
    return (r != r) ? R(0.0) : r;
}
EU_DEV real rust_signum(real x) { if (x != x) return x; return (eu_hi((eu_f64)x) >> 31) ? -R(1.0) : R(1.0); }      /* (widening keeps the sign, of zeros too) */
EU_DEV real rust_min(real a, real b) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
EU_DEV real rust_max(real a, real b) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }
EU_DEV real clamp01(real v) { if (v < R(0.0)) return R(0.0); if (v > R(1.0)) return R(1.0); return v; }
#if EU_REAL_BITS == 32
EU_DEV bool is_normal_f64(real x) { uint32_t e = (__float_as_uint(x) >> 23) & 0xffu; return e != 0 && e != 0xffu; }      /* f32::is_normal */
#else
EU_DEV bool is_normal_f64(real x) { uint32_t e = (eu_hi(x) >> 20) & 0x7ff; return e != 0 && e != 0x7ff; }
#endif
EU_DEV real remainder_f(real a, real b) {   /* util.rs:287-299 */
    /* fmod is exact by definition (a - trunc(a / b) * b, no rounding).  The texture coordinates it is used on lie within two periods:
     * |a| < b gives a itself; b <= |a| < 2 b gives a -+ b, which is exact as well (Sterbenz: the operands are within a factor of two);
     * the sign of a zero result does not matter (normalised to +0 below).  Anything else (NaN, b <= 0, further away) takes the library
     * routine, a software loop of ~80 instructions, four times per texture sample before. */
    real rem;
    const real aa = fabs(a);
    if (aa < b) rem = a;
    else if (aa < b + b) rem = a < R(0.0) ? a + b : a - b;
    else rem = fmod(a, b);
    if (rem == R(0.0)) return R(0.0);
    if (a < R(0.0)) return b + rem;
    return rem;
}

/* ------------------------------------------------------------------ leaves */
/* sphere (shape.rs:667-693) and cylinder (shape.rs:962-988) share the root selection */
struct LeafHits { int n; real t0, t1; };      /* returned by value: reference out-parameters ended up in scratch memory */

EU_DEV LeafHits quad_roots(real a, real b, real c) {
    LeafHits r = {0, R(0.0), R(0.0)};
    real d = b * b - R(4.0) * a * c;
    if (d < R(0.0)) return r;
    real d_sqrt = sqrt(d);
    real t1 = (-b - d_sqrt) / (R(2.0) * a);
    real t2 = (-b + d_sqrt) / (R(2.0) * a);
    if (t1 >= R(0.0)) {
        r.t0 = t1; r.n = 1;
        if (t2 >= R(0.0)) { r.t1 = t2; r.n = 2; }
    } else if (t2 >= R(0.0)) { r.t0 = t2; r.n = 1; }
    return r;
}

template <int D> EU_DEV LeafHits leaf_hits(uint32_t kind, const real *P, const real *o, const real *d) {
    LeafHits none = {0, R(0.0), R(0.0)};
    switch (kind) {
    case EU_SH_SPHERE: {                                  /* shape.rs:652-731 */
        real rel[D];
#pragma unroll
        for (int i = 0; i < D; i++) rel[i] = o[i] - P[i];
        real a = vnsq<D>(d);
        real b = R(2.0) * vdot<D>(d, rel);
        real c = vnsq<D>(rel) - P[D + 1];
        return quad_roots(a, b, c);
    }
    case EU_SH_PLANE: case EU_SH_HALFSPACE: {             /* shape.rs:779-809, 843-870 */
        real t = -(vdot<D>(P, o) + P[D]) / vdot<D>(P, d);
        if (t < R(0.0)) return none;
        LeafHits r = {1, t, R(0.0)};
        return r;
    }
    case EU_SH_CYLINDER: {                                /* shape.rs:935-1027 */
        const real *ax = P + D;
        real a_vec[D], delta[D], c_vec[D];
        real k = vdot<D>(d, ax);
#pragma unroll
        for (int i = 0; i < D; i++) a_vec[i] = d[i] - ax[i] * k;
#pragma unroll
        for (int i = 0; i < D; i++) delta[i] = o[i] - P[i];
        real k2 = vdot<D>(delta, ax);
#pragma unroll
        for (int i = 0; i < D; i++) c_vec[i] = delta[i] - ax[i] * k2;
        real a = vnsq<D>(a_vec);
        real b = (R(1.0) + R(1.0)) * vdot<D>(a_vec, c_vec);
        real c = vnsq<D>(c_vec) - P[2 * D + 1];
        return quad_roots(a, b, c);
    }
    default: return none;                                 /* VoidShape, shape.rs:622-631 */
    }
}

template <int D> EU_DEV void cyl_axis_point(const real *P, const real *to, real *out) {   /* shape.rs:929-932 */
    const real *ax = P + D;
    real dl[D];
#pragma unroll
    for (int i = 0; i < D; i++) dl[i] = to[i] - P[i];
    real k = vdot<D>(ax, dl);
#pragma unroll
    for (int i = 0; i < D; i++) out[i] = P[i] + ax[i] * k;
}

template <int D> EU_DEV bool leaf_inside(uint32_t kind, const real *P, const real *p) {
    switch (kind) {
    case EU_SH_VOID: return true;                         /* shape.rs:616-618 */
    case EU_SH_SPHERE: {                                  /* shape.rs:735-737 */
        real dl[D];
#pragma unroll
        for (int i = 0; i < D; i++) dl[i] = P[i] - p[i];
        return vnsq<D>(dl) <= P[D + 1];
    }
    case EU_SH_HALFSPACE: {                               /* shape.rs:874-880 */
        real result = vdot<D>(P, p) + P[D];
        return P[D + 1] == rust_signum(result);
    }
    case EU_SH_CYLINDER: {                                /* shape.rs:1032-1037 */
        real q[D], v[D];
        cyl_axis_point<D>(P, p, q);
#pragma unroll
        for (int i = 0; i < D; i++) v[i] = p[i] - q[i];
        return vnsq<D>(v) <= P[2 * D + 1];
    }
    default: return false;                                /* Hyperplane, shape.rs:814-816 */
    }
}

/* ---- half-space chains (EU_SH_CHAIN_*): all leaves of a left-fold Union / Intersection are planes ---- */
#define EU_HS_STRIDE(D) (2 * (D) + 2)
/* The chain matrices are fully unrolled: up to 8 x 7 containment tests and 28 order tests, each a lane mask (an SGPR pair) until it is
 * folded into its bit.  Left alone, the scheduler issues the comparisons of all rows first and spills the masks (v_writelane) by the
 * hundred; a scheduling fence after every row keeps one row's masks alive at a time. */
#ifndef EU_ROW_FENCE
#define EU_ROW_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

/* is_point_inside of an EU_SH_CHAIN_BOX (see chain_matrices_box for why one product replaces the dot product) */
EU_DEV bool is_neg_zero(real x) { return x == R(0.0) && (eu_hi((eu_f64)x) >> 31) != 0u; }
template <int D, bool ZC = false> EU_DEV bool chain_inside_box(const real *P, const real *p) {
    bool finite = true;
#pragma unroll
    for (int m = 0; m < D; m++) finite = finite && __builtin_isfinite(p[m]);
    if constexpr (ZC) {      /* EU_SH_CHAIN_BOX0: next to a zero constant a product of -0 is the one case the short form gets wrong: the long form then */
#pragma unroll
        for (uint32_t k = 0; k < 2 * D; k++) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            if (Pk[D] == R(0.0) && is_neg_zero(Pk[k / 2] * p[k / 2])) finite = false;
        }
    }
    bool acc = true;
    if (finite) {
#pragma unroll
        for (uint32_t k = 0; k < 2 * D; k++) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            const real r = Pk[k / 2] * p[k / 2] + Pk[D];
            acc = acc && (Pk[D + 1] == rust_signum(r));
        }
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 2 * D; k++) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            const real r = vdot<D>(Pk, p) + Pk[D];
            acc = acc && (Pk[D + 1] == rust_signum(r));
        }
    }
    return acc;
}

template <int D> EU_DEV bool chain_inside(bool is_union, uint32_t n, const real *P, const real *p) {
    bool acc = !is_union;
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) {
        if (k < n) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            const real r = vdot<D>(Pk, p) + Pk[D];
            const bool in = (Pk[D + 1] == rust_signum(r));          /* shape.rs:874-880 */
            acc = is_union ? (acc || in) : (acc && in);             /* shape.rs:591-594, no short-circuit needed: pure */
        }
    }
    return acc;
}

/* is_point_inside of the subtree ops[first..root] (shape.rs:589-600), evaluated without
 * short-circuit on a bit stack (the leaf tests are pure, so the result is the same) */
template <int D> EU_DEV bool inside_subtree(const EuScene &S, uint32_t first, uint32_t root, const real *p) {
    uint64_t st = 0;
    for (uint32_t i = first; i <= root; i++) {
        uint32_t kind, f, param, cnt;
        S.op(i, kind, f, param, cnt);
        if (kind >= EU_SH_CHAIN_UNION) {      /* chains and guards (16..): one test keeps them out of the leaves' and composites' way */
            if (kind == EU_SH_CHAIN_BOX) {
                st = (st << 1) | (chain_inside_box<D>(S.params(param), p) ? 1ull : 0ull);
            } else if (D == 3 && kind == EU_SH_CHAIN_BOX0) {      /* (the loader emits it for D = 3 only) */
                st = (st << 1) | (chain_inside_box<D, true>(S.params(param), p) ? 1ull : 0ull);
            } else if (kind == EU_SH_SKIP) {  /* guard of the bounded subtree ending at op f: a point outside its (enlarged) bounding sphere is in none of its
                                               * leaves' solids by a margin that dwarfs rounding; the walk stays wave-uniform: skipped only if no lane needs it */
                if (f <= root) {              /* (a guard whose subtree reaches beyond `root` belongs to an enclosing subtree: not ours) */
                    const real *Bd = S.bounds(param, D);
                    real rr = R(0.0);
#pragma unroll
                    for (int m = 0; m < D; m++) { const real q = p[m] - Bd[m]; rr = rr + q * q; }
                    if (__ballot(!(rr > Bd[D])) == 0ull) { st <<= 1; i = f; }
                }
            } else {
                st = (st << 1) | (chain_inside<D>(kind == EU_SH_CHAIN_UNION, cnt, S.params(param), p) ? 1ull : 0ull);
            }
        } else if (kind < EU_SH_UNION) {
            st = (st << 1) | (leaf_inside<D>(kind, S.params(param), p) ? 1ull : 0ull);
        } else {
            uint64_t b = st & 1, a = (st >> 1) & 1;
            st >>= 2;
            uint64_t r = (kind == EU_SH_UNION) ? (a | b) : (kind == EU_SH_INTERSECTION) ? (a & b)
                       : (kind == EU_SH_COMPLEMENT) ? (a & (b ^ 1)) : (a ^ b);
            st = (st << 1) | r;
        }
    }
    return (st & 1) != 0;
}

/* ------------------------------------------------------------------ per-lane state */
/* hit code: bits 0..15 op index, bits 16..23 leaf index inside a chain op, bit 30 = second root of
 * the leaf, bit 31 = normal flipped */
#define EU_HIT_SECOND 0x40000000u
#define EU_HIT_FLIP 0x80000000u

/* the per-lane hit stack: in LDS (lane-interleaved: entry k of lane l at [k*64 + l], conflict-free
 * ds_read/ds_write_b64) when the scene's static bound fits, else in private (scratch) memory */
struct HitStackLds {
    real *t; uint32_t *c; uint32_t cap;
    static constexpr bool kPrivate = false;
    EU_DEV real gt(uint32_t k) const { return t[k * 64]; }
    EU_DEV uint32_t gc(uint32_t k) const { return c[k * 64]; }
    EU_DEV void set(uint32_t k, real tt, uint32_t cc) { t[k * 64] = tt; c[k * 64] = cc; }
    EU_DEV void set_t(uint32_t k, real tt) { t[k * 64] = tt; }
};
template <int CAP> struct HitStackPriv {
    real t[CAP]; uint32_t c[CAP];
    static constexpr uint32_t cap = CAP;
    static constexpr bool kPrivate = true;
    EU_DEV real gt(uint32_t k) const { return t[k]; }
    EU_DEV uint32_t gc(uint32_t k) const { return c[k]; }
    EU_DEV void set(uint32_t k, real tt, uint32_t cc) { t[k] = tt; c[k] = cc; }
    EU_DEV void set_t(uint32_t k, real tt) { t[k] = tt; }
};

enum { FR_OVER = 0, FR_TRANS_THEN_REFL = 1, FR_COMBINE = 2 };

template <int D> struct FrameStack {
    real ratio[EU_MAX_DEPTH];
    uint32_t meta[EU_MAX_DEPTH];     /* kind | depth_of_second_child << 8 | entity << 16 */
    uint32_t px[EU_MAX_DEPTH];
    real data[EU_MAX_DEPTH][2 * D];
};

#if defined(EU_PROFILE_SHAPE) || defined(EU_PROFILE_SHADE_WAVE)      /* diagnostic builds: s_memtime shares of eval_shape's parts (EU_PROFILE_SHAPE) or of the shade kernel's
                              * sections (EU_PROFILE_SHADE_WAVE) -> EuDevCounters::phase[] (tools/shape_profile.py).
                              * The sums live in LDS, one row per wave, and are advanced by the first ACTIVE lane at each stamp: a per-lane copy
                              * would book the time a lane sits masked off to whatever region it wakes up in. */
struct LaneCounters { uint32_t rays, bg, nan_px, errors; unsigned long long *prof; };      /* prof: [16 sums][last] of this wave */
#ifdef EU_PROFILE_SHAPE_LANES  /* the same shares weighted with the lanes active at the stamp (x / 64 on the host): where the idle lanes are */
#define SHP_W_ ((unsigned long long)__builtin_popcountll(m_))
#else
#define SHP_W_ 1ull
#endif
#define SHP(c, k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); const unsigned long long m_ = __ballot(1); \
    if ((c).prof && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(m_)) { (c).prof[(k)] += (now_ - (c).prof[16]) * SHP_W_; (c).prof[16] = now_; } } while (0)
#else
struct LaneCounters { uint32_t rays, bg, nan_px, errors; };
#define SHP(c, k) do { } while (0)
#endif

/* bit 30 of LaneCounters::errors: this lane's hit stack was full (the wavefront kernels reserve the `soft` number of entries,
 * scene_host.cpp: HitUse).  The kernel's counter flush turns it into EuDevCounters::hs_full and the frame is traced again by the
 * stack kernel, whose stack has the strict size. */
#define EU_CNT_HS_FULL 0x40000000u

struct Rgba { real r, g, b, a; };

/* ------------------------------------------------------------------ half-space chains: exact, branch-light evaluation
 *
 * A cuboid is Intersection(((((h0,h1),h2),h3),h4),h5) (d3/entity/shape.rs:17-66); the reference runs
 * five nested IntersectionIterators over lazily cached one-element streams (shape.rs:291-340).  For a
 * chain every leaf yields at most one hit, so the whole cascade is a function of three small bit
 * matrices, all computed with wave-uniform control flow:
 *    present[k]           hit k exists              (!(t_k < 0), shape.rs:792)
 *    LT[k] bit a          t_a < t_k                 (the "closer" test, ties/NaN go to b: shape.rs:226,304)
 *    IN[k] bit a          leaf k contains hit a     (further_shape.is_point_inside, shape.rs:236,316)
 * Level k merges the running stream S_{k-1} (an ordered list of leaf indices, 4 bits each) with the
 * single hit of leaf k, reproducing the iterator's three modes: both present (a failed test skips),
 * only a left (a failed test ends the stream), only b left. */
template <int D>
EU_DEV void chain_matrices(uint32_t n, const real *P, const real *o, const real *d,
                           real (&tk)[EU_CHAIN_MAX], uint32_t &pres_out, uint32_t (&in_k)[EU_CHAIN_MAX], uint32_t (&lt_k)[EU_CHAIN_MAX]) {
    /* Fully unrolled over the (at most 8) leaves with wave-uniform guards: the t_k and the hit points
     * stay in registers, plane parameters arrive through scalar loads (uniform addresses), and every
     * matrix entry costs one dot product, one compare and one bit insert. */
    uint32_t pres = 0;
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) {
        tk[k] = R(0.0);
        if (k < n) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            const real t = -(vdot<D>(Pk, o) + Pk[D]) / vdot<D>(Pk, d);      /* shape.rs:789-790 */
            tk[k] = t;
            if (!(t < R(0.0))) pres |= 1u << k;
        }
    }
    /* in_k[j] bit i: leaf j contains hit i ; lt_k[j] bit i: t_i < t_j */
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) { in_k[k] = 0; lt_k[k] = 0; }
#pragma unroll
    for (uint32_t i = 0; i < EU_CHAIN_MAX; i++) {
        if (i < n && ((pres >> i) & 1u)) {      /* a leaf without a hit (t < 0) never enters a list: its row is never read (chain_merge) */
            real loc[D];
#pragma unroll
            for (int m = 0; m < D; m++) loc[m] = o[m] + d[m] * tk[i];
#pragma unroll
            for (uint32_t j = 0; j < EU_CHAIN_MAX; j++) {
                if (j < n && j != i) {
                    const real *Pj = P + j * EU_HS_STRIDE(D);
                    const real r = vdot<D>(Pj, loc) + Pj[D];
                    if (Pj[D + 1] == rust_signum(r)) in_k[j] |= 1u << i;       /* shape.rs:874-880 */
                    if (i < j && tk[i] < tk[j]) lt_k[j] |= 1u << i;
                }
            }
            EU_ROW_FENCE();
        }
    }
    pres_out = pres;
}

/* The same matrices for an EU_SH_CHAIN_BOX (leaf k: normal = s_k e_a with a = k/2, s_k = +-1, the other components +-0; constant
 * c_k finite and non-zero).  The reference's dot product with such a normal, n . x = ((n_0 x_0 + n_1 x_1) + ...), is for a
 * FINITE x a sum of signed zeros and the one term s_k x_a:
 *   - x_a != 0: adding zeros does not change a non-zero value, and s_k x_a is exact: n . x == n_a * x_a, bit for bit;
 *   - x_a == 0: n . x is a zero of some sign, and so is n_a * x_a; whatever the signs, adding c_k != 0 gives c_k exactly.
 * So `n . x + c_k` (the numerator of t, shape.rs:789, and the is_point_inside value, shape.rs:877) equals `n_a * x_a + c_k`
 * whenever x is finite.  The denominator of t, n . d, has no constant added: it is replaced only when d_a != 0.  A ray with a
 * non-finite origin or direction (the NaN rays of the general_rotation quirk) or with some d_a == 0 (parallel to a face: the
 * sign of the zero denominator decides between +inf and -inf) is not "regular" and never gets here (ray_is_regular); a
 * non-finite hit point (t overflowed) makes this routine report false.  Either way the ray is traced by the generic routine.
 * 7 (D = 3: 5) flops per dot product become 1.
 * A ZERO constant (EU_SH_CHAIN_BOX0, ZC = true; e.g. a cuboid with a face in the plane z = 0, as in 3d_room): for x_a != 0 nothing
 * changes (the sum is s_k x_a, and adding +-0 leaves it).  For x_a == +-0 every term is a zero: the reference's sum is -0 only if ALL
 * D products are -0, and after adding c_k it is -0 only if moreover c_k is -0; the short form s_k x_a + c_k is -0 iff s_k x_a is -0
 * and c_k is -0.  With c_k = +0 both are +0; with c_k = -0 they can differ only when the product s_k x_a is -0.  So the routine
 * reports false (and the wave is traced generically) whenever a product with a zero-constant leaf's normal is -0 -- a coordinate
 * that is exactly zero with the unlucky sign: it does not happen in practice and costs one comparison per use of such a leaf. */
template <int D> EU_DEV bool ray_is_regular(const real *o, const real *d) {
    bool ok = true;
#pragma unroll
    for (int m = 0; m < D; m++) ok = ok && __builtin_isfinite(o[m]) && __builtin_isfinite(d[m]) && d[m] != R(0.0);
    return ok;
}

template <int D, bool ZC = false>
EU_DEV bool chain_matrices_box(const real *P, const real *o, const real *d,
                               real (&tk)[EU_CHAIN_MAX], uint32_t &pres_out, uint32_t (&in_k)[EU_CHAIN_MAX], uint32_t (&lt_k)[EU_CHAIN_MAX]) {
    constexpr uint32_t n = 2 * D;
    bool ok = true;         /* the caller has checked the ray itself (ray_is_regular) */
    uint32_t pres = 0;
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) {
        tk[k] = R(0.0);
        if (k < n) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            const real t = -(Pk[k / 2] * o[k / 2] + Pk[D]) / (Pk[k / 2] * d[k / 2]);
            if constexpr (ZC) { if (Pk[D] == R(0.0) && is_neg_zero(Pk[k / 2] * o[k / 2])) ok = false; }
            tk[k] = t;
            if (!(t < R(0.0))) pres |= 1u << k;
        }
    }
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) { in_k[k] = 0; lt_k[k] = 0; }
#pragma unroll
    for (uint32_t i = 0; i < EU_CHAIN_MAX; i++) {
        if (i < n && ((pres >> i) & 1u)) {      /* (rows of leaves without a hit are never read) */
            real loc[D];
#pragma unroll
            for (int m = 0; m < D; m++) { loc[m] = o[m] + d[m] * tk[i]; ok = ok && __builtin_isfinite(loc[m]); }
#pragma unroll
            for (uint32_t j = 0; j < EU_CHAIN_MAX; j++) {
                if (j < n && j != i) {
                    const real *Pj = P + j * EU_HS_STRIDE(D);
                    const real r = Pj[j / 2] * loc[j / 2] + Pj[D];
                    if constexpr (ZC) { if (Pj[D] == R(0.0) && is_neg_zero(Pj[j / 2] * loc[j / 2])) ok = false; }
                    /* signum_j == rust_signum(r) (shape.rs:874-880): signum_j is +-1 (the loader only calls such chains boxes) and r is
                     * finite when loc is (else the result is discarded), so the two are equal exactly when their sign bits are */
                    if (((eu_hi(Pj[D + 1]) ^ eu_hi(r)) >> 31) == 0u) in_k[j] |= 1u << i;
                    if (i < j && tk[i] < tk[j]) lt_k[j] |= 1u << i;
                }
            }
            EU_ROW_FENCE();
        }
    }
    pres_out = pres;
    return ok;
}

template <int D>
EU_DEV uint32_t chain_merge(bool is_union, uint32_t n, uint32_t pres, const uint32_t (&in_k)[EU_CHAIN_MAX], const uint32_t (&lt_k)[EU_CHAIN_MAX], uint32_t &list_out) {
    {   /* A hit reaches the end of the cascade only if every level lets it through: at its own level k it is the B element and
         * must pass inside_of_A (Intersection: all earlier leaves contain it, Union: none does); at every later level j it is an A
         * element and is emitted only if leaf j contains it (Intersection) / does not (Union); an element that fails is never
         * emitted later (it is skipped or it ends the stream).  So the final list only holds hits that pass the test of EVERY other
         * leaf; if no present hit does, the stream is empty whatever the order of the merges. */
        uint32_t acc = is_union ? 0u : 0xffffffffu;
#pragma unroll
        for (uint32_t j = 0; j < EU_CHAIN_MAX; j++) {
            if (j < n) { if (is_union) acc |= in_k[j] & ~(1u << j); else acc &= in_k[j] | (1u << j); }
        }
        if (((is_union ? ~acc : acc) & pres & ((1u << n) - 1u)) == 0u) { list_out = 0; return 0; }
    }
    uint32_t list = 0, len = (pres & 1u);
#pragma unroll
    for (uint32_t k = 1; k < EU_CHAIN_MAX; k++) {
        if (k < n) {
            /* inside_of_A(h_k): Intersection -> every earlier leaf contains it, Union -> any (shape.rs:591-594) */
            bool all = true, any = false;
#pragma unroll
            for (uint32_t j = 0; j < k; j++) { const bool in = (in_k[j] >> k) & 1u; all = all && in; any = any || in; }
            const bool bpass = is_union ? !any : all;
            bool bpend = (pres >> k) & 1u, term = false;
            uint32_t out = 0, olen = 0;
#pragma unroll
            for (uint32_t p = 0; p < k; p++) {
                if (p < len && !term) {
                    const uint32_t a = (list >> (4 * p)) & 15u;
                    if (bpend && !((lt_k[k] >> a) & 1u)) {          /* b is closer (or tie / NaN): it goes first */
                        bpend = false;
                        if (bpass) { out |= k << (4 * olen); olen++; }
                    }
                    const bool in = (in_k[k] >> a) & 1u;
                    if (in != is_union) { out |= a << (4 * olen); olen++; }
                    else if (!bpend) term = true;                   /* only a left and it fails: the stream ends */
                }
            }
            if (bpend && !term && bpass) { out |= k << (4 * olen); olen++; }
            list = out; len = olen;
        }
    }
    list_out = list;
    return len;
}

/* The stream of an EU_SH_CHAIN_BOX in closed form -- the slab test -- for the rays where it provably equals the cascade.
 * Along a ray, leaf j (normal s_j e_a, signum_j) contains the point at parameter t iff sign(s_j d_a (t - t_j)) == signum_j: an
 * "entry" leaf (sign(s_j d_a) == signum_j) contains exactly the points behind its own hit, t > t_j, an "exit" leaf those before it.
 * Each test of the cascade (chain_merge) is therefore monotone in t, and then the cascade is a plain filter: at level k the
 * elements in front of h_k are skipped when they fail; the elements behind it, which alone can END the stream, all pass an entry
 * leaf and all fail an exit leaf (so ending there drops only elements that fail anyway); a leaf without a hit (t_k < 0) is an
 * entry leaf that every element passes or an exit leaf that every element fails.  The final stream is the ascending list of the
 * present hits that every other leaf contains: the latest entry hit t_in = max over entry leaves and the earliest exit hit
 * t_out = min over exit leaves if t_in < t_out, nothing otherwise -- [in, out], [out] (t_in < 0) or [] (t_out < 0).
 * That argument is about exact arithmetic; the cascade's tests are evaluated in floating point (shape.rs:874-880 on
 * loc = o + d t_i, two roundings, then s_j loc_a + c_j, whose last rounding cannot change the sign) and its order tests on the
 * computed t_k.  The computed test of pair (i, j) has the exact sign whenever
 *      |t_i - t_j| > eps (2.1 |t_j| + 2.02 |t_i| + 1.01 |o_a / d_a|)           (eps = 2^-53; F = f32: 2^-24)
 * (error of loc_a: eps (2 |d_a t_i| + |o_a|); error of the computed root t_j: 2 eps |t_j|).  The routine accepts a ray only if
 * EVERY pair of the 2 D computed t_k is further apart than 2^9 eps (2 max|t_k| + max|o_a| / min|d_a|) -- over a hundred times that
 * bound -- if no t_k is NaN, and if |t_k|, |o_a|, |d_a| stay below 1e100 (F = f32: 1e15) so that every hit point is finite (the reference's
 * full dot product needs all coordinates finite: chain_matrices_box's comment, which also covers the one-product form of t_k used
 * here and its ZC caveat).  Then all the cascade's decisions are the exact ones and the closed form is its result; any other
 * ray (one through an edge or a corner, a symmetric tie, an overflow) sets `fail`, and eval_chain sends the wave through the matrices.
 * A ray PARALLEL to a pair of faces (d_a == +-0; every ray of 4d_frame: the camera's rays have no w component and no mirror gives
 * them one): for such a leaf the reference divides the non-zero numerator by a zero, t_j = +-inf.  Its hit, if +inf counts as
 * present, lies at a point with a NaN coordinate (0 * inf) and fails every other leaf's test; it is the last element of any list,
 * so whether it is skipped or ends the stream is the same.  As a tester, leaf j sees loc_a = o_a + (+-0) t_i = o_a for every finite
 * hit i: it contains all of them or none, by the sign of s_j o_a + c_j (the very value the cascade computes), and then skipping
 * the failing elements (hit j present) and ending the stream at the first (hit j absent) give the same empty list.  So a parallel
 * leaf only contributes `outside`; an origin exactly in its plane (value 0: the sign of a zero sum is not reproduced here, and t_j
 * would be NaN) sends the ray to the matrices.  The other leaves are handled as above.
 * Measured (1080p frames): 5 % of 4d_frame's waves hold a refused ray (the camera sits at the centre of its concentric boxes: the
 * image diagonals tie), 8 % of 3d_room's (the camera's z = 0 plane is a face plane of a cuboid, and every descendant of the centre
 * row stays in it), 0.02 % of 3d_hallways'. */
template <int D, bool ZC = false>
EU_DEV uint32_t chain_slab(const real *P, const real *o, const real *d, real (&tk)[EU_CHAIN_MAX], uint32_t &list_out, bool &fail) {
    constexpr uint32_t n = 2 * D;
    const real big = EU_REAL_BITS == 32 ? R(1e15) : R(1e100);
    bool ok = true, have_e = false, have_x = false, outside = false;
    real t_in = R(0.0), t_out = R(0.0), tmax = R(0.0), tg[n];
    uint32_t i_in = 0, i_out = 0;
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) {
        tk[k] = R(0.0);
        if (k < n) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            const real sk = Pk[k / 2], ck = Pk[D], gk = Pk[D + 1];
            const real num = sk * o[k / 2] + ck;                                            /* as chain_matrices_box */
            if constexpr (ZC) { if (ck == R(0.0) && is_neg_zero(sk * o[k / 2])) ok = false; }
            if (d[k / 2] == R(0.0)) {
                /* the ray runs parallel to this face: see above */
                if (!(num != R(0.0))) ok = false;          /* (zero or NaN) */
                if (((eu_hi(gk) ^ eu_hi(num)) >> 31) != 0u) outside = true;
                tg[k] = big * R(16.0) * (real)(k + 1);      /* (keeps the separation test below branch-free) */
            } else {
                const real t = -num / (sk * d[k / 2]);
                tk[k] = t;
                tg[k] = t;
                if (t != t) ok = false;
                const real at = __builtin_fabs(t);
                if (at > tmax) tmax = at;
                const bool entry = ((eu_hi(sk) ^ eu_hi(d[k / 2]) ^ eu_hi(gk)) >> 31) == 0u;
                if (entry) { if (!have_e || t > t_in) { t_in = t; i_in = k; } have_e = true; }
                else { if (!have_x || t < t_out) { t_out = t; i_out = k; } have_x = true; }
            }
        }
    }
    real omax = R(0.0), dmin = big, dmax = R(0.0);
#pragma unroll
    for (int m = 0; m < D; m++) {
        const real ao = __builtin_fabs(o[m]), ad = __builtin_fabs(d[m]);
        if (ao > omax) omax = ao;
        if (ad > dmax) dmax = ad;
        if (ad < dmin && ad != R(0.0)) dmin = ad;
    }
    const real tol = (EU_REAL_BITS == 32 ? R(3.0517578125e-05) : R(5.684341886080802e-14)) * (R(2.0) * tmax + omax / dmin);
    real gap = big;
#pragma unroll
    for (uint32_t i = 0; i < n; i++) {
#pragma unroll
        for (uint32_t j = i + 1; j < n; j++) { const real g = __builtin_fabs(tg[i] - tg[j]); if (g < gap) gap = g; }
    }
    ok = ok && have_e && have_x && gap > tol && tmax < big && omax < big && dmax < big;      /* (a NaN or infinite o / d fails here) */
    uint32_t len = 0, list = 0;
    if (!outside && t_in < t_out) {
        if (!(t_in < R(0.0))) { list = i_in | (i_out << 4); len = 2; }      /* t_out > t_in >= 0 */
        else if (!(t_out < R(0.0))) { list = i_out; len = 1; }
    }
    if (!ok) fail = true;
    list_out = list;
    return len;
}

/* use_box (wave-uniform): 2 -> an EU_SH_CHAIN_BOX is answered by chain_slab; if that refuses one of the wave's rays, the wave
 * evaluates this chain again as the ordinary Intersection chain it is (nothing else is repeated, and `fail` stays untouched).
 * 1 -> round 2's route, kept for A/B builds (-DEU_NO_SLAB): chain_matrices_box, where a lane that cannot be served sets `fail` (its
 * result is then meaningless) and the caller traces the wave's rays again with 0 -> every chain through the generic matrices. */
template <int D>
EU_DEV uint32_t eval_chain(uint32_t kind, uint32_t n, const real *P, const real *o, const real *d,
                           real (&tk)[EU_CHAIN_MAX], uint32_t &list_out, int use_box, bool &fail, LaneCounters *prof = nullptr) {
    if (use_box == 2 && (kind == EU_SH_CHAIN_BOX || (D == 3 && kind == EU_SH_CHAIN_BOX0))) {      /* wave-uniform */
        bool refused = false;
        const uint32_t len = kind == EU_SH_CHAIN_BOX ? chain_slab<D>(P, o, d, tk, list_out, refused) : chain_slab<D, true>(P, o, d, tk, list_out, refused);
        if (__ballot(refused) == 0ull) return len;
        use_box = 0;
    }
    uint32_t pres = 0, in_k[EU_CHAIN_MAX], lt_k[EU_CHAIN_MAX];
    if (use_box && kind == EU_SH_CHAIN_BOX) {
        if (!chain_matrices_box<D>(P, o, d, tk, pres, in_k, lt_k)) fail = true;
    } else if (D == 3 && use_box && kind == EU_SH_CHAIN_BOX0) {
        if (!chain_matrices_box<D, true>(P, o, d, tk, pres, in_k, lt_k)) fail = true;
    } else chain_matrices<D>(n, P, o, d, tk, pres, in_k, lt_k);
#ifdef EU_PROFILE_SHAPE
    if (prof) { __builtin_amdgcn_wave_barrier(); SHP(*prof, 1); }
#endif
    const uint32_t len = chain_merge<D>(kind == EU_SH_CHAIN_UNION, n, pres, in_k, lt_k, list_out);
#ifdef EU_PROFILE_SHAPE
    if (prof) { __builtin_amdgcn_wave_barrier(); SHP(*prof, 2); }
#endif
    return len;
}

/* First element of a Union chain's stream without building the streams (trace_closest only looks at element 0,
 * universe/mod.rs:114).  Let m be the hit with the strictly smallest t among the leaves' hits, and let its point lie
 * outside every OTHER half-space of the chain (tested with is_point_inside's own arithmetic, shape.rs:874-880).  Then m is
 * the first element of every Union node above leaf m (shape.rs:212-264), by induction up the left fold:
 *   - m comes from the left stream A, B is a leaf: both present -> t_m < t_b strictly, so m is taken and B does not contain
 *     it -> returned; only A present -> `B.inside(m) ? None : m` -> m;
 *   - m is the right leaf B, A's stream holds only hits of other leaves (all with larger t): both present -> `a.t < b.t` is
 *     false, b = m is taken, `A.is_point_inside(m)` is the OR over A's half-spaces -> false -> returned; only B present -> m.
 * No hit at all -> every stream is empty.  Anything else (a tie, a NaN t, a point inside another half-space) is left to
 * eval_chain.  Returns 1 (t_out, idx_out), 0 (empty) or -1 (undecided). */
template <int D>
EU_DEV int union_chain_first(uint32_t n, const real *P, const real *o, const real *d, real &t_out, uint32_t &idx_out) {
    real tk[EU_CHAIN_MAX];
    uint32_t pres = 0;
    bool has_nan = false;
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) {
        tk[k] = R(0.0);
        if (k < n) {
            const real *Pk = P + k * EU_HS_STRIDE(D);
            const real t = -(vdot<D>(Pk, o) + Pk[D]) / vdot<D>(Pk, d);      /* shape.rs:789-790 */
            tk[k] = t;
            if (!(t < R(0.0))) { pres |= 1u << k; if (t != t) has_nan = true; }
        }
    }
    if (pres == 0) return 0;
    real best = R(0.0); uint32_t idx = 0; bool have = false;
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) {
        if (k < n && ((pres >> k) & 1u) && (!have || tk[k] < best)) { best = tk[k]; idx = k; have = true; }
    }
    bool ok = !has_nan;
#pragma unroll
    for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) {
        if (k < n && ((pres >> k) & 1u) && k != idx && !(best < tk[k])) ok = false;       /* a tie */
    }
    real loc[D];
#pragma unroll
    for (int m = 0; m < D; m++) loc[m] = o[m] + d[m] * best;
#pragma unroll
    for (uint32_t j = 0; j < EU_CHAIN_MAX; j++) {
        if (j < n) {
            const real *Pj = P + j * EU_HS_STRIDE(D);
            const real r = vdot<D>(Pj, loc) + Pj[D];
            if (j != idx && Pj[D + 1] == rust_signum(r)) ok = false;                        /* inside another half-space */
        }
    }
    if (!ok) return -1;
    t_out = best; idx_out = idx;
    return 1;
}

/* Exact culling.  The loader gives every bounded entity (sphere, cuboid / hypercuboid, and Union /
 * Intersection / Complement / SymmetricDifference trees over them) a bounding sphere enlarged by 1e-6.
 * If the half-line o + d*t, t >= 0, stays outside it, no hit point of any leaf of the entity lies in the
 * entity's shape: every hit point of a box chain violates one of the other half-spaces by a margin that
 * dwarfs rounding (so the chain's hit stream is empty), a sphere's own discriminant is negative, and an
 * operand that is not bounded only contributes points that fail the bounded operand's is_point_inside.
 * The entity's stream is therefore empty and trace_closest would skip it anyway (universe/mod.rs:114):
 * skipping the evaluation changes nothing.  The same test is applied to every box chain inside a CSG tree
 * (its stream is empty, so an empty list is pushed without evaluating the chain).  NaN anywhere makes every comparison false: no culling. */
template <int D> EU_DEV bool ray_misses_bound(const real *Bd, const real *o, const real *d) {
    real rel[D];
#pragma unroll
    for (int i = 0; i < D; i++) rel[i] = o[i] - Bd[i];
    const real rr = vdot<D>(rel, rel);
    const real cc = rr - Bd[D];
    if (cc > R(0.0)) {                                   /* origin outside the enlarged sphere */
        const real b = vdot<D>(d, rel);
        if (b >= R(0.0)) return true;                    /* moving away: the closest point is the origin */
        if (rr < Bd[D + 1]) {                         /* discriminant margin only holds for |o-c| < 1e4 R */
            const real a = vdot<D>(d, d);
            if (b * b - a * cc < R(0.0)) return true;    /* the whole line misses */
        }
    }
    return false;
}

/* t_k of a chain's leaf `idx`, picked with compares (for a hit stack in private memory: staging the t_k through it, as the LDS form
 * does, is eight scratch stores and a dependent load per chain -- 4d_frame wrote 0.7 GB per frame that way) */
EU_DEV real chain_pick_t(const real (&tk)[EU_CHAIN_MAX], uint32_t count, uint32_t idx) {
    /* a chain of selects over VALUES: every t_k passes through an (empty) asm statement first, otherwise the optimiser turns the
     * selects into ONE load from a select of addresses, the array stays in scratch memory, and every such load sits behind an
     * `s_waitcnt vmcnt(0)` in the middle of the ray loop -- which also waits for the next ray's prefetch (round 4: 80 bytes of
     * scratch and three such waits per ray batch in 3d_room's intersect kernel) */
    real t = tk[0];
#pragma unroll
    for (uint32_t k = 1; k < EU_CHAIN_MAX; k++) if (k < count) {
        real x = tk[k];
        asm("" : "+v"(x));
        t = (idx == k) ? x : t;
    }
    return t;
}

/* ------------------------------------------------------------------ CSG: eager post-order evaluation
 * The pieces below are shared by the interpreter (eval_shape walks an entity's shape program op by op) and by the
 * scene-specialised kernels (jit.cpp emits the same calls in a straight line, every kind, count and parameter a constant). */

/* one evaluated hit list on the per-lane hit stack: n entries; rep: the stream repeats its last element for ever;
 * unk: "unknown beyond" (see eval_shape) */
struct CsgList { uint32_t n; bool rep, unk; };

/* a bare leaf or chain that is a whole entity: element 0 of its stream, no list machinery (trace_closest only looks at
 * element 0, universe/mod.rs:114).  P: the op's parameters (a chain's bounding sphere follows its leaves). */
template <int D, class HS>
EU_DEV uint32_t eval_single(uint32_t kind, uint32_t count, const real *P, const real *o, const real *d, HS &hs, uint32_t op_index,
                            LaneCounters &cnt, real &first_t, uint32_t &first_c, int use_box, bool &fail) {
    SHP(cnt, 6);
    if (kind >= EU_SH_CHAIN_UNION) {
        real tk[EU_CHAIN_MAX]; uint32_t list;
        const real *Pb = P + count * EU_HS_STRIDE(D);          /* the chain's bounding sphere (r2 < 0: none) */
        if (Pb[D] >= R(0.0) && ray_misses_bound<D>(Pb, o, d)) { SHP(cnt, 0); return 0u; }
        SHP(cnt, 0);
        if (kind == EU_SH_CHAIN_UNION) {
            real tf = R(0.0); uint32_t idx = 0;
            const int q = union_chain_first<D>(count, P, o, d, tf, idx);
            SHP(cnt, 7);
            if (q == 0) return 0u;
            if (q > 0) { first_t = tf; first_c = op_index | (idx << 16); return 1u; }
        }
#ifdef EU_PROFILE_SHAPE
        const uint32_t n = eval_chain<D>(kind, count, P, o, d, tk, list, use_box, fail, &cnt);
#else
        const uint32_t n = eval_chain<D>(kind, count, P, o, d, tk, list, use_box, fail);
#endif
        if (n) {
            first_t = chain_pick_t(tk, count, list & 15u);
            first_c = op_index | ((list & 15u) << 16);
        }
        SHP(cnt, 3);
        return n;
    }
    const LeafHits lh = leaf_hits<D>(kind, P, o, d);
    if (lh.n) { first_t = lh.t0; first_c = op_index; }
    SHP(cnt, 4);
    return (uint32_t)lh.n;
}

/* the hits of a leaf op inside a tree: pushed at hs[sp ..) */
template <int D, class HS>
EU_DEV CsgList push_leaf(uint32_t kind, const real *P, const real *o, const real *d, HS &hs, uint32_t &sp, uint32_t op_index, LaneCounters &cnt) {
    SHP(cnt, 6);
    const LeafHits lh = leaf_hits<D>(kind, P, o, d);
    int n = lh.n;
    if (sp + 2 > hs.cap) { cnt.errors |= EU_CNT_HS_FULL; n = 0; }
    if (n >= 1) hs.set(sp, lh.t0, op_index);
    if (n >= 2) hs.set(sp + 1, lh.t1, op_index | EU_HIT_SECOND);
    sp += (uint32_t)n;
    SHP(cnt, 4);
    return CsgList{(uint32_t)n, false, false};
}

/* the stream of a half-space chain op inside a tree */
template <int D, class HS>
EU_DEV CsgList push_chain(uint32_t kind, uint32_t count, const real *P, const real *o, const real *d, HS &hs, uint32_t &sp, uint32_t op_index,
                          LaneCounters &cnt, int use_box, bool &fail) {
    real tk[EU_CHAIN_MAX]; uint32_t list = 0, n = 0;
    const real *Pb = P + count * EU_HS_STRIDE(D);          /* the chain's bounding sphere (r2 < 0: none) */
    SHP(cnt, 6);
    /* A Union chain may emit a hit per leaf; on an LDS stack their t_k are staged behind the list and picked by run-time index.
     * An Intersection chain (a box) emits at most two hits but for rounding noise: its t are picked out of the registers, and
     * the loader reserves two entries (scene_host.cpp: HitUse) -- a stream that does not fit marks the lane (EU_CNT_HS_FULL). */
    const bool staged = !HS::kPrivate && kind == EU_SH_CHAIN_UNION;
    if (!(Pb[D] >= R(0.0) && ray_misses_bound<D>(Pb, o, d))) {
        SHP(cnt, 0);
#ifdef EU_PROFILE_SHAPE
        n = eval_chain<D>(kind, count, P, o, d, tk, list, use_box, fail, &cnt);
#else
        n = eval_chain<D>(kind, count, P, o, d, tk, list, use_box, fail);
#endif
    }
    SHP(cnt, 0);
    if (sp + (staged ? 2 * count : n) > hs.cap) { cnt.errors |= EU_CNT_HS_FULL; n = 0; }
#ifdef EU_TEST_HS_FULL      /* test hook (tests/test_gpu_jit.py, through eu_renderer_opts.jit_flags): every box that is hit pretends not to fit */
    if (!staged && n == 2) { cnt.errors |= EU_CNT_HS_FULL; n = 0; }
#endif
    if (!staged) {
        for (uint32_t p = 0; p < n; p++) {
            const uint32_t idx = (list >> (4 * p)) & 15u;
            hs.set(sp + p, chain_pick_t(tk, count, idx), op_index | (idx << 16));
        }
    } else {
        if (n) {
#pragma unroll
            for (uint32_t k = 0; k < EU_CHAIN_MAX; k++) if (k < count) hs.set_t(sp + count + k, tk[k]);
        }
        for (uint32_t p = 0; p < n; p++) {
            const uint32_t idx = (list >> (4 * p)) & 15u;
            hs.set(sp + p, hs.gt(sp + count + idx), op_index | (idx << 16));
        }
    }
    sp += n;
    SHP(cnt, 3);
    return CsgList{n, false, false};
}

/* One composite op: merges the two topmost lists of the hit stack (A below B) the way the reference's four iterators do
 * (shape.rs:188-497) and leaves the result in their place.  inside_a / inside_b: is_point_inside of the child subtrees
 * (shape.rs:589-600).  The reference evaluates streams lazily and trace_closest only asks for element 0; this evaluation is
 * eager.  Where the eager merge meets something the reference would never finish computing (a stream that neither ends nor
 * yields), the list is cut there and marked "unknown beyond".  A parent that runs past such a cut inherits the mark; only an
 * entity whose FIRST element is unknown counts as an error (the reference would spin). */
template <int D, class HS, class InsA, class InsB>
EU_DEV CsgList csg_merge(uint32_t kind, bool is_root, HS &hs, uint32_t &sp, const CsgList A, const CsgList B, const real *o, const real *d,
                         LaneCounters &cnt, InsA inside_a, InsB inside_b) {
    SHP(cnt, 6);
    const uint32_t CAP = hs.cap;
    const uint32_t nb = B.n, na = A.n;
    const bool unk_a = A.unk, unk_b = B.unk;
    bool out_unk = false;
    const bool rep_a = A.rep && na > 0, rep_b = B.rep && nb > 0;
    const uint32_t b0 = sp - nb, a0 = b0 - na;
    /* A right operand of at most two hits (a leaf, a box: every B of a left fold) is taken into registers and the output is written
     * over it -- the loader's bound counts on that (scene_host.cpp: emit_shape); a longer B stays where it is, output above it. */
    const bool b_regs = nb <= 2u;
    real rb_t0 = R(0.0), rb_t1 = R(0.0); uint32_t rb_c0 = 0, rb_c1 = 0;
    if (b_regs && nb >= 1u) { rb_t0 = hs.gt(b0); rb_c0 = hs.gc(b0); }
    if (b_regs && nb == 2u) { rb_t1 = hs.gt(b0 + 1); rb_c1 = hs.gc(b0 + 1); }
    const uint32_t o0 = b_regs ? b0 : sp;
    uint32_t ia = 0, ib = 0, no = 0;
    bool out_rep = false;
    const uint32_t guard_max = 4 * (na + nb) + 8;
    for (uint32_t guard = 0;; guard++) {
        const bool sa = ia < na || rep_a, sb = ib < nb || rep_b;
        if ((!sa && unk_a) || (!sb && unk_b)) { out_unk = true; break; }      /* next() asks both children first (shape.rs:214-215 ...) */
        if (!sa && !sb) break;
        if (guard >= guard_max) { out_unk = true; break; }                    /* runaway: the reference would spin here */
        real ta = R(0.0), tb = R(0.0); uint32_t ca = 0, cb = 0;
        if (sa) { uint32_t k = a0 + (ia < na ? ia : na - 1); ta = hs.gt(k); ca = hs.gc(k); }
        if (sb) {
            const uint32_t kb = ib < nb ? ib : nb - 1;
            if (b_regs) { tb = kb == 0u ? rb_t0 : rb_t1; cb = kb == 0u ? rb_c0 : rb_c1; }
            else { tb = hs.gt(b0 + kb); cb = hs.gc(b0 + kb); }
        }
        const bool both = sa && sb;
        const bool take_a = both ? (ta < tb) : sa;       /* ties go to b (shape.rs:226,304,375,448) */
        if (kind == EU_SH_COMPLEMENT && !both && sa) {   /* shape.rs:390-392: returns a without advancing */
            if (o0 + no >= CAP) { if (!(is_root && no > 0)) cnt.errors |= EU_CNT_HS_FULL; break; }  /* capacity (na + nb + 1; at the root only element 0 matters) */
            hs.set(o0 + no, ta, ca); no++;
            out_rep = true;
            break;
        }
        /* consuming the repeated tail of a never-ending child stream leaves the iterator state
         * unchanged: the same decision recurs forever */
        const bool stuck = take_a ? (ia >= na) : (ib >= nb);
        const real t = take_a ? ta : tb;
        uint32_t c = take_a ? ca : cb;
        real loc[D];
#pragma unroll
        for (int k = 0; k < D; k++) loc[k] = o[k] + d[k] * t;
        const bool ins = take_a ? inside_b(loc) : inside_a(loc);
        if (take_a) ia++; else ib++;
        bool emit = false, end = false;
        switch (kind) {
        case EU_SH_UNION:                                /* shape.rs:212-264 */
            if (both) emit = !ins; else { if (ins) end = true; else emit = true; }
            break;
        case EU_SH_INTERSECTION:                         /* shape.rs:291-340 */
            if (both) emit = ins; else { if (ins) emit = true; else end = true; }
            break;
        case EU_SH_COMPLEMENT:                           /* shape.rs:365-409 */
            if (take_a) emit = !ins;                      /* only reachable with both present */
            else { if (ins) { emit = true; c ^= EU_HIT_FLIP; } else if (!both) end = true; }
            break;
        default:                                         /* SymmetricDifference, shape.rs:436-496 */
            emit = true;
            if (ins) c ^= EU_HIT_FLIP;
            break;
        }
        /* a decision taken on a repeated tail recurs forever with the same outcome (the state did not change): the
         * element it would emit is the one emitted the step before, so the list is only marked as repeating */
        if (stuck) { if (emit) out_rep = true; else if (!end) out_unk = true; break; }   /* no output and no end, forever: the reference would spin */
        if (emit) {
            if (o0 + no >= CAP) { if (!(is_root && no > 0)) cnt.errors |= EU_CNT_HS_FULL; break; }   /* capacity (the loader's bound is na + nb) */
            hs.set(o0 + no, t, c); no++;
            /* trace_closest asks an entity's stream for element 0 only (universe/mod.rs:114) and the reference's iterators are
             * lazy: what the root's merge would produce after its first element is never computed there */
            if (is_root) break;
        }
        if (end) break;
    }
    for (uint32_t k = 0; k < no; k++) hs.set(a0 + k, hs.gt(o0 + k), hs.gc(o0 + k));
    SHP(cnt, 5);
    sp = a0 + no;
    return CsgList{no, out_rep, out_unk};
}

/* element 0 of a tree's root list (after its last csg_merge) */
template <class HS>
EU_DEV uint32_t csg_root_result(const CsgList L, HS &hs, LaneCounters &cnt, real &first_t, uint32_t &first_c) {
    if (L.n) { first_t = hs.gt(0); first_c = hs.gc(0); }
    else if (L.unk) cnt.errors++;          /* element 0 itself is something the reference never finishes computing */
    return L.n;
}

/* Evaluates entity shape program ops[first..root] for ray (o, d); returns the number of hits of the
 * entity's stream and its first element (only that is used by trace_closest, universe/mod.rs:114). */
template <int D, class HS>
EU_DEV uint32_t eval_shape(const EuScene &S, uint32_t first, uint32_t root, const real *o, const real *d,
                           HS &hs, LaneCounters &cnt, real &first_t, uint32_t &first_c, int use_box, bool &fail) {
    if (first == root) {   /* a bare leaf or chain: no list machinery */
        uint32_t kind, f, param, count;
        S.op(root, kind, f, param, count);
        return eval_single<D>(kind, count, S.params(param), o, d, hs, root, cnt, first_t, first_c, use_box, fail);
    }
    uint32_t sp = 0;          /* entries in use */
    uint64_t lens = 0;        /* stack of list lengths, 8 bits each (bit 7: stream repeats its last element forever) */
    uint32_t unk = 0;         /* "unknown beyond" marks, one bit per stacked list */
    for (uint32_t i = first; i <= root; i++) {
        uint32_t kind, f, param, count;
        S.op(i, kind, f, param, count);
        if (kind >= EU_SH_CHAIN_UNION) {
            if (kind == EU_SH_SKIP) {      /* guard of the bounded subtree ending at op f (flat_scene.h): every ray of the wave misses its sphere -> its stream is empty */
                if (__ballot(!ray_misses_bound<D>(S.bounds(param, D), o, d)) == 0ull) { lens <<= 8; unk <<= 1; i = f; }
                continue;
            }
            const CsgList L = push_chain<D>(kind, count, S.params(param), o, d, hs, sp, i, cnt, use_box, fail);
            lens = (lens << 8) | (uint64_t)L.n;
            unk <<= 1;
            continue;
        }
        if (kind < EU_SH_UNION) {
            const CsgList L = push_leaf<D>(kind, S.params(param), o, d, hs, sp, i, cnt);
            lens = (lens << 8) | (uint64_t)L.n;
            unk <<= 1;
            continue;
        }
        /* composite: children b = ops[i-1] (subtree [fb, i-1]), a = ops[fb-1] (subtree [f, fb-1]) */
        uint32_t kb, fb, pb, cb_;
        S.op(i - 1, kb, fb, pb, cb_);
        const uint32_t fa = f + count;      /* a composite op's `count`: 1 if its own guard op sits at f (not part of child a), else 0 */
        const uint32_t ra = fb - 1, rb = i - 1;
        const uint32_t lb = (uint32_t)(lens & 0xff), la = (uint32_t)((lens >> 8) & 0xff);
        lens >>= 16;
        const CsgList B = {lb & 0x7f, (lb & 0x80) != 0, (unk & 1u) != 0}, A = {la & 0x7f, (la & 0x80) != 0, (unk & 2u) != 0};
        unk >>= 2;
        const CsgList L = csg_merge<D>(kind, i == root, hs, sp, A, B, o, d, cnt,
                                       [&](const real *p) { return inside_subtree<D>(S, fa, ra, p); },
                                       [&](const real *p) { return inside_subtree<D>(S, fb, rb, p); });
        lens = (lens << 8) | (uint64_t)(L.n | (L.rep ? 0x80u : 0u));
        unk = (unk << 1) | (L.unk ? 1u : 0u);
    }
    const CsgList top = {(uint32_t)(lens & 0x7f), false, (unk & 1u) != 0};
    return csg_root_result(top, hs, cnt, first_t, first_c);
}

template <int D, class HS>
EU_DEV uint32_t eval_shape(const EuScene &S, uint32_t first, uint32_t root, const real *o, const real *d,
                           HS &hs, LaneCounters &cnt, real &first_t, uint32_t &first_c) {
    bool fail = false;
    return eval_shape<D, HS>(S, first, root, o, d, hs, cnt, first_t, first_c, false, fail);
}

/* normal of a hit on a leaf (P: the leaf's own parameters; a chain's leaf counts as a half-space), recomputed from the leaf */
template <int D>
EU_DEV void leaf_normal(uint32_t kind, const real *P, const real *o, const real *d, const real *loc, real *n) {
    switch (kind) {
    case EU_SH_SPHERE: {                                  /* shape.rs:708-709 */
        real v[D];
#pragma unroll
        for (int i = 0; i < D; i++) v[i] = loc[i] - P[i];
        vnormalize<D>(v, n);
        break;
    }
    case EU_SH_PLANE:
#pragma unroll
        for (int i = 0; i < D; i++) n[i] = P[i];
        break;
    case EU_SH_HALFSPACE:                                 /* shape.rs:860 */
#pragma unroll
        for (int i = 0; i < D; i++) n[i] = P[D + 2 + i];
        break;
    default: {                                            /* cylinder: axis point of hit 1 serves both hits (shape.rs:999,1017) */
        const real t0 = leaf_hits<D>(kind, P, o, d).t0;
        real l1[D], q[D], v[D];
#pragma unroll
        for (int i = 0; i < D; i++) l1[i] = o[i] + d[i] * t0;
        cyl_axis_point<D>(P, l1, q);
#pragma unroll
        for (int i = 0; i < D; i++) v[i] = loc[i] - q[i];
        vnormalize<D>(v, n);
        break;
    }
    }
}

/* normal of the hit described by `code` at parameter t */
template <int D>
EU_DEV void hit_normal(const EuScene &S, uint32_t code, const real *o, const real *d, const real *loc, real *n) {
    uint32_t kind, f, param, count;
    S.op(code & 0xffffu, kind, f, param, count);
    const real *P = S.params(param);
    if (kind >= EU_SH_CHAIN_UNION) { P += ((code >> 16) & 0xffu) * EU_HS_STRIDE(D); kind = EU_SH_HALFSPACE; }
    leaf_normal<D>(kind, P, o, d, loc, n);
    if (code & EU_HIT_FLIP) {
#pragma unroll
        for (int i = 0; i < D; i++) n[i] = -n[i];
    }
}

/* Universe::material_at (universe/mod.rs:229-251): first entity containing the point */
template <int D> EU_DEV int material_at_range(const EuScene &S, uint32_t e0, uint32_t e1, const real *p) {
    for (uint32_t e = e0; e < e1; e++) {
        const EuScene::EntityView E = S.entity(e);
        if (inside_subtree<D>(S, E.shape_first, E.shape_root, p)) return (int)e;
    }
    return -1;
}
template <int D> EU_DEV int material_at(const EuScene &S, const real *p) { return material_at_range<D>(S, 0u, S.n_entities, p); }

/* ------------------------------------------------------------------ materials */
/* LinearSpace expressions: meval evaluates in f64 whatever F is and the result is cast to F (material.rs:99-111), so these
 * routines are eu_f64 throughout */
EU_DEV eu_f64 rpn_min(eu_f64 a, eu_f64 b) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
EU_DEV eu_f64 rpn_max(eu_f64 a, eu_f64 b) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }
EU_DEV eu_f64 rpn_signum(eu_f64 x) { if (x != x) return x; return (eu_hi(x) >> 31) ? -1.0 : 1.0; }
EU_DEV eu_f64 pow_int(eu_f64 x, eu_f64 y) {   /* meval powf restricted to integral |y| <= 64 (documented deviation) */
    if (!(y == floor(y)) || fabs(y) > 64.0) return __longlong_as_double(0x7ff8000000000000ll);
    int n = (int)fabs(y);
    eu_f64 r = 1.0;
    for (int i = 0; i < n; i++) r = r * x;
    return (y < 0.0) ? 1.0 / r : r;
}

template <int D> EU_RPN_INLINE eu_f64 eval_rpn(const EuScene &S, uint64_t prog, const eu_f64 *ctx) {
    uint32_t off = (uint32_t)prog, len = (uint32_t)(prog >> 32);
    eu_f64 st[8];
    int sp = 0;
    for (uint32_t i = 0; i < len; i++) {
        uint64_t wd = S.word(S.off_code + off + i);
        uint32_t op = (uint32_t)wd, arg = (uint32_t)(wd >> 32);
        switch (op) {
        case EU_RPN_CONST: i++; st[sp++ & 7] = S.dbl(S.off_code + off + i); break;
        case EU_RPN_VAR: { eu_f64 v = ctx[0];
#pragma unroll
            for (int k = 1; k < D; k++) if ((int)arg == k) v = ctx[k];
            st[sp++ & 7] = v; break; }
        case EU_RPN_NEG: st[(sp - 1) & 7] = -st[(sp - 1) & 7]; break;
        case EU_RPN_FN: {
            eu_f64 y = st[(sp - 1) & 7], x = y;
            if (arg == EU_FN_MIN || arg == EU_FN_MAX || arg == EU_FN_ATAN2) { sp--; x = st[(sp - 1) & 7]; }
            eu_f64 r;
            switch (arg) {
            case EU_FN_SQRT: r = sqrt((eu_f64)x); break;
            case EU_FN_ABS: r = fabs(x); break;
            case EU_FN_FLOOR: r = floor(x); break;
            case EU_FN_CEIL: r = ceil(x); break;
            case EU_FN_MIN: r = rpn_min(x, y); break;
            case EU_FN_MAX: r = rpn_max(x, y); break;
            case EU_FN_SIN: r = eu_sin_f64(x); break;
            case EU_FN_COS: r = eu_cos_f64(x); break;
            case EU_FN_TAN: r = eu_tan_f64(x); break;
            case EU_FN_ASIN: r = eu_asin_f64(x); break;
            case EU_FN_ACOS: r = eu_acos_f64(x); break;
            case EU_FN_ATAN: r = eu_atan_f64(x); break;
            case EU_FN_ATAN2: r = eu_atan2_f64(x, y); break;
            default: r = rpn_signum(x); break;
            }
            st[(sp - 1) & 7] = r;
            break;
        }
        default: {
            eu_f64 y = st[(sp - 1) & 7]; sp--;
            eu_f64 x = st[(sp - 1) & 7];
            eu_f64 r;
            switch (op) {
            case EU_RPN_ADD: r = x + y; break;
            case EU_RPN_SUB: r = x - y; break;
            case EU_RPN_MUL: r = x * y; break;
            case EU_RPN_DIV: r = x / y; break;
            case EU_RPN_REM: r = fmod((eu_f64)x, (eu_f64)y); break;
            default: r = pow_int(x, y); break;
            }
            st[(sp - 1) & 7] = r;
            break;
        }
        }
    }
    return st[0];
}

/* Material::enter / exit (material.rs:135-162); the evaluation context is the vector BEFORE the
 * transformation (material.rs:99-111) */
template <int D> EU_DEV void material_apply(const EuScene &S, uint32_t material, real *dir, bool exit_) {
    uint64_t m = S.word(S.off_materials + material);
    uint32_t kind = (uint32_t)m & 0xff, ntr = ((uint32_t)m >> 8), first = (uint32_t)(m >> 32);
    if (kind != EU_MAT_LINEAR) return;
    for (uint32_t k = 0; k < ntr; k++) {
        uint32_t tr = exit_ ? (first + ntr - 1 - k) : (first + k);
        eu_f64 ctx[D];
#pragma unroll
        for (int i = 0; i < D; i++) ctx[i] = dir[i];
#pragma unroll
        for (int i = 0; i < D; i++) dir[i] = (real)eval_rpn<D>(S, S.word(S.off_transforms + 8 * tr + (exit_ ? 4 : 0) + (uint32_t)i), ctx);
    }
}

/* ------------------------------------------------------------------ palette 0.2.1 (UNVERIFIED third-party semantics) */
EU_DEV Rgba into_premultiplied(Rgba c) { real a = clamp01(c.a); return Rgba{c.r * a, c.g * a, c.b * a, a}; }
EU_DEV Rgba from_premultiplied(Rgba p) {
    real a = clamp01(p.a);
    Rgba c;
    if (is_normal_f64(a)) { c.r = p.r / a; c.g = p.g / a; c.b = p.b / a; }
    else { c.r = R(0.0); c.g = R(0.0); c.b = R(0.0); }
    c.a = a;
    return c;
}
__device__ __noinline__ real blend_chan(uint32_t fn, real a, real b, real sa, real da) {
    const real one = R(1.0), two = R(2.0);
    switch (fn) {
    case EU_BL_OVER: return a + b * (one - sa);
    case EU_BL_INSIDE: return a * da;
    case EU_BL_OUTSIDE: return a * (one - da);
    case EU_BL_ATOP: return a * da + b * (one - sa);
    case EU_BL_XOR: return a * (one - da) + b * (one - sa);
    case EU_BL_PLUS: return a + b;
    case EU_BL_MULTIPLY: return a * b + a * (one - da) + b * (one - sa);
    case EU_BL_SCREEN: return a + b - a * b;
    case EU_BL_OVERLAY:
        if (b * two <= da) return two * a * b + a * (one - da) + b * (one - sa);
        return a * (one + da) + b * (one + sa) - two * a * b - sa * da;
    case EU_BL_DARKEN: return rust_min(a * da, b * sa) + a * (one - da) + b * (one - sa);
    case EU_BL_LIGHTEN: return rust_max(a * da, b * sa) + a * (one - da) + b * (one - sa);
    case EU_BL_DODGE:
        if (a == sa && !is_normal_f64(b)) return a * (one - da);
        if (a == sa) return sa * da + a * (one - da) + b * (one - sa);
        return sa * da * rust_min(one, (b / da) * sa / (sa - a)) + a * (one - da) + b * (one - sa);
    case EU_BL_BURN:
        if (!is_normal_f64(a) && b == da) return sa * da + b * (one - sa);
        if (!is_normal_f64(a)) return b * (one - sa);
        return sa * da * (one - rust_min(one, (one - b / da) * sa / a)) + a * (one - da) + b * (one - sa);
    case EU_BL_HARD_LIGHT:
        if (a * two <= sa) return two * a * b + a * (one - da) + b * (one - sa);
        return a * (one + da) + b * (one + sa) - two * a * b - sa * da;
    case EU_BL_SOFT_LIGHT: {
        real m = is_normal_f64(da) ? b / da : R(0.0);
        if (a * two <= sa) return b * (sa + (two * a - sa) * (one - m)) + a * (one - da) + b * (one - sa);
        if (b * R(4.0) <= da) {
            real m2 = m * m, m3 = m2 * m;
            return da * (two * a - sa) * (m3 * R(16.0) - m2 * R(12.0) - m * R(3.0)) + a - a * da + b;
        }
        return da * (two * a - sa) * (sqrt(m) - m) + a - a * da + b;
    }
    case EU_BL_DIFFERENCE: return a + b - two * rust_min(a * da, b * sa);
    default: return a + b - two * a * b;
    }
}
EU_DEV real blend_alpha(uint32_t fn, real sa, real da) {
    switch (fn) {
    case EU_BL_INSIDE: return clamp01(sa * da);
    case EU_BL_OUTSIDE: return clamp01(sa * (R(1.0) - da));
    case EU_BL_ATOP: return clamp01(da);
    case EU_BL_XOR: return clamp01(sa + da - R(2.0) * sa * da);
    case EU_BL_PLUS: return clamp01(sa + da);
    default: return clamp01(sa + da - sa * da);
    }
}
EU_DEV Rgba blend_pre(uint32_t fn, Rgba s, Rgba d) {
    Rgba o;
    const real one = R(1.0), two = R(2.0), sa = s.a, da = d.a;
    switch (fn) {      /* the modes the shipped scenes use are expanded in line; the rest go through blend_chan */
    case EU_BL_OVER:
        o.r = s.r + d.r * (one - sa); o.g = s.g + d.g * (one - sa); o.b = s.b + d.b * (one - sa);
        break;
    case EU_BL_DARKEN:
        o.r = rust_min(s.r * da, d.r * sa) + s.r * (one - da) + d.r * (one - sa);
        o.g = rust_min(s.g * da, d.g * sa) + s.g * (one - da) + d.g * (one - sa);
        o.b = rust_min(s.b * da, d.b * sa) + s.b * (one - da) + d.b * (one - sa);
        break;
    case EU_BL_DIFFERENCE:
        o.r = s.r + d.r - two * rust_min(s.r * da, d.r * sa);
        o.g = s.g + d.g - two * rust_min(s.g * da, d.g * sa);
        o.b = s.b + d.b - two * rust_min(s.b * da, d.b * sa);
        break;
    default:
        o.r = blend_chan(fn, s.r, d.r, sa, da);
        o.g = blend_chan(fn, s.g, d.g, sa, da);
        o.b = blend_chan(fn, s.b, d.b, sa, da);
        break;
    }
    o.a = blend_alpha(fn, sa, da);
    return o;
}
EU_DEV Rgba blend_rgba(uint32_t fn, Rgba s, Rgba d) { return from_premultiplied(blend_pre(fn, into_premultiplied(s), into_premultiplied(d))); }   /* surface.rs:315-322 */
EU_DEV Rgba combine_palette_color(Rgba a, Rgba b, real r) {   /* util.rs:265-285 */
    if (r <= R(0.0)) return b;
    if (r >= R(1.0)) return a;
    return Rgba{a.r * r + b.r * (R(1.0) - r), a.g * r + b.g * (R(1.0) - r), a.b * r + b.b * (R(1.0) - r), a.a * r + b.a * (R(1.0) - r)};
}
EU_DEV uint32_t to_u8(real c, LaneCounters &cnt) {
    real v = clamp01(c) * R(255.0);
    if (v != v) { cnt.nan_px++; return 0; }
    return (uint32_t)v;
}
EU_DEV uint32_t to_pixel4(Rgba c, LaneCounters &cnt) {
    return to_u8(c.r, cnt) | (to_u8(c.g, cnt) << 8) | (to_u8(c.b, cnt) << 16) | (to_u8(c.a, cnt) << 24);
}
EU_DEV Rgba new_u8(uint32_t px) {
    return Rgba{(real)(px & 0xff) / R(255.0), (real)((px >> 8) & 0xff) / R(255.0), (real)((px >> 16) & 0xff) / R(255.0), (real)(px >> 24) / R(255.0)};
}
EU_DEV void hsv_to_rgb(real hue, real saturation, real value, real &r, real &g, real &b) {
    real deg = hue;
    if (fabs(deg) < R(1.0e9)) {
        while (deg >= R(360.0)) deg = deg - R(360.0);
        while (deg < R(0.0)) deg = deg + R(360.0);
    }
    real c = value * saturation;
    real h = deg / R(60.0);
    real x = c * (R(1.0) - fabs(fmod(h, R(2.0)) - R(1.0)));
    real m = value - c;
    real red, green, blue;
    if (h >= R(0.0) && h < R(1.0)) { red = c; green = x; blue = R(0.0); }
    else if (h >= R(1.0) && h < R(2.0)) { red = x; green = c; blue = R(0.0); }
    else if (h >= R(2.0) && h < R(3.0)) { red = R(0.0); green = c; blue = x; }
    else if (h >= R(3.0) && h < R(4.0)) { red = R(0.0); green = x; blue = c; }
    else if (h >= R(4.0) && h < R(5.0)) { red = x; green = R(0.0); blue = c; }
    else { red = c; green = R(0.0); blue = x; }
    r = red + m; g = green + m; b = blue + m;
}

/* ------------------------------------------------------------------ own 4-D gradient noise (documented substitute
 * for noise 0.4.1 Perlin + rand::random() seed, d3/entity/surface.rs:22-58) */
EU_DEV real pfade(real t) { return t * t * t * (t * (t * R(6.0) - R(15.0)) + R(10.0)); }
EU_DEV real plerp(real t, real a, real b) { return a + t * (b - a); }
EU_DEV real pgrad4(int hash, real x, real y, real z, real w) {
    int h = hash & 31;
    real a = (h < 24) ? x : y;
    real b = (h < 16) ? y : z;
    real c = (h < 8) ? z : w;
    return ((h & 1) ? -a : a) + ((h & 2) ? -b : b) + ((h & 4) ? -c : c);
}
EU_DEV int pcell(real f) { real m = fmod(f, R(256.0)); return (m == m) ? (((int)m) & 255) : 0; }
__device__ __noinline__ real perlin4(const uint8_t *perm, real x, real y, real z, real w) {
    real fx = floor(x), fy = floor(y), fz = floor(z), fw = floor(w);
    int xi = pcell(fx), yi = pcell(fy), zi = pcell(fz), wi = pcell(fw);
    real xf = x - fx, yf = y - fy, zf = z - fz, wf = w - fw;
    real u = pfade(xf), v = pfade(yf), s = pfade(zf), q = pfade(wf);
    real lw[2];
    for (int dw = 0; dw < 2; dw++) {
        real lz[2];
        for (int dz = 0; dz < 2; dz++) {
            real ly[2];
            for (int dy = 0; dy < 2; dy++) {
                real n[2];
                for (int dx = 0; dx < 2; dx++) {
                    int hsh = perm[perm[perm[perm[xi + dx] + yi + dy] + zi + dz] + wi + dw];
                    n[dx] = pgrad4(hsh, xf - (real)dx, yf - (real)dy, zf - (real)dz, wf - (real)dw);
                }
                ly[dy] = plerp(u, n[0], n[1]);
            }
            lz[dz] = plerp(v, ly[0], ly[1]);
        }
        lw[dw] = plerp(s, lz[0], lz[1]);
    }
    return R(0.87) * plerp(q, lw[0], lw[1]);
}

/* ------------------------------------------------------------------ textures */
EU_DEV bool cast_u32(real x, uint32_t &out, LaneCounters &cnt) {   /* NumCast: None (panic) when NaN / out of range */
    if (!(x > -R(1.0) && x < R(4294967296.0))) { cnt.errors++; out = 0; return false; }
    out = (uint32_t)x;
    return true;
}

/* MappedTextureImpl::get_color (surface.rs:528-534) = texture(uv_sphere(point)); M: the mapped-texture record, texels: the device
 * address of its RGBA8 texels (patched into the device copy of the scene at upload) */
EU_DEV Rgba mapped_get_color(const EuFlatMapped *M, uint64_t texels, const real *point, LaneCounters &cnt) {
    real p[3], pn[3];                                        /* d3/entity/surface.rs:60-68 (uv_derank drops w) */
#pragma unroll
    for (int i = 0; i < 3; i++) p[i] = point[i] - M->center[i];
    vnormalize<3>(p, pn);
    real pu = R(0.5) + eu_atan2(pn[1], pn[0]) / (R(2.0) * EU_PI_C);
    real pv = R(0.5) - eu_asin(pn[2]) / EU_PI_C;
    const uint32_t W = M->w, H = M->h;
    /* texels live in device (global) memory; the address comes out of the scene blob, so say so: a generic pointer would be
     * read with flat loads (vmcnt and lgkmcnt both) */
    typedef const uint32_t __attribute__((address_space(1))) *global_u32_ptr;
    const global_u32_ptr tex = (global_u32_ptr)(uintptr_t)texels;
    if (M->tex_kind == EU_TEX_NEAREST) {                       /* surface.rs:434-451 */
        real x = floor(pu * M->wd), y = floor(pv * M->hd);
        uint32_t xi, yi;
        cast_u32(x, xi, cnt); cast_u32(y, yi, cnt);
        xi = xi % W; yi = yi % H;
        return new_u8(tex[(size_t)yi * W + xi]);
    }
    real x = pu * M->wd - R(0.5), y = pv * M->hd - R(0.5);        /* surface.rs:453-489 */
    real ox = x - floor(x), oy = y - floor(y);
    uint32_t x0, x1, y0, y1;
    /* the reference casts x and y once per texel (4 texels, surface.rs:462-472); each coordinate
     * serves two texels, so a failed cast counts twice */
    if (!cast_u32(remainder_f(x + R(0.0), M->wd), x0, cnt)) cnt.errors++;
    if (!cast_u32(remainder_f(x + R(1.0), M->wd), x1, cnt)) cnt.errors++;
    if (!cast_u32(remainder_f(y + R(0.0), M->hd), y0, cnt)) cnt.errors++;
    if (!cast_u32(remainder_f(y + R(1.0), M->hd), y1, cnt)) cnt.errors++;
    if (x0 >= W) { cnt.errors++; x0 = W - 1; }
    if (x1 >= W) { cnt.errors++; x1 = W - 1; }
    if (y0 >= H) { cnt.errors++; y0 = H - 1; }
    if (y1 >= H) { cnt.errors++; y1 = H - 1; }
    uint32_t p0 = tex[(size_t)y0 * W + x0], p1 = tex[(size_t)y0 * W + x1];
    uint32_t p2 = tex[(size_t)y1 * W + x0], p3 = tex[(size_t)y1 * W + x1];
    real ch[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        real c0 = (real)((p0 >> (8 * k)) & 0xff), c1 = (real)((p1 >> (8 * k)) & 0xff);
        real c2 = (real)((p2 >> (8 * k)) & 0xff), c3 = (real)((p3 >> (8 * k)) & 0xff);
        ch[k] = ((c0 * (R(1.0) - ox) + c1 * ox) * (R(1.0) - oy) + (c2 * (R(1.0) - ox) + c3 * ox) * oy) / R(255.0);
    }
    return Rgba{ch[0], ch[1], ch[2], ch[3]};
}
EU_DEV Rgba mapped_get_color(const EuScene &S, uint32_t id, const real *point, LaneCounters &cnt) { return mapped_get_color(S.mapped(id), S.texels(id), point, cnt); }

/* ------------------------------------------------------------------ surface providers */
/* TracingContext (shape.rs:111-125) plus a cache of the angles the providers keep asking for.
 *
 * The reference calls angle_between (util.rs:712-722: acos(a.b / (|a||b|)), NaN -> 0) once in
 * trace_closest (direction vs normal, universe/mod.rs:118), again in the Fresnel ratio and in Snell
 * (direction vs -normal_closer, surface.rs:219,274) and again in both illumination providers
 * (surface.rs:399,415).  normal_closer is +-normal, negation is exact and IEEE division is
 * sign-symmetric, so all of them are acos(+x_e) or acos(-x_e) of ONE quotient
 *     x_e = dot(direction, normal) / (|direction| * |normal|)
 * and each value is computed at most once per hit, bit-identical to the separate evaluations. */
template <int D> struct HitCtx {
    real loc[D], dir[D], normal[D], nc[D];
    bool exiting;
    real x_e, ang_e;          /* ang_e = angle_between(direction, normal) */
    real ang_me;              /* acos(-x_e), NaN -> 0; valid when have_me */
    bool have_me;
    real sin_from, to_mult, to_theta;   /* Fresnel / Snell: sin(from_theta); asin(to_mult * sin_from) */
    bool have_sin, have_to;

    EU_DEV void finish(real best_t, const real *o, const real *d) {
#pragma unroll
        for (int k = 0; k < D; k++) { loc[k] = o[k] + d[k] * best_t; dir[k] = d[k]; }
    }
    EU_DEV void classify() {    /* universe/mod.rs:118-125 */
        x_e = vdot<D>(dir, normal) / (vnorm<D>(dir) * vnorm<D>(normal));
        const real r = eu_acos(x_e);
        ang_e = (r != r) ? R(0.0) : r;
        have_me = false; have_sin = false; have_to = false;
        exiting = ang_e < EU_FRAC_PI_2_C;
#pragma unroll
        for (int k = 0; k < D; k++) nc[k] = exiting ? -normal[k] : normal[k];
    }
    EU_DEV real angle_neg() {
        if (!have_me) { const real r = eu_acos(-x_e); ang_me = (r != r) ? R(0.0) : r; have_me = true; }
        return ang_me;
    }
    EU_DEV real angle_dir_nc() { return exiting ? angle_neg() : ang_e; }          /* angle_between(nc, dir) */
    EU_DEV real angle_dir_minus_nc() { return exiting ? ang_e : angle_neg(); }    /* angle_between(dir, -nc) */
    EU_DEV real sin_from_theta() {
        if (!have_sin) { sin_from = eu_sin(angle_dir_minus_nc()); have_sin = true; }
        return sin_from;
    }
    EU_DEV real to_theta_for(real mult) {                                        /* asin(mult * sin(from_theta)) */
        if (have_to && mult == to_mult) return to_theta;
        to_mult = mult; to_theta = eu_asin(mult * sin_from_theta()); have_to = true;
        return to_theta;
    }
};

template <int D> EU_DEV real reflection_ratio(const EuFlatSurface *F, HitCtx<D> &c) {
    if (F->ratio_kind == EU_RATIO_UNIFORM) return c.exiting ? R(0.0) : F->ratio_p0;     /* surface.rs:200-211 */
    const real from_theta = c.angle_dir_minus_nc();                                 /* surface.rs:213-244 */
    const real from_index = c.exiting ? F->ratio_p0 : F->ratio_p1;
    const real to_index = c.exiting ? F->ratio_p1 : F->ratio_p0;
    const real to_theta = c.to_theta_for(from_index / to_index);
    if (to_theta != to_theta) return R(1.0);
    const real cos_from = eu_cos(from_theta), cos_to = eu_cos(to_theta);
    real p1s = from_index * cos_from;
    real p2s = to_index * cos_to;
    real p1p = from_index * cos_to;
    real p2p = to_index * cos_from;
    real rs = (p1s - p2s) / (p1s + p2s); rs = rs * rs;
    real rp = (p1p - p2p) / (p1p + p2p); rp = rp * rp;
    return (rs + rp) / (R(1.0) + R(1.0));
}

/* GeneralRotation::general_rotation for one vector (util.rs:631-666) */
template <int D> EU_DEV void general_rotation(const real *self, const real *other, real angle, real *vec) {
    real orig[D][D], res[D][D];
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) orig[r][c] = (r == c) ? R(1.0) : R(0.0);
#pragma unroll
    for (int r = 0; r < D; r++) { orig[r][0] = self[r]; orig[r][1] = other[r]; }
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) res[r][c] = orig[r][c];
#pragma unroll
    for (int i = 1; i < D; i++) {
#pragma unroll
        for (int j = 0; j < i; j++) {
            real oc[D], rj[D];
#pragma unroll
            for (int r = 0; r < D; r++) { oc[r] = orig[r][i]; rj[r] = res[r][j]; }
            real dd = vdot<D>(rj, oc);
#pragma unroll
            for (int r = 0; r < D; r++) orig[r][i] = oc[r] - rj[r] * dd;
        }
        real col[D], ncol[D];
#pragma unroll
        for (int r = 0; r < D; r++) col[r] = orig[r][i];
        vnormalize<D>(col, ncol);
#pragma unroll
        for (int r = 0; r < D; r++) res[r][i] = ncol[r];
    }
    real rot[D][D];
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) rot[r][c] = (r == c) ? R(1.0) : R(0.0);
    real ca = eu_cos(angle), sa = eu_sin(angle);
    rot[0][0] = ca; rot[0][1] = -sa; rot[1][0] = sa; rot[1][1] = ca;
    real tmp[D][D], fin[D][D];
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) {
            real acc = R(0.0);
#pragma unroll
            for (int k = 0; k < D; k++) acc = acc + rot[r][k] * res[c][k];
            tmp[r][c] = acc;
        }
#pragma unroll
    for (int r = 0; r < D; r++)
#pragma unroll
        for (int c = 0; c < D; c++) {
            real acc = R(0.0);
#pragma unroll
            for (int k = 0; k < D; k++) acc = acc + res[r][k] * tmp[k][c];
            fin[r][c] = acc;
        }
    real out[D];
#pragma unroll
    for (int r = 0; r < D; r++) {
        real acc = R(0.0);
#pragma unroll
        for (int k = 0; k < D; k++) acc = acc + fin[r][k] * vec[k];
        out[r] = acc;
    }
#pragma unroll
    for (int r = 0; r < D; r++) vec[r] = out[r];
}

template <int D> EU_DEV void threshold_direction(const EuFlatSurface *F, HitCtx<D> &c, real *out) {
#pragma unroll
    for (int i = 0; i < D; i++) out[i] = c.dir[i];
    if (F->thr_kind == EU_THR_IDENTITY) return;                     /* surface.rs:258-266 */
    real normal[D];                                               /* surface.rs:268-288 */
#pragma unroll
    for (int i = 0; i < D; i++) normal[i] = -c.nc[i];
    const real from_theta = c.angle_dir_minus_nc();
    const real modifier = c.exiting ? F->thr_p0 : F->thr_p0_inv;
    const real to_theta = c.to_theta_for(modifier);
    const real angle_delta = to_theta - from_theta;
    general_rotation<D>(normal, c.dir, angle_delta, out);
}

/* the surface-colour providers, one function per kind (C: the provider's record).  Shared by the interpreter's program loop
 * (surface_color) and the scene-specialised kernels, which call them in post-order with the records as constants. */
EU_DEV Rgba col_uniform(const EuFlatColorOp *C) { return Rgba{(real)C->c0[0], (real)C->c0[1], (real)C->c0[2], (real)C->c0[3]}; }       /* surface.rs:424-429 */
EU_DEV Rgba col_blend(const EuFlatColorOp *C, const Rgba src, const Rgba dst) {                                                         /* surface.rs:295-322 */
    return (C->fn == EU_BL_RATIO) ? combine_palette_color(src, dst, C->v[0]) : blend_rgba(C->fn, src, dst);
}
template <int D> EU_DEV Rgba col_illum_global(const EuFlatColorOp *C, HitCtx<D> &c) {                                                   /* surface.rs:410-422 */
    real original_angle = c.angle_dir_nc();
    real angle = EU_PI_C - original_angle;
    real ratio = angle / EU_FRAC_PI_2_C;
    return combine_palette_color(Rgba{(real)C->c1[0], (real)C->c1[1], (real)C->c1[2], (real)C->c1[3]}, Rgba{(real)C->c0[0], (real)C->c0[1], (real)C->c0[2], (real)C->c0[3]}, ratio);
}
template <int D> EU_DEV Rgba col_illum_dir(const EuFlatColorOp *C, HitCtx<D> &c) {                                                      /* surface.rs:392-408 */
    real normal[D];
#pragma unroll
    for (int k = 0; k < D; k++) normal[k] = c.normal[k];
    if (c.ang_e > EU_FRAC_PI_2_C) {
#pragma unroll
        for (int k = 0; k < D; k++) normal[k] = -normal[k];
    }
    real nl[D];
#pragma unroll
    for (int k = 0; k < D; k++) nl[k] = C->v[k];            /* = -light_direction, negated at load */
    real angle = angle_between<D>(normal, nl);
    real ratio = R(1.0) - angle / EU_PI_C;
    return combine_palette_color(Rgba{(real)C->c1[0], (real)C->c1[1], (real)C->c1[2], (real)C->c1[3]}, Rgba{(real)C->c0[0], (real)C->c0[1], (real)C->c0[2], (real)C->c0[3]}, ratio);
}
template <int D> EU_DEV Rgba col_perlin(const EuFlatColorOp *C, const uint8_t *perm, HitCtx<D> &c, real time_s) {                       /* d3/entity/surface.rs:22-40 */
    real value = perlin4(perm, c.loc[0] / C->v[0], c.loc[1] / C->v[0], c.loc[D > 2 ? 2 : 0] / C->v[0], time_s * C->v[1]);
    Rgba v;
    hsv_to_rgb(value * R(360.0), R(1.0), R(1.0), v.r, v.g, v.b);
    v.a = R(1.0);
    return v;
}

/* the surface-colour provider tree, evaluated as a post-order program on a small stack */
template <int D>
EU_DEV Rgba surface_color(const EuScene &S, const EuFlatSurface *F, HitCtx<D> &c, real time_s, LaneCounters &cnt, real *cst, uint32_t stride) {
    /* stack of at most 4 colours in LDS, lane-interleaved (entry k, channel j of this lane at cst[(4 k + j) * stride]):
     * a run-time indexed private array would be scratch memory, and every scratch access waits on vmcnt */
    int sp = 0;
    for (uint32_t i = F->color_first; i <= F->color_root; i++) {
        const EuFlatColorOp *C = S.color_op(i);
        Rgba v;
        switch (C->kind) {
        case EU_COL_UNIFORM: v = col_uniform(C); break;
        case EU_COL_BLEND: {
            const real *pd = cst + (uint32_t)(((sp - 1) & 3) * 4) * stride, *ps = cst + (uint32_t)(((sp - 2) & 3) * 4) * stride;
            const Rgba dst = {pd[0], pd[stride], pd[2 * stride], pd[3 * stride]}, src = {ps[0], ps[stride], ps[2 * stride], ps[3 * stride]};
            sp -= 2;
            v = col_blend(C, src, dst);
            break;
        }
        case EU_COL_ILLUM_GLOBAL: v = col_illum_global<D>(C, c); break;
        case EU_COL_ILLUM_DIR: v = col_illum_dir<D>(C, c); break;
        case EU_COL_PERLIN: v = col_perlin<D>(C, S.perlin(C->aux), c, time_s); break;
        default: v = mapped_get_color(S, C->aux, c.loc, cnt); break;                          /* surface.rs:536-542 */
        }
        if (i == F->color_root) return v;        /* the root's value is the result (post-order: the stack is empty below it) */
        real *pw = cst + (uint32_t)((sp & 3) * 4) * stride;
        pw[0] = v.r; pw[stride] = v.g; pw[2 * stride] = v.b; pw[3 * stride] = v.a;
        sp++;
    }
    return Rgba{R(0.0), R(0.0), R(0.0), R(0.0)};     /* not reached: color_first <= color_root */
}

/* ------------------------------------------------------------------ what get_color asks of the hit surface
 * ComposableSurface::get_color up to its recursive calls (surface.rs:62-162): the clamped reflection ratio, then -- unless the
 * surface is a perfect mirror -- the surface colour, its u8 quantisation (get_intersection_color tests the QUANTISED alpha,
 * surface.rs:72-76), and for a translucent colour the transmission direction.  colour(): the surface-colour provider tree. */
template <int D> struct SurfaceEval {
    real ratio;             /* clamped to [0, 1] (surface.rs:145-147) */
    bool have_color;        /* ratio < 1: sc / spx are valid */
    Rgba sc; uint32_t spx;
    bool translucent;       /* quantised alpha != 255: thr[] is the transmission direction before the materials' transformations */
    real thr[D];
};
template <int D, class ColorFn>
EU_DEV void surface_eval(const EuFlatSurface *F, HitCtx<D> &c, LaneCounters &cnt, SurfaceEval<D> &E, ColorFn color) {
    real ratio = reflection_ratio<D>(F, c);
    E.ratio = rust_max(rust_min(ratio, R(1.0)), R(0.0));                               /* surface.rs:145-147 */
    E.have_color = false; E.translucent = false; E.spx = 0;
    E.sc = Rgba{R(0.0), R(0.0), R(0.0), R(0.0)};
    if (!(E.ratio >= R(1.0))) {                                                       /* get_intersection_color, surface.rs:62-117 */
        E.sc = color();
        E.have_color = true;
        E.spx = to_pixel4(E.sc, cnt);
        if ((E.spx >> 24) != 255u) { threshold_direction<D>(F, c, E.thr); E.translucent = true; }
    }
}

/* ------------------------------------------------------------------ the scene as the trace kernels see it
 * The kernels of trace_wavefront.h are templates over a scene policy P.  EuInterp<D> below is the interpreter: every question is
 * answered by walking the flat scene (shape programs, colour programs, RPN code).  A scene-specialised policy (generated and
 * compiled when the renderer is created, jit.cpp) answers the same questions with straight-line code for ONE scene; both call the
 * arithmetic of this header, so they cannot differ in a single bit. */
/* trace_closest's loop (universe/mod.rs:85-147) over the entities [e0, e1), their shape programs read from the flat scene: the whole
 * scene for the interpreter kernels; for the specialised kernels the entities whose programs did not fit the generator's budget of
 * straight-line code (jit.cpp) -- in entity order between the others, so that the strict minimum sees them in the reference's order. */
template <int D, class HS>
EU_DEV void interp_entities(const EuScene &S, uint32_t e0, uint32_t e1, const real *o, const real *d, HS &hs, LaneCounters &cnt, int use_box, bool &fail,
                            bool &have, real &best_t, uint32_t &best_code, uint32_t &best_ent) {
    for (uint32_t e = e0; e < e1; e++) {
        const EuScene::EntityView E = S.entity(e);
        if (E.surface < 0) continue;
        if (E.bound != 0xffffffffu && ray_misses_bound<D>(S.bounds(E.bound, D), o, d)) continue;
        real t = R(0.0); uint32_t code = 0;
        const uint32_t n = eval_shape<D>(S, E.shape_first, E.shape_root, o, d, hs, cnt, t, code, use_box, fail);
        if (n == 0) continue;
        if (!have || best_t > t) { have = true; best_t = t; best_code = code; best_ent = e; }
    }
}

template <int D> struct EuInterp {
    static constexpr bool kInterpreter = true;
    /* trace_closest (universe/mod.rs:85-147): first hit of every surfaced entity, strict minimum */
    template <class HS>
    static EU_DEV void trace_closest(const EuScene &S, const real *o, const real *d, HS &hs, LaneCounters &cnt, int use_box, bool &fail,
                                     bool &have, real &best_t, uint32_t &best_code, uint32_t &best_ent) {
        interp_entities<D>(S, 0u, S.n_entities, o, d, hs, cnt, use_box, fail, have, best_t, best_code, best_ent);
    }
    static EU_DEV void hit_normal(const EuScene &S, uint32_t ent, uint32_t code, const real *o, const real *d, const real *loc, real *n) {
        ::hit_normal<D>(S, code, o, d, loc, n);
    }
    static EU_DEV void surface(const EuScene &S, uint32_t ent, HitCtx<D> &c, real time_s, LaneCounters &cnt, real *cst, uint32_t stride, SurfaceEval<D> &E) {
        const EuScene::EntityView HE = S.entity(ent);
        const EuFlatSurface *F = S.surface((uint32_t)HE.surface);
        surface_eval<D>(F, c, cnt, E, [&]() { return surface_color<D>(S, F, c, time_s, cnt, cst, stride); });
    }
    static EU_DEV int material_at(const EuScene &S, const real *p) { return ::material_at<D>(S, p); }
    /* Material::enter / exit of the material of entity `ent` */
    static EU_DEV void material_apply(const EuScene &S, uint32_t ent, real *dir, bool exit_) { ::material_apply<D>(S, S.entity(ent).material, dir, exit_); }
    static EU_DEV Rgba background(const EuScene &S, const real *point, LaneCounters &cnt) { return mapped_get_color(S, S.background, point, cnt); }
};

#endif
