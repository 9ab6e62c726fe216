/*
 * eo_oracle.c -- ORACLE (test infrastructure, NOT product code).  See eo_oracle.h.
 *
 * Literal CPU restatement of the reference's trace loop.  It deliberately keeps the reference's
 * structure (lazy, cached intersection "providers"; one iterator state machine per CSG
 * operation; recursive trace) instead of the flattened/eager form the HIP kernel uses, so that
 * the parity tests compare two independently written implementations.
 * All citations are file:line under /root/reference/src/.
 */
#define _GNU_SOURCE
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ctype.h>

#include "eo_math.h"      /* the elementary functions work in f64 whatever F is: before the redefinition below */

/* The reference is generic over F: f64, or f32 with the cargo feature `low_precision` (Cargo.toml:18-20, main.rs:46-49).  The
 * f32 oracle (libeo_oracle_f32.so, -DEO_LOW_PRECISION) is this file with `double` redefined AFTER the system headers -- API
 * (eo_oracle.h, included below) and all: the Python loader talks float to that build.  R() types every literal of the path as
 * F; eo_f64 marks what stays 64-bit (LinearSpace expressions: meval evaluates in f64, material.rs:99-111); each elementary
 * function's result is rounded to F at once, so that no expression continues in double behind a call. */
typedef double eo_f64;
static inline eo_f64 eo_sqrt_f64(eo_f64 x) { return sqrt(x); }
static inline eo_f64 eo_fmod_f64(eo_f64 x, eo_f64 y) { return fmod(x, y); }
static inline eo_f64 eo_sin_f64(eo_f64 x) { return eo_sin(x); }
static inline eo_f64 eo_cos_f64(eo_f64 x) { return eo_cos(x); }
static inline eo_f64 eo_tan_f64(eo_f64 x) { return eo_tan(x); }
static inline eo_f64 eo_asin_f64(eo_f64 x) { return eo_asin(x); }
static inline eo_f64 eo_acos_f64(eo_f64 x) { return eo_acos(x); }
static inline eo_f64 eo_atan_f64(eo_f64 x) { return eo_atan(x); }
static inline eo_f64 eo_atan2_f64(eo_f64 y, eo_f64 x) { return eo_atan2(y, x); }
#define R(x) ((double)(x))
#ifdef EO_LOW_PRECISION
#define double float
#endif
#include "eo_oracle.h"

#define MAXD 4
#define EO_EPS128_A R(1.0e-6)   /* nalgebra 0.8.2 ApproxEq::approx_epsilon for f64 (UNVERIFIED) */
#define EO_PI_C R(3.14159265358979323846264338327950288)   /* std::f64::consts::PI */
#define EO_FRAC_PI_2_C R(1.57079632679489661923132169163975144)

/* ------------------------------------------------------------------ object model */

typedef struct { double r, g, b, a; } rgba_t;

enum { K_FREE = 0, K_SHAPE, K_MATERIAL, K_EXPR, K_TRANSFORM, K_RATIO, K_REFLDIR, K_THRDIR, K_BLEND,
       K_COLOR, K_UV, K_TEXTURE, K_MAPPED, K_SURFACE, K_ENTITY };

enum { SH_VOID, SH_SPHERE, SH_PLANE, SH_HALFSPACE, SH_CYLINDER, SH_COMPOSE };
enum { MAT_VACUUM, MAT_LINEAR };
enum { RATIO_UNIFORM, RATIO_FRESNEL };
enum { THR_IDENTITY, THR_SNELL };
enum { COL_UNIFORM, COL_BLEND, COL_ILLUM_GLOBAL, COL_ILLUM_DIR, COL_PERLIN, COL_TEXTURE };
enum { UV_SPHERE3, UV_DERANK };
enum { BL_OVER, BL_INSIDE, BL_OUTSIDE, BL_ATOP, BL_XOR, BL_PLUS, BL_MULTIPLY, BL_SCREEN, BL_OVERLAY,
       BL_DARKEN, BL_LIGHTEN, BL_DODGE, BL_BURN, BL_HARD_LIGHT, BL_SOFT_LIGHT, BL_DIFFERENCE,
       BL_EXCLUSION, BL_RATIO, BL_COUNT };
static const char *BLEND_NAMES[BL_COUNT] = { "over", "inside", "outside", "atop", "xor", "plus", "multiply",
    "screen", "overlay", "darken", "lighten", "dodge", "burn", "hard_light", "soft_light", "difference",
    "exclusion", "ratio" };

/* expression tree (meval 0.1.0 subset, material.rs:99-111) */
enum { EX_NUM, EX_VAR, EX_ADD, EX_SUB, EX_MUL, EX_DIV, EX_REM, EX_POW, EX_NEG, EX_FUNC };
enum { FN_SQRT, FN_ABS, FN_FLOOR, FN_CEIL, FN_MIN, FN_MAX, FN_SIN, FN_COS, FN_TAN, FN_ASIN, FN_ACOS,
       FN_ATAN, FN_ATAN2, FN_SIGNUM, FN_COUNT };
static const char *FN_NAMES[FN_COUNT] = { "sqrt", "abs", "floor", "ceil", "min", "max", "sin", "cos", "tan",
    "asin", "acos", "atan", "atan2", "signum" };
static const int FN_ARITY[FN_COUNT] = { 1, 1, 1, 1, 2, 2, 1, 1, 1, 1, 1, 1, 2, 1 };

typedef struct expr_node {
    int kind, fn;
    eo_f64 num;
    char var[16];
    struct expr_node *l, *r;
} expr_node;

typedef struct obj {
    int kind, sub;
    /* shapes */
    double a[MAXD], b[MAXD];   /* sphere centre / plane normal / cylinder centre ; cylinder axis */
    double r;                  /* radius or plane constant */
    double signum;
    int op;
    struct obj *sa, *sb;
    /* material */
    char legend[16];
    int n_children;
    struct obj **children;     /* LinearSpace -> transformations ; transformation -> exprs */
    expr_node *ex, *ex_inv;    /* K_EXPR */
    /* providers */
    double p0, p1;             /* ratio / refractive indices / blend ratio / perlin size,speed */
    rgba_t c0, c1;             /* uniform colour / light,dark */
    double dir[MAXD];
    struct obj *o0, *o1, *o2, *o3; /* generic links (see constructors) */
    uint8_t perm[512];
    /* texture */
    uint32_t w, h;
    uint8_t *pixels;
} obj;

struct eo_scene {
    int dim;
    obj **objs;
    int n_objs, cap_objs;
    obj **entities;
    int n_entities;
    obj *background;
    eo_camera camera;
    char err[512];
};

static int fail(eo_scene *s, const char *msg) { snprintf(s->err, sizeof s->err, "%s", msg); return -1; }

static obj *new_obj(eo_scene *s, int kind, int sub, int *handle) {
    if (s->n_objs == s->cap_objs) {
        s->cap_objs = s->cap_objs ? s->cap_objs * 2 : 64;
        s->objs = realloc(s->objs, sizeof(obj *) * (size_t)s->cap_objs);
    }
    obj *o = calloc(1, sizeof(obj));
    o->kind = kind; o->sub = sub;
    *handle = s->n_objs;
    s->objs[s->n_objs++] = o;
    return o;
}
static obj *get_obj(const eo_scene *s, int h, int kind) {
    if (h < 0 || h >= s->n_objs) return NULL;
    obj *o = s->objs[h];
    return (o->kind == kind) ? o : NULL;
}

eo_scene *eo_scene_new(int dim) {
    if (dim < 2 || dim > MAXD) return NULL;
    eo_scene *s = calloc(1, sizeof *s);
    s->dim = dim;
    eo_default_camera(dim, NULL, &s->camera);
    return s;
}
static void free_expr(expr_node *e) { if (!e) return; free_expr(e->l); free_expr(e->r); free(e); }
void eo_scene_free(eo_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->n_objs; i++) {
        obj *o = s->objs[i];
        free(o->children); free(o->pixels); free_expr(o->ex); free_expr(o->ex_inv); free(o);
    }
    free(s->objs); free(s->entities); free(s);
}
const char *eo_last_error(const eo_scene *s) { return s->err; }

/* ------------------------------------------------------------------ flop accounting (-DEO_COUNT_FLOPS builds only)
 * f64 operations of the reference ALGORITHM per ray, counted where this restatement performs them: FL = add /
 * subtract / multiply (1 each; negation, abs, floor, compares and selects are not counted), FLD = divide (and fmod), FLS = sqrt,
 * FLT = calls of acos / asin / sin / cos / tan / atan / atan2.  The counters are thread-local and summed into eo_stats.flops[4]
 * by eo_render; the shipped oracle (the timed CPU baseline) is built without them. */
#ifdef EO_USE_LIBM      /* variant build: the platform libm (what Rust's f64::acos etc. call) instead of eo_math.h -- measures how far
                         * a <= 1 ulp difference in the elementary functions moves the rendered bytes (tests/test_oracle_libm.py) */
#ifdef EO_LOW_PRECISION      /* F = f32 with the platform libm: the f32 entry points -- what the reference's `low_precision` binary calls (F::acos with F = f32,
                              * util.rs:712-722, surface.rs:225,280; Cargo.toml:18-20) */
#define EO_FN(name) name##f
#else
#define EO_FN(name) name
#endif
#else
#define EO_FN(name) eo_##name
#endif
static unsigned long long eo_fl_total[4];
static pthread_mutex_t eo_fl_mu = PTHREAD_MUTEX_INITIALIZER;
#ifdef EO_COUNT_FLOPS
static __thread unsigned long long eo_fl[4];
#define FL(n) (eo_fl[0] += (unsigned long long)(n))
#define FLD(n) (eo_fl[1] += (unsigned long long)(n))
#define FLS(n) (eo_fl[2] += (unsigned long long)(n))
#define FLT(n) (eo_fl[3] += (unsigned long long)(n))
static inline eo_f64 eo_cnt_t_(eo_f64 v) { FLT(1); return v; }
static inline eo_f64 eo_cnt_s_(eo_f64 v) { FLS(1); return v; }
static inline eo_f64 eo_cnt_d_(eo_f64 v) { FLD(1); return v; }
#else
#define FL(n) ((void)0)
#define FLD(n) ((void)0)
#define FLS(n) ((void)0)
#define FLT(n) ((void)0)
#define eo_cnt_t_(v) (v)
#define eo_cnt_s_(v) (v)
#define eo_cnt_d_(v) (v)
#endif
/* every elementary function of the path: evaluated in f64 (eo_math.h or libm), counted, rounded to F at once */
#if defined(EO_LOW_PRECISION) && !defined(EO_USE_LIBM)      /* F = f32: the 1-ulp f64 routines, rounded to f32 (eo_math.h: correctly rounded f32 but for one argument in 2^28) */
#define eo_acos(x) ((double)eo_cnt_t_(eo_acos32(x)))
#define eo_asin(x) ((double)eo_cnt_t_(eo_asin32(x)))
#define eo_sin(x) ((double)eo_cnt_t_(eo_sin32(x)))
#define eo_cos(x) ((double)eo_cnt_t_(eo_cos32(x)))
#else
#define eo_acos(x) ((double)eo_cnt_t_(EO_FN(acos)(x)))
#define eo_asin(x) ((double)eo_cnt_t_(EO_FN(asin)(x)))
#define eo_sin(x) ((double)eo_cnt_t_(EO_FN(sin)(x)))
#define eo_cos(x) ((double)eo_cnt_t_(EO_FN(cos)(x)))
#endif
#define eo_tan(x) ((double)eo_cnt_t_(EO_FN(tan)(x)))
#define eo_atan(x) ((double)eo_cnt_t_(EO_FN(atan)(x)))
#define eo_atan2(y, x) ((double)eo_cnt_t_(EO_FN(atan2)(y, x)))
#define sqrt(x) ((double)eo_cnt_s_(sqrt(x)))
#define fmod(x, y) ((double)eo_cnt_d_(fmod(x, y)))

/* ------------------------------------------------------------------ vectors (nalgebra 0.8.2) */
/* dot / norm summation order x -> w (UNVERIFIED for nalgebra 0.8.2) */
static double v_dot(int D, const double *a, const double *b) {
    FL(2 * D - 1);
    double s = a[0] * b[0];
    for (int i = 1; i < D; i++) s = s + a[i] * b[i];
    return s;
}
static double v_nsq(int D, const double *a) { return v_dot(D, a, a); }
static double v_norm(int D, const double *a) { return sqrt(v_nsq(D, a)); }
static void v_normalize(int D, const double *a, double *out) {       /* v / |v| */
    double n = v_norm(D, a);
    FLD(D);
    for (int i = 0; i < D; i++) out[i] = a[i] / n;
}
static void v_sub(int D, const double *a, const double *b, double *o) { FL(D); for (int i = 0; i < D; i++) o[i] = a[i] - b[i]; }
static void v_add(int D, const double *a, const double *b, double *o) { FL(D); for (int i = 0; i < D; i++) o[i] = a[i] + b[i]; }
static void v_scale(int D, const double *a, double k, double *o) { FL(D); for (int i = 0; i < D; i++) o[i] = a[i] * k; }
static void v_neg(int D, const double *a, double *o) { for (int i = 0; i < D; i++) o[i] = -a[i]; }
static void v_copy(int D, const double *a, double *o) { for (int i = 0; i < D; i++) o[i] = a[i]; }
static void v_cross3(const double *a, const double *b, double *o) {
    FL(9);
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}

/* util.rs:712-722 */
static double angle_between(int D, const double *a, const double *b) {
    FL(1); FLD(1);
    double result = eo_acos(v_dot(D, a, b) / (v_norm(D, a) * v_norm(D, b)));
    return eo_isnan(result) ? R(0.0) : result;
}

/* Rust f64::signum: NaN -> NaN, +-0 -> +-1 */
static double rust_signum(double x) { if (eo_isnan(x)) return x; return signbit(x) ? -R(1.0) : R(1.0); }
static double rust_min(double a, double b) { if (eo_isnan(a)) return b; if (eo_isnan(b)) return a; return a < b ? a : b; }
static double rust_max(double a, double b) { if (eo_isnan(a)) return b; if (eo_isnan(b)) return a; return a > b ? a : b; }

/* util.rs:287-299 */
static double remainder_f(double a, double b) {
    double rem = fmod(a, b);
    if (rem == R(0.0)) return R(0.0);
    if (a < R(0.0)) return b + rem;
    return rem;
}
static int64_t remainder_i(int64_t a, int64_t b) {
    int64_t rem = a % b;
    if (rem == 0) return 0;
    if (a < 0) return b + rem;
    return rem;
}

/* ------------------------------------------------------------------ shapes: constructors */

int eo_shape_void(eo_scene *s) { int h; new_obj(s, K_SHAPE, SH_VOID, &h); return h; }

int eo_shape_sphere(eo_scene *s, const double *center, double radius) {   /* shape.rs:643-649 */
    int h; obj *o = new_obj(s, K_SHAPE, SH_SPHERE, &h);
    v_copy(s->dim, center, o->a); o->r = radius;
    return h;
}

int eo_shape_hyperplane(eo_scene *s, const double *normal, double constant) {   /* shape.rs:750-759 */
    if (!(v_nsq(s->dim, normal) > R(0.0))) return fail(s, "Cannot have a normal with length of 0.");
    int h; obj *o = new_obj(s, K_SHAPE, SH_PLANE, &h);
    v_copy(s->dim, normal, o->a); o->r = constant;
    return h;
}
int eo_shape_hyperplane_with_point(eo_scene *s, const double *normal, const double *point) {  /* shape.rs:761-766 */
    double constant = -v_dot(s->dim, normal, point);
    return eo_shape_hyperplane(s, normal, constant);
}
int eo_shape_hyperplane_with_vectors(eo_scene *s, const double *a, const double *b, const double *point) { /* :768-776 */
    if (s->dim != 3) return fail(s, "new_with_vectors needs a cross product (3-D only)");
    double n[MAXD] = {0};
    v_cross3(a, b, n);
    return eo_shape_hyperplane_with_point(s, n, point);
}
int eo_shape_halfspace(eo_scene *s, int plane, double sign) {   /* shape.rs:828-835 */
    obj *p = get_obj(s, plane, K_SHAPE);
    if (!p || p->sub != SH_PLANE) return fail(s, "Invalid type, expected a `Hyperplane`.");
    int h; obj *o = new_obj(s, K_SHAPE, SH_HALFSPACE, &h);
    v_copy(s->dim, p->a, o->a); o->r = p->r;
    o->signum = sign / fabs(sign);
    return h;
}
int eo_shape_halfspace_with_point(eo_scene *s, int plane, const double *point) {   /* shape.rs:837-841 */
    obj *p = get_obj(s, plane, K_SHAPE);
    if (!p || p->sub != SH_PLANE) return fail(s, "Invalid type, expected a `Hyperplane`.");
    double identifier = v_dot(s->dim, p->a, point) + p->r;
    return eo_shape_halfspace(s, plane, identifier);
}

int eo_shape_composable_of(eo_scene *s, const int *shapes, int n, int op) {   /* shape.rs:523-545 */
    if (n < 2) return fail(s, "2 or more `Shape`s are needed to construct a `ComposableShape`.");
    if (op < 0 || op > 3) return fail(s, "Invalid `SetOperation`");
    for (int i = 0; i < n; i++) if (!get_obj(s, shapes[i], K_SHAPE)) return fail(s, "not a shape");
    int h = shapes[0];
    for (int i = 1; i < n; i++) {
        int nh; obj *o = new_obj(s, K_SHAPE, SH_COMPOSE, &nh);
        o->op = op; o->sa = s->objs[h]; o->sb = s->objs[shapes[i]];
        h = nh;
    }
    return h;
}

/* d3/entity/shape.rs:17-66 (cuboid), d4/entity/shape.rs:18-76 (hypercuboid) */
int eo_shape_cuboid(eo_scene *s, const double *center, const double *abc) {
    int D = s->dim;
    if (D != 3 && D != 4) return fail(s, "cuboid: 3-D or 4-D only");
    double half[MAXD];
    for (int i = 0; i < D; i++) half[i] = abc[i] / R(2.0);
    int shapes[8], n = 0;
    double axis[MAXD][MAXD] = {{0}};
    for (int i = 0; i < D; i++) axis[i][i] = R(1.0);
    for (int ax = 0; ax < D; ax++) {
        for (int sgn = 0; sgn < 2; sgn++) {
            double off[MAXD], pt[MAXD];
            for (int i = 0; i < D; i++) off[i] = axis[ax][i] * half[i];     /* component-wise x * half_abc */
            if (sgn) v_neg(D, off, off);
            v_add(D, center, off, pt);                                       /* na::translate(&v, &center) */
            int plane;
            if (D == 3) {
                /* new_with_vectors(&y,&z) for x; (&x,&z) for y; (&x,&y) for z */
                const double *va = (ax == 0) ? axis[1] : axis[0];
                const double *vb = (ax == 2) ? axis[1] : axis[2];
                plane = eo_shape_hyperplane_with_vectors(s, va, vb, pt);
            } else {
                double nn[MAXD];
                v_normalize(D, axis[ax], nn);
                plane = eo_shape_hyperplane_with_point(s, nn, pt);
            }
            if (plane < 0) return plane;
            int hs = eo_shape_halfspace_with_point(s, plane, center);
            if (hs < 0) return hs;
            shapes[n++] = hs;
        }
    }
    return eo_shape_composable_of(s, shapes, n, EO_OP_INTERSECTION);
}

int eo_shape_cylinder(eo_scene *s, const double *center, const double *direction, double radius) {   /* shape.rs:893-904 */
    if (!(v_nsq(s->dim, direction) > R(0.0))) return fail(s, "Cannot have a direction with length of 0.");
    if (!(radius > R(0.0))) return fail(s, "The radius must be positive.");
    int h; obj *o = new_obj(s, K_SHAPE, SH_CYLINDER, &h);
    v_copy(s->dim, center, o->a);
    v_normalize(s->dim, direction, o->b);
    o->r = radius;
    return h;
}
int eo_shape_cylinder_with_height(eo_scene *s, const double *center, const double *direction, double radius, double height) { /* :906-927 */
    int D = s->dim;
    double nd[MAXD], off[MAXD], pt[MAXD];
    v_normalize(D, direction, nd);
    double half_height = height / (R(1.0) + R(1.0));
    int shapes[3];
    shapes[0] = eo_shape_cylinder(s, center, direction, radius);
    if (shapes[0] < 0) return shapes[0];
    v_scale(D, nd, half_height, off); v_add(D, center, off, pt);
    int p1 = eo_shape_hyperplane_with_point(s, nd, pt); if (p1 < 0) return p1;
    shapes[1] = eo_shape_halfspace_with_point(s, p1, center); if (shapes[1] < 0) return shapes[1];
    v_scale(D, nd, -half_height, off); v_add(D, center, off, pt);
    int p2 = eo_shape_hyperplane_with_point(s, nd, pt); if (p2 < 0) return p2;
    shapes[2] = eo_shape_halfspace_with_point(s, p2, center); if (shapes[2] < 0) return shapes[2];
    return eo_shape_composable_of(s, shapes, 3, EO_OP_INTERSECTION);
}

/* ------------------------------------------------------------------ shapes: is_point_inside */

static void cyl_closest_point_on_axis(int D, const obj *c, const double *to, double *out) {   /* shape.rs:929-932 */
    double d[MAXD], sc[MAXD];
    v_sub(D, to, c->a, d);
    v_scale(D, c->b, v_dot(D, c->b, d), sc);
    v_add(D, c->a, sc, out);
}

static int shape_inside(int D, const obj *sh, const double *p) {
    switch (sh->sub) {
    case SH_VOID: return 1;                                                   /* shape.rs:616-618 */
    case SH_SPHERE: {                                                         /* shape.rs:735-737 */
        double d[MAXD]; v_sub(D, sh->a, p, d);
        FL(1);
        return v_nsq(D, d) <= sh->r * sh->r;
    }
    case SH_PLANE: return 0;                                                  /* shape.rs:814-816 */
    case SH_HALFSPACE: {                                                      /* shape.rs:874-880 */
        FL(1);
        double result = v_dot(D, sh->a, p) + sh->r;
        return sh->signum == rust_signum(result);
    }
    case SH_CYLINDER: {                                                       /* shape.rs:1032-1037 */
        double q[MAXD], v[MAXD];
        cyl_closest_point_on_axis(D, sh, p, q);
        v_sub(D, p, q, v);
        FL(1);
        return v_nsq(D, v) <= sh->r * sh->r;
    }
    default: {                                                                /* shape.rs:589-600 */
        switch (sh->op) {
        case EO_OP_UNION: return shape_inside(D, sh->sa, p) || shape_inside(D, sh->sb, p);
        case EO_OP_INTERSECTION: return shape_inside(D, sh->sa, p) && shape_inside(D, sh->sb, p);
        case EO_OP_COMPLEMENT: return shape_inside(D, sh->sa, p) && !shape_inside(D, sh->sb, p);
        default: return shape_inside(D, sh->sa, p) ^ shape_inside(D, sh->sb, p);
        }
    }
    }
}

/* ------------------------------------------------------------------ per-thread context + arena */

typedef struct {
    const eo_scene *scene;
    int D;
    char *arena; size_t arena_used, arena_cap;
    eo_stats stats;
    uint64_t time_ms;
} tctx;

#define EO_ARENA_BYTES ((size_t)64 << 20)    /* per thread, touched only as far as used (runaway CSG streams stop at CSG_GUARD) */
static void *arena_alloc(tctx *t, size_t n) {
    n = (n + 15) & ~(size_t)15;
    if (t->arena_used + n > t->arena_cap) { fprintf(stderr, "eo_oracle: arena exhausted\n"); abort(); }
    void *p = t->arena + t->arena_used;
    t->arena_used += n;
    memset(p, 0, n);
    return p;
}

/* ------------------------------------------------------------------ intersection providers */

typedef eo_intersection hit_t;

/* Provider<T> = lazily filled cache (including the terminating None) over an iterator,
 * util.rs:356-450.  The iterator is either an immediate list (leaf shapes) or one of the four
 * CSG state machines (shape.rs:188-497). */
typedef struct provider {
    hit_t *items; unsigned char *some; int n_items, cap_items;
    int is_csg;
    hit_t imm[2]; int n_imm, imm_pos;
    int op; const obj *shape_a, *shape_b; struct provider *pa, *pb; int index_a, index_b;
} provider;

static provider *universe_intersect(tctx *t, const double *loc, const double *dir, const obj *shape);
static int iter_next(tctx *t, provider *p, hit_t *out);

static int provider_get(tctx *t, provider *p, int index, hit_t *out) {   /* util.rs:393-419 */
    while (index >= p->n_items) {
        if (p->n_items == p->cap_items) {
            int ncap = p->cap_items ? p->cap_items * 2 : 4;
            hit_t *ni = arena_alloc(t, sizeof(hit_t) * (size_t)ncap);
            unsigned char *ns = arena_alloc(t, (size_t)ncap);
            if (p->n_items) { memcpy(ni, p->items, sizeof(hit_t) * (size_t)p->n_items); memcpy(ns, p->some, (size_t)p->n_items); }
            p->items = ni; p->some = ns; p->cap_items = ncap;
        }
        hit_t h;
        int some = iter_next(t, p, &h);
        p->some[p->n_items] = (unsigned char)some;
        if (some) p->items[p->n_items] = h;
        p->n_items++;
    }
    if (p->some[index]) { *out = p->items[index]; return 1; }
    return 0;
}

static void negate_normal(int D, hit_t *h) { for (int i = 0; i < D; i++) h->normal[i] = -h->normal[i]; }

#define CSG_GUARD 10000       /* iterations of one next(): the reference would spin forever; counted in stats.errors */

static int iter_next(tctx *t, provider *p, hit_t *out) {
    int D = t->D;
    if (!p->is_csg) {
        if (p->imm_pos < p->n_imm) { *out = p->imm[p->imm_pos++]; return 1; }
        return 0;
    }
    hit_t a, b;
    switch (p->op) {
    case EO_OP_UNION:                                                     /* shape.rs:212-264 */
        for (int guard = 0;; guard++) {
            if (guard > CSG_GUARD) { t->stats.errors++; t->stats.spins++; return 0; }
            int sa = provider_get(t, p->pa, p->index_a, &a);
            int sb = provider_get(t, p->pb, p->index_b, &b);
            if (sa) {
                if (sb) {
                    if (a.distance < b.distance) {
                        if (!shape_inside(D, p->shape_b, a.location)) { p->index_a++; *out = a; return 1; }
                        p->index_a++;
                    } else {
                        if (!shape_inside(D, p->shape_a, b.location)) { p->index_b++; *out = b; return 1; }
                        p->index_b++;
                    }
                } else {
                    p->index_a++;
                    if (shape_inside(D, p->shape_b, a.location)) return 0;
                    *out = a; return 1;
                }
            } else {
                if (sb) {
                    p->index_b++;
                    if (shape_inside(D, p->shape_a, b.location)) return 0;
                    *out = b; return 1;
                }
                return 0;
            }
        }
    case EO_OP_INTERSECTION:                                              /* shape.rs:291-340 */
        for (int guard = 0;; guard++) {
            if (guard > CSG_GUARD) { t->stats.errors++; t->stats.spins++; return 0; }
            int sa = provider_get(t, p->pa, p->index_a, &a);
            int sb = provider_get(t, p->pb, p->index_b, &b);
            if (sa) {
                if (sb) {
                    if (a.distance < b.distance) {
                        p->index_a++;
                        if (shape_inside(D, p->shape_b, a.location)) { *out = a; return 1; }
                    } else {
                        p->index_b++;
                        if (shape_inside(D, p->shape_a, b.location)) { *out = b; return 1; }
                    }
                } else {
                    p->index_a++;
                    if (shape_inside(D, p->shape_b, a.location)) { *out = a; return 1; }
                    return 0;
                }
            } else {
                if (sb) {
                    p->index_b++;
                    if (shape_inside(D, p->shape_a, b.location)) { *out = b; return 1; }
                }
                return 0;
            }
        }
    case EO_OP_COMPLEMENT:                                                /* shape.rs:365-409 */
        for (int guard = 0;; guard++) {
            if (guard > CSG_GUARD) { t->stats.errors++; t->stats.spins++; return 0; }
            int sa = provider_get(t, p->pa, p->index_a, &a);
            int sb = provider_get(t, p->pb, p->index_b, &b);
            if (sa) {
                if (sb) {
                    if (a.distance < b.distance) {
                        p->index_a++;
                        if (!shape_inside(D, p->shape_b, a.location)) { *out = a; return 1; }
                    } else {
                        p->index_b++;
                        if (shape_inside(D, p->shape_a, b.location)) { negate_normal(D, &b); *out = b; return 1; }
                    }
                } else {
                    *out = a; return 1;      /* NB: index_a is NOT advanced (shape.rs:390-392) */
                }
            } else {
                if (sb) {
                    p->index_b++;
                    if (shape_inside(D, p->shape_a, b.location)) { negate_normal(D, &b); *out = b; return 1; }
                }
                return 0;
            }
        }
    default: {                                                            /* shape.rs:436-496 (no loop) */
        int sa = provider_get(t, p->pa, p->index_a, &a);
        int sb = provider_get(t, p->pb, p->index_b, &b);
        if (sa) {
            if (sb) {
                if (a.distance < b.distance) {
                    p->index_a++;
                    if (shape_inside(D, p->shape_b, a.location)) negate_normal(D, &a);
                    *out = a; return 1;
                } else {
                    p->index_b++;
                    if (shape_inside(D, p->shape_a, b.location)) negate_normal(D, &b);
                    *out = b; return 1;
                }
            } else {
                p->index_a++;
                if (shape_inside(D, p->shape_b, a.location)) negate_normal(D, &a);
                *out = a; return 1;
            }
        } else {
            if (sb) {
                p->index_b++;
                if (shape_inside(D, p->shape_a, b.location)) negate_normal(D, &b);
                *out = b; return 1;
            }
            return 0;
        }
    }
    }
}

static void make_hit(int D, const double *loc, const double *dir, double tt, hit_t *h) {
    double rv[MAXD];
    v_scale(D, dir, tt, rv);
    v_add(D, loc, rv, h->location);
    v_copy(D, dir, h->direction);
    h->distance = tt;
}

/* quadratic shared by sphere (shape.rs:667-693) and cylinder (shape.rs:962-988) */
static int quad_roots(double a, double b, double c, double *t_first, double *t_second, int *has_second) {
    FL(4);
    double d = b * b - R(4.0) * a * c;
    if (d < R(0.0)) return 0;
    FL(4); FLD(2);
    double d_sqrt = sqrt(d);
    double t1 = (-b - d_sqrt) / (R(2.0) * a);
    double t2 = (-b + d_sqrt) / (R(2.0) * a);
    int has_first = 0; *has_second = 0;
    if (t1 >= R(0.0)) {
        *t_first = t1; has_first = 1;
        if (t2 >= R(0.0)) { *t_second = t2; *has_second = 1; }
    } else if (t2 >= R(0.0)) {
        *t_first = t2; has_first = 1;
    }
    return has_first;
}

/* Universe::intersect (universe/mod.rs:61-83); the (material, shape) table maps Vacuum and
 * LinearSpace to the same routines (d3/mod.rs:35-60), so the material plays no role here. */
static provider *universe_intersect(tctx *t, const double *loc, const double *dir, const obj *sh) {
    int D = t->D;
    provider *p = arena_alloc(t, sizeof *p);
    switch (sh->sub) {
    case SH_VOID: break;                                                  /* shape.rs:622-631 */
    case SH_SPHERE: {                                                     /* shape.rs:652-731 */
        double rel[MAXD]; v_sub(D, loc, sh->a, rel);
        double a = v_nsq(D, dir);
        FL(3);
        double b = R(2.0) * v_dot(D, dir, rel);
        double c = v_nsq(D, rel) - sh->r * sh->r;
        double t1, t2; int has2;
        if (!quad_roots(a, b, c, &t1, &t2, &has2)) break;
        double tt[2] = { t1, t2 };
        for (int k = 0; k < 1 + has2; k++) {
            hit_t *h = &p->imm[p->n_imm++];
            make_hit(D, loc, dir, tt[k], h);
            double n[MAXD]; v_sub(D, h->location, sh->a, n);
            v_normalize(D, n, h->normal);
        }
        break;
    }
    case SH_PLANE: case SH_HALFSPACE: {                                   /* shape.rs:779-809, 843-870 */
        FL(1); FLD(1);
        double tt = -(v_dot(D, sh->a, loc) + sh->r) / v_dot(D, sh->a, dir);
        if (tt < R(0.0)) break;
        hit_t *h = &p->imm[p->n_imm++];
        make_hit(D, loc, dir, tt, h);
        v_copy(D, sh->a, h->normal);
        if (sh->sub == SH_HALFSPACE) { double k = -sh->signum; FL(D); for (int i = 0; i < D; i++) h->normal[i] *= k; }
        break;
    }
    case SH_CYLINDER: {                                                   /* shape.rs:935-1027 */
        double tmp[MAXD], a_vec[MAXD], delta[MAXD], c_vec[MAXD];
        v_scale(D, sh->b, v_dot(D, dir, sh->b), tmp); v_sub(D, dir, tmp, a_vec);
        v_sub(D, loc, sh->a, delta);
        v_scale(D, sh->b, v_dot(D, delta, sh->b), tmp); v_sub(D, delta, tmp, c_vec);
        double a = v_nsq(D, a_vec);
        FL(4);
        double b = (R(1.0) + R(1.0)) * v_dot(D, a_vec, c_vec);
        double c = v_nsq(D, c_vec) - sh->r * sh->r;
        double t1, t2; int has2;
        if (!quad_roots(a, b, c, &t1, &t2, &has2)) break;
        hit_t *h1 = &p->imm[p->n_imm++];
        make_hit(D, loc, dir, t1, h1);
        double axis_pt[MAXD], n[MAXD];
        cyl_closest_point_on_axis(D, sh, h1->location, axis_pt);   /* from hit 1, reused for hit 2 (shape.rs:999,1017) */
        v_sub(D, h1->location, axis_pt, n); v_normalize(D, n, h1->normal);
        if (has2) {
            hit_t *h2 = &p->imm[p->n_imm++];
            make_hit(D, loc, dir, t2, h2);
            v_sub(D, h2->location, axis_pt, n); v_normalize(D, n, h2->normal);
        }
        break;
    }
    default:                                                              /* shape.rs:548-584 */
        p->is_csg = 1; p->op = sh->op; p->shape_a = sh->sa; p->shape_b = sh->sb;
        p->pa = universe_intersect(t, loc, dir, sh->sa);
        p->pb = universe_intersect(t, loc, dir, sh->sb);
        break;
    }
    return p;
}

/* ------------------------------------------------------------------ materials */

int eo_material_vacuum(eo_scene *s) { int h; new_obj(s, K_MATERIAL, MAT_VACUUM, &h); return h; }

/* expression parser: numbers, identifiers, + - * / % ^, unary +-, parentheses, a few functions.
 * Precedence as in meval 0.1.0's shunting yard (UNVERIFIED): +- (1, left) < * / % (2, left) <
 * unary (3) < ^ (4, right). */
typedef struct { const char *s; int pos; int ok; } eparser;
static void ep_ws(eparser *p) { while (isspace((unsigned char)p->s[p->pos])) p->pos++; }
static expr_node *ep_expr(eparser *p);
static expr_node *ep_unary(eparser *p);
static expr_node *mk(int kind, expr_node *l, expr_node *r) { expr_node *e = calloc(1, sizeof *e); e->kind = kind; e->l = l; e->r = r; return e; }

static expr_node *ep_atom(eparser *p) {
    ep_ws(p);
    char c = p->s[p->pos];
    if (c == '(') {
        p->pos++;
        expr_node *e = ep_expr(p);
        ep_ws(p);
        if (p->s[p->pos] != ')') { p->ok = 0; return e; }
        p->pos++;
        return e;
    }
    if (isdigit((unsigned char)c) || c == '.') {
        char *end;
        eo_f64 v = strtod(p->s + p->pos, &end);
        if (end == p->s + p->pos) { p->ok = 0; return NULL; }
        p->pos = (int)(end - p->s);
        expr_node *e = mk(EX_NUM, NULL, NULL); e->num = v; return e;
    }
    if (isalpha((unsigned char)c) || c == '_') {
        char name[16]; int n = 0;
        while ((isalnum((unsigned char)p->s[p->pos]) || p->s[p->pos] == '_') && n < 15) name[n++] = p->s[p->pos++];
        name[n] = 0;
        ep_ws(p);
        if (p->s[p->pos] == '(') {
            int fn = -1;
            for (int i = 0; i < FN_COUNT; i++) if (!strcmp(name, FN_NAMES[i])) fn = i;
            if (fn < 0) { p->ok = 0; return NULL; }
            p->pos++;
            expr_node *a0 = ep_expr(p), *a1 = NULL;
            ep_ws(p);
            if (FN_ARITY[fn] == 2) {
                if (p->s[p->pos] != ',') { p->ok = 0; free_expr(a0); return NULL; }
                p->pos++;
                a1 = ep_expr(p); ep_ws(p);
            }
            if (p->s[p->pos] != ')') { p->ok = 0; free_expr(a0); free_expr(a1); return NULL; }
            p->pos++;
            expr_node *e = mk(EX_FUNC, a0, a1); e->fn = fn; return e;
        }
        if (!strcmp(name, "pi")) { expr_node *e = mk(EX_NUM, NULL, NULL); e->num = 3.14159265358979323846264338327950288; return e; }
        if (!strcmp(name, "e")) { expr_node *e = mk(EX_NUM, NULL, NULL); e->num = 2.71828182845904523536028747135266250; return e; }
        expr_node *e = mk(EX_VAR, NULL, NULL); strcpy(e->var, name); return e;
    }
    p->ok = 0;
    return NULL;
}
static expr_node *ep_power(eparser *p) {
    expr_node *base = ep_atom(p);
    ep_ws(p);
    if (p->ok && p->s[p->pos] == '^') { p->pos++; expr_node *ex = ep_unary(p); return mk(EX_POW, base, ex); }
    return base;
}
static expr_node *ep_unary(eparser *p) {
    ep_ws(p);
    if (p->s[p->pos] == '-') { p->pos++; return mk(EX_NEG, ep_unary(p), NULL); }
    if (p->s[p->pos] == '+') { p->pos++; return ep_unary(p); }
    return ep_power(p);
}
static expr_node *ep_term(eparser *p) {
    expr_node *l = ep_unary(p);
    for (;;) {
        ep_ws(p);
        char c = p->s[p->pos];
        if (!p->ok || (c != '*' && c != '/' && c != '%')) return l;
        p->pos++;
        expr_node *r = ep_unary(p);
        l = mk(c == '*' ? EX_MUL : c == '/' ? EX_DIV : EX_REM, l, r);
    }
}
static expr_node *ep_expr(eparser *p) {
    expr_node *l = ep_term(p);
    for (;;) {
        ep_ws(p);
        char c = p->s[p->pos];
        if (!p->ok || (c != '+' && c != '-')) return l;
        p->pos++;
        expr_node *r = ep_term(p);
        l = mk(c == '+' ? EX_ADD : EX_SUB, l, r);
    }
}
static expr_node *parse_expr(const char *s) {
    eparser p = { s, 0, 1 };
    expr_node *e = ep_expr(&p);
    ep_ws(&p);
    if (!p.ok || s[p.pos] != 0 || !e) { free_expr(e); return NULL; }
    return e;
}

/* LinearSpace expressions are evaluated in f64 whatever F is (meval; the result is cast to F, material.rs:99-111) */
static eo_f64 expr_min(eo_f64 a, eo_f64 b) { if (a != a) return b; if (b != b) return a; return a < b ? a : b; }
static eo_f64 expr_max(eo_f64 a, eo_f64 b) { if (a != a) return b; if (b != b) return a; return a > b ? a : b; }
static eo_f64 expr_signum(eo_f64 x) { if (x != x) return x; return signbit(x) ? -1.0 : 1.0; }
/* x^y restricted to integral |y| <= 64 (repeated multiplication; meval uses powf -- deviation) */
static eo_f64 pow_int(eo_f64 x, eo_f64 y) {
    if (!(y == floor(y)) || fabs(y) > 64.0) return NAN;
    int n = (int)fabs(y);
    eo_f64 r = 1.0;
    FL(n);
    for (int i = 0; i < n; i++) r = r * x;
    if (y < 0.0) FLD(1);
    return (y < 0.0) ? 1.0 / r : r;
}

static eo_f64 eval_expr(const expr_node *e, const char *legend, const eo_f64 *ctx, int D, int *err) {
    switch (e->kind) {
    case EX_NUM: return e->num;
    case EX_VAR:
        if (strlen(e->var) == 1) for (int i = 0; i < D && legend[i]; i++) if (legend[i] == e->var[0]) return ctx[i];
        *err = 1; return NAN;
    case EX_ADD: FL(1); return eval_expr(e->l, legend, ctx, D, err) + eval_expr(e->r, legend, ctx, D, err);
    case EX_SUB: FL(1); return eval_expr(e->l, legend, ctx, D, err) - eval_expr(e->r, legend, ctx, D, err);
    case EX_MUL: FL(1); return eval_expr(e->l, legend, ctx, D, err) * eval_expr(e->r, legend, ctx, D, err);
    case EX_DIV: FLD(1); return eval_expr(e->l, legend, ctx, D, err) / eval_expr(e->r, legend, ctx, D, err);
    case EX_REM: return eo_fmod_f64(eval_expr(e->l, legend, ctx, D, err), eval_expr(e->r, legend, ctx, D, err));
    case EX_POW: return pow_int(eval_expr(e->l, legend, ctx, D, err), eval_expr(e->r, legend, ctx, D, err));
    case EX_NEG: return -eval_expr(e->l, legend, ctx, D, err);
    default: {
        eo_f64 x = eval_expr(e->l, legend, ctx, D, err);
        eo_f64 y = e->r ? eval_expr(e->r, legend, ctx, D, err) : 0.0;
        switch (e->fn) {
        case FN_SQRT: return eo_sqrt_f64(x);
        case FN_ABS: return fabs(x);
        case FN_FLOOR: return floor(x);
        case FN_CEIL: return ceil(x);
        case FN_MIN: return expr_min(x, y);
        case FN_MAX: return expr_max(x, y);
        case FN_SIN: return eo_sin_f64(x);
        case FN_COS: return eo_cos_f64(x);
        case FN_TAN: return eo_tan_f64(x);
        case FN_ASIN: return eo_asin_f64(x);
        case FN_ACOS: return eo_acos_f64(x);
        case FN_ATAN: return eo_atan_f64(x);
        case FN_ATAN2: return eo_atan2_f64(x, y);
        default: return expr_signum(x);
        }
    }
    }
}

static int check_vars(const expr_node *e, const char *legend, int D) {
    if (!e) return 1;
    if (e->kind == EX_VAR) {
        if (strlen(e->var) != 1) return 0;
        for (int i = 0; i < D && legend[i]; i++) if (legend[i] == e->var[0]) return 1;
        return 0;
    }
    return check_vars(e->l, legend, D) && check_vars(e->r, legend, D);
}

int eo_transformation_expr(eo_scene *s, const char *expression, const char *inverse_expression) {   /* scene.rs:961-989 */
    expr_node *a = parse_expr(expression), *b = parse_expr(inverse_expression);
    if (!a || !b) { free_expr(a); free_expr(b); return fail(s, "Invalid component transformation expression"); }
    int h; obj *o = new_obj(s, K_EXPR, 0, &h);
    o->ex = a; o->ex_inv = b;
    return h;
}
int eo_component_transformation(eo_scene *s, const int *exprs, int n) {   /* scene.rs:991-1008, material.rs:95-97 */
    if (n != s->dim) return fail(s, "The number of functions must be equal to the number of dimensions!");
    int h; obj *o = new_obj(s, K_TRANSFORM, 0, &h);
    o->n_children = n; o->children = calloc((size_t)n, sizeof(obj *));
    for (int i = 0; i < n; i++) { obj *e = get_obj(s, exprs[i], K_EXPR); if (!e) return fail(s, "not an expr"); o->children[i] = e; }
    return h;
}
int eo_material_linear_space(eo_scene *s, const char *legend, const int *transformations, int n) {   /* scene.rs:1011-1033 */
    if ((int)strlen(legend) < s->dim) return fail(s, "The legend is too short!");
    if (strlen(legend) > 15) return fail(s, "legend too long");
    int h; obj *o = new_obj(s, K_MATERIAL, MAT_LINEAR, &h);
    strcpy(o->legend, legend);
    o->n_children = n; o->children = calloc((size_t)(n ? n : 1), sizeof(obj *));
    for (int i = 0; i < n; i++) {
        obj *tr = get_obj(s, transformations[i], K_TRANSFORM);
        if (!tr) return fail(s, "not a transformation");
        for (int k = 0; k < tr->n_children; k++)
            if (!check_vars(tr->children[k]->ex, legend, s->dim) || !check_vars(tr->children[k]->ex_inv, legend, s->dim))
                return fail(s, "Could not evaluate the expression. (unknown variable)");
        o->children[i] = tr;
    }
    return h;
}

/* ComponentTransformation::transform_with (material.rs:91-112): the context is built from the
 * vector BEFORE any component is overwritten. */
static void transform_with(tctx *t, const obj *tr, const char *legend, double *v, int inverse) {
    int D = t->D;
    eo_f64 ctx[MAXD];
    for (int i = 0; i < D; i++) ctx[i] = v[i];
    for (int i = 0; i < D; i++) {
        int err = 0;
        const obj *e = tr->children[i];
        v[i] = (double)eval_expr(inverse ? e->ex_inv : e->ex, legend, ctx, D, &err);
        if (err) t->stats.errors++;
    }
}
static void material_enter(tctx *t, const obj *m, double *dir) {          /* material.rs:135-137,150-154 */
    if (m->sub != MAT_LINEAR) return;
    for (int i = 0; i < m->n_children; i++) transform_with(t, m->children[i], m->legend, dir, 0);
}
static void material_exit(tctx *t, const obj *m, double *dir) {           /* material.rs:140-142,156-162 */
    if (m->sub != MAT_LINEAR) return;
    for (int i = m->n_children - 1; i >= 0; i--) transform_with(t, m->children[i], m->legend, dir, 1);
}

/* ------------------------------------------------------------------ palette 0.2.1 (UNVERIFIED) */

static double clamp01(double v) { if (v < R(0.0)) return R(0.0); if (v > R(1.0)) return R(1.0); return v; }
static int is_normal(double x) { return isnormal(x); }

typedef struct { double r, g, b, a; } pre_t;   /* PreAlpha<Rgb<F>, F> */

static pre_t into_premultiplied(rgba_t c) {
    FL(3);
    double alpha = clamp01(c.a);
    pre_t p = { c.r * alpha, c.g * alpha, c.b * alpha, alpha };
    return p;
}
static rgba_t from_premultiplied(pre_t p) {
    double alpha = clamp01(p.a);
    rgba_t c;
    if (is_normal(alpha)) { FLD(3); c.r = p.r / alpha; c.g = p.g / alpha; c.b = p.b / alpha; }
    else { c.r = R(0.0); c.g = R(0.0); c.b = R(0.0); }
    c.a = alpha;
    return c;
}

static double blend_chan(int fn, double a, double b, double sa, double da) {
    const double one = R(1.0), two = R(2.0);
#ifdef EO_COUNT_FLOPS
    {   /* adds / multiplies of the formula taken (the divisions of dodge / burn / soft_light are counted with them: 1-2 each) */
        static const unsigned char BLEND_FL[BL_COUNT] = { 3, 1, 2, 5, 7, 1, 9, 3, 10, 10, 10, 13, 13, 10, 16, 6, 4, 0 };
        FL(BLEND_FL[fn < BL_COUNT ? fn : 0]);
    }
#endif
    switch (fn) {
    case BL_OVER: return a + b * (one - sa);
    case BL_INSIDE: return a * da;
    case BL_OUTSIDE: return a * (one - da);
    case BL_ATOP: return a * da + b * (one - sa);
    case BL_XOR: return a * (one - da) + b * (one - sa);
    case BL_PLUS: return a + b;
    case BL_MULTIPLY: return a * b + a * (one - da) + b * (one - sa);
    case BL_SCREEN: return a + b - a * b;
    case BL_OVERLAY:
        if (b * two <= da) return two * a * b + a * (one - da) + b * (one - sa);
        return a * (one + da) + b * (one + sa) - two * a * b - sa * da;
    case BL_DARKEN: return rust_min(a * da, b * sa) + a * (one - da) + b * (one - sa);
    case BL_LIGHTEN: return rust_max(a * da, b * sa) + a * (one - da) + b * (one - sa);
    case BL_DODGE:
        if (a == sa && !is_normal(b)) return a * (one - da);
        if (a == sa) return sa * da + a * (one - da) + b * (one - sa);
        return sa * da * rust_min(one, (b / da) * sa / (sa - a)) + a * (one - da) + b * (one - sa);
    case BL_BURN:
        if (!is_normal(a) && b == da) return sa * da + b * (one - sa);
        if (!is_normal(a)) return b * (one - sa);
        return sa * da * (one - rust_min(one, (one - b / da) * sa / a)) + a * (one - da) + b * (one - sa);
    case BL_HARD_LIGHT:
        if (a * two <= sa) return two * a * b + a * (one - da) + b * (one - sa);
        return a * (one + da) + b * (one + sa) - two * a * b - sa * da;
    case BL_SOFT_LIGHT: {
        double m = is_normal(da) ? b / da : R(0.0);
        if (a * two <= sa) return b * (sa + (two * a - sa) * (one - m)) + a * (one - da) + b * (one - sa);
        if (b * R(4.0) <= da) {
            double m2 = m * m, m3 = m2 * m;
            return da * (two * a - sa) * (m3 * R(16.0) - m2 * R(12.0) - m * R(3.0)) + a - a * da + b;
        }
        return da * (two * a - sa) * (sqrt(m) - m) + a - a * da + b;
    }
    case BL_DIFFERENCE: return a + b - two * rust_min(a * da, b * sa);
    default: return a + b - two * a * b;   /* exclusion */
    }
}
static double blend_alpha(int fn, double sa, double da) {
    FL(3);
    switch (fn) {
    case BL_INSIDE: return clamp01(sa * da);
    case BL_OUTSIDE: return clamp01(sa * (R(1.0) - da));
    case BL_ATOP: return clamp01(da);
    case BL_XOR: return clamp01(sa + da - R(2.0) * sa * da);
    case BL_PLUS: return clamp01(sa + da);
    default: return clamp01(sa + da - sa * da);
    }
}
static pre_t blend_pre(int fn, pre_t s, pre_t d) {
    pre_t o;
    o.r = blend_chan(fn, s.r, d.r, s.a, d.a);
    o.g = blend_chan(fn, s.g, d.g, s.a, d.a);
    o.b = blend_chan(fn, s.b, d.b, s.a, d.a);
    o.a = blend_alpha(fn, s.a, d.a);
    return o;
}
/* surface.rs:315-322 */
static rgba_t blend_rgba(int fn, rgba_t s, rgba_t d) { return from_premultiplied(blend_pre(fn, into_premultiplied(s), into_premultiplied(d))); }

/* util.rs:265-285 */
static rgba_t combine_palette_color(rgba_t a, rgba_t b, double a_ratio) {
    if (a_ratio <= R(0.0)) return b;
    if (a_ratio >= R(1.0)) return a;
    rgba_t o;
    FL(16);
    o.r = a.r * a_ratio + b.r * (R(1.0) - a_ratio);
    o.g = a.g * a_ratio + b.g * (R(1.0) - a_ratio);
    o.b = a.b * a_ratio + b.b * (R(1.0) - a_ratio);
    o.a = a.a * a_ratio + b.a * (R(1.0) - a_ratio);
    return o;
}

/* RgbPixel for [u8;N]: clamp(c,0,1)*255 cast with truncation (UNVERIFIED).  A NaN would panic
 * in NumCast; the oracle writes 0 and counts it. */
static uint8_t to_u8(tctx *t, double c) {
    FL(1);
    double v = clamp01(c) * R(255.0);
    if (eo_isnan(v)) { if (t) t->stats.nan_pixels++; return 0; }
    return (uint8_t)v;
}
static void to_pixel4(tctx *t, rgba_t c, uint8_t *px) { px[0] = to_u8(t, c.r); px[1] = to_u8(t, c.g); px[2] = to_u8(t, c.b); px[3] = to_u8(t, c.a); }
static rgba_t new_u8(const uint8_t *px) { FLD(4); rgba_t c = { (double)px[0] / R(255.0), (double)px[1] / R(255.0), (double)px[2] / R(255.0), (double)px[3] / R(255.0) }; return c; }

/* Hsv -> Rgb, RgbHue::to_positive_degrees (palette 0.2.1, UNVERIFIED) */
static void hsv_to_rgb(double hue, double saturation, double value, double *r, double *g, double *b) {
    double deg = hue;
    if (fabs(deg) < R(1.0e9)) {       /* guard for the while loops; larger hues fall through like NaN */
        while (deg >= R(360.0)) deg = deg - R(360.0);
        while (deg < R(0.0)) deg = deg + R(360.0);
    }
    FL(7); FLD(1);
    double c = value * saturation;
    double h = deg / R(60.0);
    double x = c * (R(1.0) - fabs(fmod(h, R(2.0)) - R(1.0)));
    double m = value - c;
    double red, green, blue;
    if (h >= R(0.0) && h < R(1.0)) { red = c; green = x; blue = R(0.0); }
    else if (h >= R(1.0) && h < R(2.0)) { red = x; green = c; blue = R(0.0); }
    else if (h >= R(2.0) && h < R(3.0)) { red = R(0.0); green = c; blue = x; }
    else if (h >= R(3.0) && h < R(4.0)) { red = R(0.0); green = x; blue = c; }
    else if (h >= R(4.0) && h < R(5.0)) { red = x; green = R(0.0); blue = c; }
    else { red = c; green = R(0.0); blue = x; }
    *r = red + m; *g = green + m; *b = blue + m;
}
void eo_rgba_from_hsva(double hue, double s, double v, double a, double *out) {   /* scene.rs:663-667 */
    hsv_to_rgb(hue, s, v, &out[0], &out[1], &out[2]);
    out[3] = a;
}

/* ------------------------------------------------------------------ own 4-D gradient noise
 * The reference uses noise 0.4.1's Perlin with a rand::random() seed and wall-clock time
 * (d3/entity/surface.rs:22-58): not reproducible by design.  Both the oracle and the product
 * implement this documented substitute: classic improved Perlin noise in 4-D, permutation
 * = Fisher-Yates driven by xorshift32(seed + 0x9E3779B9). */
static void perlin_build_perm(uint32_t seed, uint8_t *perm512) {
    uint8_t p[256];
    for (int i = 0; i < 256; i++) p[i] = (uint8_t)i;
    uint32_t st = seed + 0x9E3779B9u;
    if (st == 0) st = 1;
    for (int i = 255; i >= 1; i--) {
        st ^= st << 13; st ^= st >> 17; st ^= st << 5;
        uint32_t j = st % (uint32_t)(i + 1);
        uint8_t tmp = p[i]; p[i] = p[j]; p[j] = tmp;
    }
    for (int i = 0; i < 512; i++) perm512[i] = p[i & 255];
}
static double pfade(double t) { FL(7); return t * t * t * (t * (t * R(6.0) - R(15.0)) + R(10.0)); }
static double plerp(double t, double a, double b) { FL(3); return a + t * (b - a); }
static double pgrad4(int hash, double x, double y, double z, double w) {
    FL(2);
    int h = hash & 31;
    double a = (h < 24) ? x : y;
    double b = (h < 16) ? y : z;
    double c = (h < 8) ? z : w;
    return ((h & 1) ? -a : a) + ((h & 2) ? -b : b) + ((h & 4) ? -c : c);
}
static int pcell(double f) { double m = fmod(f, R(256.0)); return (m == m) ? (((int)m) & 255) : 0; }
static double perlin4(const uint8_t *perm, double x, double y, double z, double w) {
    double fx = floor(x), fy = floor(y), fz = floor(z), fw = floor(w);
    int xi = pcell(fx), yi = pcell(fy), zi = pcell(fz), wi = pcell(fw);
    FL(4 + 16 * 4 + 1);      /* fractions, the 16 corners' offsets, the final scale */
    double xf = x - fx, yf = y - fy, zf = z - fz, wf = w - fw;
    double u = pfade(xf), v = pfade(yf), s = pfade(zf), q = pfade(wf);
    double n[16];
    for (int c = 0; c < 16; c++) {
        int dx = c & 1, dy = (c >> 1) & 1, dz = (c >> 2) & 1, dw = (c >> 3) & 1;
        int hsh = perm[perm[perm[perm[xi + dx] + yi + dy] + zi + dz] + wi + dw];
        n[c] = pgrad4(hsh, xf - (double)dx, yf - (double)dy, zf - (double)dz, wf - (double)dw);
    }
    double lx[8], ly[4], lz[2];
    for (int c = 0; c < 8; c++) lx[c] = plerp(u, n[2 * c], n[2 * c + 1]);
    for (int c = 0; c < 4; c++) ly[c] = plerp(v, lx[2 * c], lx[2 * c + 1]);
    for (int c = 0; c < 2; c++) lz[c] = plerp(s, ly[2 * c], ly[2 * c + 1]);
    return R(0.87) * plerp(q, lz[0], lz[1]);
}

/* ------------------------------------------------------------------ surface providers */

static int simple_obj(eo_scene *s, int kind, int sub, obj **out) { int h; *out = new_obj(s, kind, sub, &h); return h; }

int eo_reflection_ratio_uniform(eo_scene *s, double ratio) { obj *o; int h = simple_obj(s, K_RATIO, RATIO_UNIFORM, &o); o->p0 = ratio; return h; }
int eo_reflection_ratio_fresnel(eo_scene *s, double inside, double outside) { obj *o; int h = simple_obj(s, K_RATIO, RATIO_FRESNEL, &o); o->p0 = inside; o->p1 = outside; return h; }
int eo_reflection_direction_specular(eo_scene *s) { obj *o; return simple_obj(s, K_REFLDIR, 0, &o); }
int eo_threshold_direction_identity(eo_scene *s) { obj *o; return simple_obj(s, K_THRDIR, THR_IDENTITY, &o); }
int eo_threshold_direction_snell(eo_scene *s, double n) { obj *o; int h = simple_obj(s, K_THRDIR, THR_SNELL, &o); o->p0 = n; return h; }
int eo_blend_function(eo_scene *s, const char *name, double ratio) {
    for (int i = 0; i < BL_COUNT; i++) if (!strcmp(name, BLEND_NAMES[i])) { obj *o; int h = simple_obj(s, K_BLEND, i, &o); o->p0 = ratio; return h; }
    return fail(s, "unknown blend function");
}
static rgba_t rgba_of(const double *c) { rgba_t r = { c[0], c[1], c[2], c[3] }; return r; }
int eo_color_uniform(eo_scene *s, const double *rgba) { obj *o; int h = simple_obj(s, K_COLOR, COL_UNIFORM, &o); o->c0 = rgba_of(rgba); return h; }
int eo_color_blend(eo_scene *s, int source, int destination, int blend_function) {
    obj *a = get_obj(s, source, K_COLOR), *b = get_obj(s, destination, K_COLOR), *f = get_obj(s, blend_function, K_BLEND);
    if (!a || !b || !f) return fail(s, "surface_color_blend: wrong argument types");
    obj *o; int h = simple_obj(s, K_COLOR, COL_BLEND, &o); o->o0 = a; o->o1 = b; o->o2 = f; return h;
}
int eo_color_illumination_global(eo_scene *s, const double *light, const double *dark) {
    obj *o; int h = simple_obj(s, K_COLOR, COL_ILLUM_GLOBAL, &o); o->c0 = rgba_of(light); o->c1 = rgba_of(dark); return h;
}
int eo_color_illumination_directional(eo_scene *s, const double *direction, const double *light, const double *dark) {
    obj *o; int h = simple_obj(s, K_COLOR, COL_ILLUM_DIR, &o); v_copy(s->dim, direction, o->dir); o->c0 = rgba_of(light); o->c1 = rgba_of(dark); return h;
}
int eo_color_perlin_hue(eo_scene *s, uint32_t seed, double size, double speed) {
    if (s->dim != 3) return fail(s, "perlin hue is 3-D only");
    obj *o; int h = simple_obj(s, K_COLOR, COL_PERLIN, &o); o->p0 = size; o->p1 = speed; perlin_build_perm(seed, o->perm); return h;
}
int eo_color_texture(eo_scene *s, int mapped) {
    obj *m = get_obj(s, mapped, K_MAPPED); if (!m) return fail(s, "not a mapped texture");
    obj *o; int h = simple_obj(s, K_COLOR, COL_TEXTURE, &o); o->o0 = m; return h;
}
int eo_uv_sphere(eo_scene *s, const double *center3) { obj *o; int h = simple_obj(s, K_UV, UV_SPHERE3, &o); v_copy(3, center3, o->a); return h; }
int eo_uv_derank(eo_scene *s, int uvfn) {
    obj *u = get_obj(s, uvfn, K_UV); if (!u) return fail(s, "not a uv fn");
    obj *o; int h = simple_obj(s, K_UV, UV_DERANK, &o); o->o0 = u; return h;
}
int eo_texture_image(eo_scene *s, int kind, uint32_t w, uint32_t h_, const uint8_t *rgba8) {
    if (w == 0 || h_ == 0) return fail(s, "empty texture");
    obj *o; int h = simple_obj(s, K_TEXTURE, kind, &o);
    o->w = w; o->h = h_; o->pixels = malloc((size_t)w * h_ * 4); memcpy(o->pixels, rgba8, (size_t)w * h_ * 4);
    return h;
}
int eo_mapped_texture(eo_scene *s, int uvfn, int texture) {
    obj *u = get_obj(s, uvfn, K_UV), *tx = get_obj(s, texture, K_TEXTURE);
    if (!u || !tx) return fail(s, "MappedTextureImpl: wrong argument types");
    obj *o; int h = simple_obj(s, K_MAPPED, 0, &o); o->o0 = u; o->o1 = tx; return h;
}
int eo_surface_composable(eo_scene *s, int ratio, int refl, int thr, int color) {
    obj *a = get_obj(s, ratio, K_RATIO), *b = get_obj(s, refl, K_REFLDIR), *c = get_obj(s, thr, K_THRDIR), *d = get_obj(s, color, K_COLOR);
    if (!a || !b || !c || !d) return fail(s, "ComposableSurface: wrong argument types");
    obj *o; int h = simple_obj(s, K_SURFACE, 0, &o); o->o0 = a; o->o1 = b; o->o2 = c; o->o3 = d; return h;
}
int eo_entity(eo_scene *s, int shape, int material, int surface) {
    obj *sh = get_obj(s, shape, K_SHAPE), *m = get_obj(s, material, K_MATERIAL);
    obj *sf = surface >= 0 ? get_obj(s, surface, K_SURFACE) : NULL;
    if (!sh || !m || (surface >= 0 && !sf)) return fail(s, "Entity: wrong argument types");
    obj *o; int h = simple_obj(s, K_ENTITY, 0, &o); o->o0 = sh; o->o1 = m; o->o2 = sf; return h;
}
int eo_entity_void(eo_scene *s, int material) {   /* universe/entity/mod.rs:64-107 */
    int sh = eo_shape_void(s);
    return eo_entity(s, sh, material, -1);
}

int eo_default_camera(int dim, const double *loc, eo_camera *c) {   /* d3/entity/camera.rs:42-52, d4/entity/camera.rs:47-58 */
    memset(c, 0, sizeof *c);
    c->dim = dim;
    if (loc) for (int i = 0; i < dim; i++) c->location[i] = loc[i];
    c->forward[0] = R(1.0);
    c->up[2 < dim ? 2 : dim - 1] = R(1.0);
    if (dim == 2) { c->up[0] = R(0.0); c->up[1] = R(1.0); }
    c->left[1] = R(1.0);
    c->fov_deg = 90;
    c->max_depth = 10;
    return 0;
}
int eo_universe(eo_scene *s, const eo_camera *camera, const int *entities, int n, int background) {   /* scene.rs:1339-1351 */
    obj *bg = get_obj(s, background, K_MAPPED);
    if (!bg) return fail(s, "Universe: background must be a MappedTexture");
    free(s->entities);
    s->entities = calloc((size_t)(n ? n : 1), sizeof(obj *));
    for (int i = 0; i < n; i++) { obj *e = get_obj(s, entities[i], K_ENTITY); if (!e) return fail(s, "Universe: not an entity"); s->entities[i] = e; }
    s->n_entities = n;
    s->background = bg;
    s->camera = *camera;
    return 0;
}
int eo_scene_camera(const eo_scene *s, eo_camera *out) { *out = s->camera; return 0; }

/* ------------------------------------------------------------------ textures */

static void uv_eval(const obj *uv, const double *point, double *u, double *v) {
    if (uv->sub == UV_DERANK) { uv_eval(uv->o0, point, u, v); return; }    /* d4/entity/surface.rs:11-15 */
    double p[3], pn[3];                                                     /* d3/entity/surface.rs:60-68 */
    v_sub(3, point, uv->a, p);
    v_normalize(3, p, pn);
    FL(3); FLD(2);
    *u = R(0.5) + eo_atan2(pn[1], pn[0]) / (R(2.0) * EO_PI_C);
    *v = R(0.5) - eo_asin(pn[2]) / EO_PI_C;
}

/* NumCast::from(f64) -> u32: None (panic) if NaN or outside the u32 range, else truncation */
static int cast_u32(tctx *t, double x, uint32_t *out) {
    if (!(x > -R(1.0) && x < R(4294967296.0))) { t->stats.errors++; *out = 0; return 0; }
    *out = (uint32_t)x;
    return 1;
}

static rgba_t texture_eval(tctx *t, const obj *tx, double pu, double pv) {
    uint32_t W = tx->w, H = tx->h;
    if (tx->sub == EO_TEX_NEAREST) {                                        /* surface.rs:434-451 */
        FL(2); FLD(4);
        double x = floor(pu * (double)W), y = floor(pv * (double)H);
        uint32_t xi, yi;
        cast_u32(t, x, &xi); cast_u32(t, y, &yi);
        xi = (uint32_t)remainder_i((int64_t)xi, (int64_t)W);
        yi = (uint32_t)remainder_i((int64_t)yi, (int64_t)H);
        const uint8_t *px = tx->pixels + ((size_t)yi * W + xi) * 4;
        rgba_t c = { (double)px[0] / R(255.0), (double)px[1] / R(255.0), (double)px[2] / R(255.0), (double)px[3] / R(255.0) };
        return c;
    }
    FL(6 + 4 + 2 + 4 * 9); FLD(4);            /* coordinates, offsets, the two weights, 4 channels x 9 */
    double x = pu * (double)W - R(0.5), y = pv * (double)H - R(0.5);             /* surface.rs:453-489 */
    double ox = x - floor(x), oy = y - floor(y);
    static const double OFF[4][2] = { {0, 0}, {1, 0}, {0, 1}, {1, 1} };
    const uint8_t *px[4];
    for (int i = 0; i < 4; i++) {
        uint32_t xi, yi;
        cast_u32(t, remainder_f(x + OFF[i][0], (double)W), &xi);
        cast_u32(t, remainder_f(y + OFF[i][1], (double)H), &yi);
        if (xi >= W) { t->stats.errors++; xi = W - 1; }                    /* image::get_pixel would panic */
        if (yi >= H) { t->stats.errors++; yi = H - 1; }
        px[i] = tx->pixels + ((size_t)yi * W + xi) * 4;
    }
    double data[4];
    for (int k = 0; k < 4; k++)
        data[k] = (((double)px[0][k] * (R(1.0) - ox) + (double)px[1][k] * ox) * (R(1.0) - oy) +
                   ((double)px[2][k] * (R(1.0) - ox) + (double)px[3][k] * ox) * oy) / R(255.0);
    rgba_t c = { data[0], data[1], data[2], data[3] };
    return c;
}
static rgba_t mapped_get_color(tctx *t, const obj *m, const double *point) {   /* surface.rs:528-534 */
    double u, v;
    uv_eval(m->o0, point, &u, &v);
    return texture_eval(t, m->o1, u, v);
}

/* ------------------------------------------------------------------ tracing */

typedef struct {           /* TracingContext, shape.rs:111-125 */
    const obj *origin_traceable;
    double origin_location[MAXD], origin_direction[MAXD];
    const obj *intersection_traceable;
    hit_t intersection;
    double normal_closer[MAXD];
    int exiting;
} trace_ctx;

/* util.rs:631-666, for one vector */
static void general_rotation(int D, const double *self, const double *other, double angle, double *vec) {
    double orig[MAXD][MAXD], res[MAXD][MAXD];     /* [row][col] */
    for (int r = 0; r < D; r++) for (int c = 0; c < D; c++) orig[r][c] = (r == c) ? R(1.0) : R(0.0);
    for (int r = 0; r < D; r++) { orig[r][0] = self[r]; orig[r][1] = other[r]; }
    memcpy(res, orig, sizeof res);
    for (int i = 1; i < D; i++) {
        for (int j = 0; j < i; j++) {
            double oc[MAXD], rj[MAXD];
            for (int r = 0; r < D; r++) { oc[r] = orig[r][i]; rj[r] = res[r][j]; }
            double d = v_dot(D, rj, oc);
            FL(2 * D);
            for (int r = 0; r < D; r++) orig[r][i] = oc[r] - rj[r] * d;
        }
        double col[MAXD], nc[MAXD];
        for (int r = 0; r < D; r++) col[r] = orig[r][i];
        v_normalize(D, col, nc);
        for (int r = 0; r < D; r++) res[r][i] = nc[r];
    }
    double rot[MAXD][MAXD];
    for (int r = 0; r < D; r++) for (int c = 0; c < D; c++) rot[r][c] = (r == c) ? R(1.0) : R(0.0);
    double ca = eo_cos(angle), sa = eo_sin(angle);
    rot[0][0] = ca; rot[0][1] = -sa; rot[1][0] = sa; rot[1][1] = ca;
    double tmp[MAXD][MAXD], fin[MAXD][MAXD];
    FL(2 * (2 * D * D * D) + 2 * D * D);      /* two matrix products and the matrix-vector product */
    for (int r = 0; r < D; r++) for (int c = 0; c < D; c++) {           /* rotation * result^T */
        double acc = R(0.0);
        for (int k = 0; k < D; k++) acc = acc + rot[r][k] * res[c][k];
        tmp[r][c] = acc;
    }
    for (int r = 0; r < D; r++) for (int c = 0; c < D; c++) {           /* result * (...) */
        double acc = R(0.0);
        for (int k = 0; k < D; k++) acc = acc + res[r][k] * tmp[k][c];
        fin[r][c] = acc;
    }
    double out[MAXD];
    for (int r = 0; r < D; r++) {
        double acc = R(0.0);
        for (int k = 0; k < D; k++) acc = acc + fin[r][k] * vec[k];
        out[r] = acc;
    }
    v_copy(D, out, vec);
}

static double reflection_ratio(tctx *t, const obj *p, const trace_ctx *c) {
    int D = t->D;
    if (p->sub == RATIO_UNIFORM) return c->exiting ? R(0.0) : p->p0;          /* surface.rs:200-211 */
    double normal[MAXD];                                                    /* surface.rs:213-244 */
    v_neg(D, c->normal_closer, normal);
    double from_theta = angle_between(D, c->intersection.direction, normal);
    double from_index, to_index;
    if (c->exiting) { from_index = p->p0; to_index = p->p1; } else { from_index = p->p1; to_index = p->p0; }
    FL(1); FLD(1);
    double to_theta = eo_asin((from_index / to_index) * eo_sin(from_theta));
    if (eo_isnan(to_theta)) return R(1.0);
    FL(4 + 4 + 2 + 1); FLD(3);
    double product_1_s = from_index * eo_cos(from_theta);
    double product_2_s = to_index * eo_cos(to_theta);
    double product_1_p = from_index * eo_cos(to_theta);
    double product_2_p = to_index * eo_cos(from_theta);
    double rs = (product_1_s - product_2_s) / (product_1_s + product_2_s); rs = rs * rs;
    double rp = (product_1_p - product_2_p) / (product_1_p + product_2_p); rp = rp * rp;
    return (rs + rp) / (R(1.0) + R(1.0));
}

static void reflection_direction(tctx *t, const trace_ctx *c, double *out) {   /* surface.rs:246-256 */
    int D = t->D;
    double d = v_dot(D, c->intersection.direction, c->normal_closer);
    FL(3 * D);
    for (int i = 0; i < D; i++) out[i] = c->normal_closer[i] * -R(2.0) * d + c->intersection.direction[i];
}

static void threshold_direction(tctx *t, const obj *p, const trace_ctx *c, double *out) {
    int D = t->D;
    v_copy(D, c->intersection.direction, out);
    if (p->sub == THR_IDENTITY) return;                                     /* surface.rs:258-266 */
    double normal[MAXD];                                                    /* surface.rs:268-288 */
    v_neg(D, c->normal_closer, normal);
    double from_theta = angle_between(D, c->intersection.direction, normal);
    if (!c->exiting) FLD(1);
    FL(2);
    double modifier = c->exiting ? p->p0 : R(1.0) / p->p0;
    double to_theta = eo_asin(modifier * eo_sin(from_theta));
    double angle_delta = to_theta - from_theta;
    general_rotation(D, normal, c->intersection.direction, angle_delta, out);
}

static rgba_t surface_color(tctx *t, const obj *p, const trace_ctx *c) {
    int D = t->D;
    switch (p->sub) {
    case COL_UNIFORM: return p->c0;                                         /* surface.rs:424-429 */
    case COL_BLEND: {                                                       /* surface.rs:295-322 */
        rgba_t src = surface_color(t, p->o0, c);
        rgba_t dst = surface_color(t, p->o1, c);
        if (p->o2->sub == BL_RATIO) return combine_palette_color(src, dst, p->o2->p0);
        return blend_rgba(p->o2->sub, src, dst);
    }
    case COL_ILLUM_GLOBAL: {                                                /* surface.rs:410-422 */
        double original_angle = angle_between(D, c->normal_closer, c->intersection.direction);
        FL(1); FLD(1);
        double angle = EO_PI_C - original_angle;
        double ratio = angle / EO_FRAC_PI_2_C;
        return combine_palette_color(p->c1, p->c0, ratio);
    }
    case COL_ILLUM_DIR: {                                                   /* surface.rs:392-408 */
        double normal[MAXD], nl[MAXD];
        v_copy(D, c->intersection.normal, normal);
        if (angle_between(D, c->intersection.direction, normal) > EO_FRAC_PI_2_C) v_neg(D, normal, normal);
        v_neg(D, p->dir, nl);
        double angle = angle_between(D, normal, nl);
        FL(1); FLD(1);
        double ratio = R(1.0) - angle / EO_PI_C;
        return combine_palette_color(p->c1, p->c0, ratio);
    }
    case COL_PERLIN: {                                                      /* d3/entity/surface.rs:22-40 */
        double time_millis = (double)t->time_ms / R(1000.0);
        const double *l = c->intersection.location;
        FL(2); FLD(4);
        double value = perlin4(p->perm, l[0] / p->p0, l[1] / p->p0, l[2] / p->p0, time_millis * p->p1);
        rgba_t o;
        hsv_to_rgb(value * R(360.0), R(1.0), R(1.0), &o.r, &o.g, &o.b);
        o.a = R(1.0);
        return o;
    }
    default: return mapped_get_color(t, p->o0, c->intersection.location);   /* surface.rs:536-542 */
    }
}

static const obj *material_at(tctx *t, const double *loc) {                 /* universe/mod.rs:229-251 */
    const eo_scene *s = t->scene;
    for (int i = 0; i < s->n_entities; i++)
        if (shape_inside(t->D, s->entities[i]->o0, loc)) return s->entities[i];
    return NULL;
}

static rgba_t trace(tctx *t, uint32_t max_depth, const obj *belongs_to, const double *loc, const double *dir, double *first_hit_t);

static int trace_closest(tctx *t, const obj *belongs_to, const double *loc, const double *dir, trace_ctx *out) {   /* universe/mod.rs:85-147 */
    const eo_scene *s = t->scene;
    int D = t->D;
    int have = 0;
    double closest_distance = R(0.0);
    for (int e = 0; e < s->n_entities; e++) {
        const obj *other = s->entities[e];
        if (!other->o2) continue;                          /* filter: surface().is_some() (universe/mod.rs:158-160) */
        size_t mark = t->arena_used;
        provider *p = universe_intersect(t, loc, dir, other->o0);
        hit_t first;
        if (provider_get(t, p, 0, &first)) {
            int exiting; double closer[MAXD];
            if (angle_between(D, first.direction, first.normal) < EO_FRAC_PI_2_C) { v_neg(D, first.normal, closer); exiting = 1; }
            else { v_copy(D, first.normal, closer); exiting = 0; }
            if (!have || closest_distance > first.distance) {
                out->origin_traceable = belongs_to;
                v_copy(D, loc, out->origin_location);
                v_copy(D, dir, out->origin_direction);
                out->intersection_traceable = other;
                out->intersection = first;
                v_copy(D, closer, out->normal_closer);
                out->exiting = exiting;
                have = 1;
                closest_distance = first.distance;
            }
        }
        t->arena_used = mark;
    }
    return have;
}

/* ComposableSurface::get_color (surface.rs:62-162) */
static rgba_t surface_get_color(tctx *t, const obj *surface, const trace_ctx *c, uint32_t depth_remaining) {
    int D = t->D;
    double ratio = reflection_ratio(t, surface->o0, c);
    ratio = rust_max(rust_min(ratio, R(1.0)), R(0.0));

    int have_inter = 0, have_refl = 0;
    rgba_t inter = {0, 0, 0, 0}, refl = {0, 0, 0, 0};

    if (!(ratio >= R(1.0))) {                                                  /* get_intersection_color */
        rgba_t sc = surface_color(t, surface->o3, c);
        uint8_t px[4];
        to_pixel4(t, sc, px);
        if (px[3] == 255) { inter = sc; have_inter = 1; }
        else {
            double tdir[MAXD], new_origin[MAXD];
            threshold_direction(t, surface->o2, c, tdir);
            FL(3 * D);
            for (int i = 0; i < D; i++) new_origin[i] = c->intersection.location[i] + -c->normal_closer[i] * EO_EPS128_A * R(128.0);
            const obj *dest = c->exiting ? material_at(t, new_origin) : c->intersection_traceable;
            if (dest) {
                material_exit(t, c->origin_traceable->o1, tdir);
                material_enter(t, dest->o1, tdir);
                rgba_t tc = trace(t, depth_remaining - 1, dest, new_origin, tdir, NULL);
                uint8_t tpx[4];
                to_pixel4(t, tc, tpx);
                inter = blend_rgba(BL_OVER, new_u8(px), new_u8(tpx));
                have_inter = 1;
            }
        }
    }
    if (!(ratio <= R(0.0))) {                                                  /* get_reflection_color */
        double rdir[MAXD], new_origin[MAXD];
        reflection_direction(t, c, rdir);
        FL(3 * D);
        for (int i = 0; i < D; i++) new_origin[i] = c->intersection.location[i] + c->normal_closer[i] * EO_EPS128_A * R(128.0);
        refl = trace(t, depth_remaining - 1, c->origin_traceable, new_origin, rdir, NULL);
        have_refl = 1;
    }
    if (!have_inter) {
        if (!have_refl) { t->stats.errors++; rgba_t z = {0, 0, 0, 0}; return z; }   /* reference: expect() panic */
        return refl;
    }
    if (!have_refl) return inter;
    return combine_palette_color(refl, inter, ratio);
}

static rgba_t trace(tctx *t, uint32_t max_depth, const obj *belongs_to, const double *loc, const double *dir, double *first_hit_t) {   /* universe/mod.rs:149-184 */
    int D = t->D;
    if (first_hit_t) *first_hit_t = -R(1.0);
    if (max_depth > 0) {
        t->stats.rays++;
        trace_ctx c;
        if (trace_closest(t, belongs_to, loc, dir, &c)) {
            if (first_hit_t) *first_hit_t = c.intersection.distance;
            return surface_get_color(t, c.intersection_traceable->o2, &c, max_depth);
        }
    }
    t->stats.bg_samples++;
    double pt[MAXD];
    FL(D);
    for (int i = 0; i < D; i++) pt[i] = R(0.0) + dir[i];                       /* direction.to_point(), util.rs:616-618 */
    return mapped_get_color(t, t->scene->background, pt);
}

/* Universe::trace_path (universe/mod.rs:186-227) with ComposableSurface::get_path (surface.rs:164-197) and
 * Material::trace_path (material.rs:54-56,144-146) inlined where the reference calls them.  The reference recurses
 * (every level returns its callee's result unchanged); so does this.  A step cap stands in for the stack overflow an
 * endless chain of grazing hits would end in: beyond EO_PATH_MAX_STEPS levels the call fails (-1). */
#define EO_PATH_MAX_STEPS 4096
static int trace_path(tctx *t, double distance, const obj *belongs_to, const double *loc, const double *dir,
                      double *out_loc, double *out_dir, int level) {
    int D = t->D;
    if (level > EO_PATH_MAX_STEPS) { t->stats.errors++; return -1; }
    trace_ctx c;
    if (trace_closest(t, belongs_to, loc, dir, &c)) {                         /* filter: surface().is_some(), mod.rs:194-196 */
        if (!(distance - c.intersection.distance <= R(0.0))) {                   /* surface.rs:165-167 */
            double new_distance = distance - c.intersection.distance;
            double new_origin[MAXD], tdir[MAXD];
            for (int i = 0; i < D; i++) new_origin[i] = c.intersection.location[i] + -c.normal_closer[i] * EO_EPS128_A * R(128.0);
            const obj *dest = c.exiting ? material_at(t, new_origin) : c.intersection_traceable;   /* surface.rs:177-185 */
            if (dest) {
                v_copy(D, c.intersection.direction, tdir);
                material_exit(t, c.origin_traceable->o1, tdir);               /* surface.rs:188-189 */
                material_enter(t, dest->o1, tdir);
                return trace_path(t, new_distance, dest, new_origin, tdir, out_loc, out_dir, level + 1);
            }
        }
    }
    for (int i = 0; i < D; i++) { out_loc[i] = loc[i] + dir[i] * distance; out_dir[i] = dir[i]; }   /* material.rs:54-56 */
    material_exit(t, belongs_to->o1, out_dir);                                /* mod.rs:224 */
    return 1;
}

/* Universe::trace_path_unknown (universe/mod.rs:273-286): 1 = Some((location, direction)), 0 = None (no material at
 * `location`), -1 = step cap. */
int eo_trace_path_unknown(const eo_scene *s, const double *loc, const double *dir, double distance, double *out_loc, double *out_dir) {
    tctx t;
    memset(&t, 0, sizeof t);
    t.scene = s; t.D = s->dim; t.arena_cap = EO_ARENA_BYTES; t.arena = malloc(t.arena_cap);
    int rc = 0;
    const obj *belongs_to = material_at(&t, loc);
    if (belongs_to) {
        double d[MAXD];
        v_copy(t.D, dir, d);
        material_enter(&t, belongs_to->o1, d);
        rc = trace_path(&t, distance, belongs_to, loc, d, out_loc, out_dir, 0);
    }
    free(t.arena);
    return rc;
}

/* ------------------------------------------------------------------ Camera::update (camera motion, "next" row f3)
 * d3/entity/camera.rs:94-145,191-245 (PitchYawCamera3), :299-346,396-451 (FreeCamera3); d4/entity/camera.rs:68-136,182-241
 * (FreeCamera4); util.rs:301-322.  Third-party pieces restated from their published behaviour (parity unpinned): nalgebra
 * 0.8.2 UnitQuaternion::new / rotate, cross, Matrix4 (column-major storage), approx_eq_ulps with approx_ulps = 8; det 0.1.0
 * det_copy! as a first-row Laplace expansion. */
typedef struct { double w, i, j, k; } quat_t;
static quat_t quat_from_axisangle(const double *aa) {
    double sqang = v_nsq(3, aa);
    quat_t q = { R(1.0), R(0.0), R(0.0), R(0.0) };
    if (sqang == R(0.0)) return q;
    double ang = sqrt(sqang);
    double s = eo_sin(ang / R(2.0)), c = eo_cos(ang / R(2.0));
    double s_ang = s / ang;
    q.w = c; q.i = aa[0] * s_ang; q.j = aa[1] * s_ang; q.k = aa[2] * s_ang;
    return q;
}
static void quat_rotate(quat_t q, const double *v, double *out) {
    double qv[3] = { q.i, q.j, q.k }, t[3], u[3];
    v_cross3(qv, v, t);
    t[0] = t[0] * R(2.0); t[1] = t[1] * R(2.0); t[2] = t[2] * R(2.0);
    v_cross3(qv, t, u);
    for (int a = 0; a < 3; a++) out[a] = (t[a] * q.w + u[a]) + v[a];
}
static void rotate_axis_angle(const double *axis, double angle, double *v) {
    double aa[3], r[3];
    v_scale(3, axis, angle, aa);
    quat_rotate(quat_from_axisangle(aa), v, r);
    v_copy(3, r, v);
}
static void normalize_mut(int D, double *v) { double r[MAXD]; v_normalize(D, v, r); v_copy(D, r, v); }
static int approx_eq_ulps(double a, double b, uint32_t ulps) {
    if (a == b) return 1;
    if (eo_isnan(a) || eo_isnan(b) || signbit(a) != signbit(b)) return 0;
    int64_t ia, ib;
    memcpy(&ia, &a, 8); memcpy(&ib, &b, 8);
    int64_t d = ia - ib;
    if (d < 0) d = -d;
    return d < (int64_t)ulps;
}
static void rotate_pitch_static(double *forward, double *up, double angle, int snap) {   /* d3/entity/camera.rs:116-137 */
    static const double Z[3] = { R(0.0), R(0.0), R(1.0) };
    double axis_h[3];
    v_cross3(forward, up, axis_h); normalize_mut(3, axis_h);
    if (snap) {
        double result_angle = angle_between(3, forward, Z);
        int to_pole = 0;
        if (result_angle < angle) to_pole = 1;
        else if (EO_PI_C - result_angle < -angle) to_pole = -1;
        if (to_pole) {
            forward[0] = to_pole > 0 ? R(0.0) : -R(0.0); forward[1] = forward[0]; forward[2] = to_pole > 0 ? R(1.0) : -R(1.0);
            v_cross3(axis_h, forward, up); normalize_mut(3, up);
            return;
        }
    }
    rotate_axis_angle(axis_h, angle, forward); normalize_mut(3, forward);
    v_cross3(axis_h, forward, up); normalize_mut(3, up);
}
static double det3x3(double a, double b, double c, double d, double e, double f, double g, double h, double i) {
    return (a * (e * i - f * h) - b * (d * i - f * g)) + c * (d * h - e * g);
}
static void find_orthonormal_4(const double *a, const double *b, const double *c, double *out) {   /* util.rs:301-308 */
    double r[4];
    r[0] = det3x3(a[1], a[2], a[3], b[1], b[2], b[3], c[1], c[2], c[3]);
    r[1] = -det3x3(a[0], a[2], a[3], b[0], b[2], b[3], c[0], c[2], c[3]);
    r[2] = det3x3(a[0], a[1], a[3], b[0], b[1], b[3], c[0], c[1], c[3]);
    r[3] = -det3x3(a[0], a[1], a[2], b[0], b[1], b[2], c[0], c[1], c[2]);
    v_copy(4, r, out);
}
static void mat4_mul_vec(double m[4][4], const double *v, double *out) {   /* m[row][col] */
    double r[4];
    for (int a = 0; a < 4; a++) r[a] = v_dot(4, m[a], v);
    v_copy(4, r, out);
}

/* returns 0 ok, 1 = the reference reaches unimplemented!() (d4/entity/camera.rs:234), -1 = trace_path step cap */
int eo_camera_update(const eo_scene *s, int kind, eo_camera *camera, const eo_input *in) {
    eo_camera cam = *camera;
    int D = cam.dim;
    double sens = in->mouse_sensitivity != R(0.0) ? in->mouse_sensitivity : R(0.01);
    double speed = in->speed != R(0.0) ? in->speed : R(10.0);
    double delta_millis = (double)in->delta_time_ms / R(1000.0);
    double mx = (double)in->delta_mouse_x, my = (double)in->delta_mouse_y;
    uint32_t keys = in->keys;
    double direction[MAXD] = { R(0.0), R(0.0), R(0.0), R(0.0) };
    static const double Z[3] = { R(0.0), R(0.0), R(1.0) };
    if (D == 3 && kind == EO_CAMERA_PITCH_YAW_3) {
        if (!(mx * mx + my * my <= R(0.0))) {                                   /* update_rotation :94-108 */
            double dir2[2] = { mx * sens, my * sens };
            rotate_axis_angle(Z, -dir2[0], cam.forward); normalize_mut(3, cam.forward);     /* rotate_yaw_static :110-114 */
            rotate_axis_angle(Z, -dir2[0], cam.up); normalize_mut(3, cam.up);
            rotate_pitch_static(cam.forward, cam.up, -dir2[1], 1);
        }
    } else if (D == 3) {                                                     /* FreeCamera3::update_rotation :299-324 */
        double dir2[2] = { mx * sens, my * sens };
        double roll = R(0.0);
        if (keys & EO_KEY_Q) roll -= R(1.0);
        if (keys & EO_KEY_E) roll += R(1.0);
        roll *= delta_millis * R(2.0);
        if (dir2[0] != R(0.0)) { rotate_axis_angle(cam.up, -dir2[0], cam.forward); normalize_mut(3, cam.forward); }
        if (dir2[1] != R(0.0)) rotate_pitch_static(cam.forward, cam.up, -dir2[1], 0);
        if (roll != R(0.0)) { rotate_axis_angle(cam.forward, roll, cam.up); normalize_mut(3, cam.up); }
    } else {                                                                 /* FreeCamera4::update_rotation d4:68-126 */
        double angle = R(0.0);
        if (keys & EO_KEY_C) angle += R(1.0);
        if (keys & EO_KEY_M) angle -= R(1.0);
        if (angle != R(0.0)) {
            angle *= delta_millis * R(2.0);
            int ax[4] = { !!(keys & EO_KEY_I), !!(keys & EO_KEY_O), !!(keys & EO_KEY_K), !!(keys & EO_KEY_L) };
            if (ax[0] + ax[1] + ax[2] + ax[3] == 2) {
                double storage[16];                                          /* column-major: storage[col*4 + row] */
                for (int idx = 0; idx < 16; idx++) storage[idx] = (idx / 4 == idx % 4) ? R(1.0) : R(0.0);
                for (int idx = 0; idx < 16; idx++) {
                    int row = idx / 4, column = idx % 4;                     /* the reference's names for them */
                    if (ax[row] && ax[column]) storage[idx] = row == column ? eo_cos(angle) : (row < column ? -eo_sin(angle) : eo_sin(angle));
                }
                double rot[4][4], nm[4][4], nt[4][4], ana[4];
                for (int c = 0; c < 4; c++) for (int r = 0; r < 4; r++) rot[r][c] = storage[c * 4 + r];
                find_orthonormal_4(cam.forward, cam.left, cam.up, ana);
                for (int r = 0; r < 4; r++) { nm[r][0] = cam.forward[r]; nm[r][1] = cam.left[r]; nm[r][2] = cam.up[r]; nm[r][3] = ana[r]; }
                for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) nt[r][c] = nm[c][r];
                double *vs[3] = { cam.forward, cam.left, cam.up };
                for (int q = 0; q < 3; q++) {
                    double a1[4], a2[4];
                    mat4_mul_vec(nt, vs[q], a1); mat4_mul_vec(rot, a1, a2); mat4_mul_vec(nm, a2, vs[q]);
                }
                find_orthonormal_4(cam.forward, cam.left, cam.up, ana);
                find_orthonormal_4(cam.up, cam.left, ana, cam.forward); normalize_mut(4, cam.forward);       /* reorthonormalize_4 util.rs:310-322 */
                find_orthonormal_4(cam.forward, cam.up, ana, cam.left); normalize_mut(4, cam.left);
                find_orthonormal_4(cam.forward, ana, cam.left, cam.up); normalize_mut(4, cam.up);
            }
        }
    }
    double distance = speed * delta_millis;
    if (distance == R(0.0)) { *camera = cam; return 0; }
    double left[MAXD], vertical[MAXD], ana[4] = { 0, 0, 0, 0 };
    if (D == 3) {
        v_cross3(cam.up, cam.forward, left); normalize_mut(3, left);         /* get_left :60-62 */
        v_copy(3, kind == EO_CAMERA_PITCH_YAW_3 ? Z : cam.up, vertical);
    } else {
        v_copy(4, cam.left, left); v_copy(4, cam.up, vertical);
        find_orthonormal_4(cam.forward, cam.left, cam.up, ana);
    }
    if (keys & EO_KEY_W) v_add(D, direction, cam.forward, direction);
    if (keys & EO_KEY_S) v_sub(D, direction, cam.forward, direction);
    if (keys & EO_KEY_A) v_add(D, direction, left, direction);
    if (keys & EO_KEY_D) v_sub(D, direction, left, direction);
    if (keys & EO_KEY_LSHIFT) v_add(D, direction, vertical, direction);
    if (keys & EO_KEY_LCONTROL) v_sub(D, direction, vertical, direction);
    if (D == 4 && (keys & EO_KEY_Q)) v_add(D, direction, ana, direction);
    if (D == 4 && (keys & EO_KEY_E)) v_sub(D, direction, ana, direction);
    if (v_nsq(D, direction) != R(0.0)) {
        distance *= v_norm(D, direction);
        normalize_mut(D, direction);
        double nl[MAXD], nd[MAXD];
        int rc = eo_trace_path_unknown(s, cam.location, direction, distance, nl, nd);
        if (rc < 0) return -1;
        if (rc == 1) {
            double rotation_scale = angle_between(D, direction, nd);
            if (D == 3) {
                if (!approx_eq_ulps(rotation_scale, R(0.0), 8)) {               /* d3 camera.rs:230-239 */
                    double axis[3], aa[3], r[3];
                    v_cross3(direction, nd, axis);
                    v_scale(3, axis, rotation_scale, aa);
                    quat_t q = quat_from_axisangle(aa);
                    quat_rotate(q, cam.forward, r); v_copy(3, r, cam.forward);
                    quat_rotate(q, cam.up, r); v_copy(3, r, cam.up);
                }
            } else if (!approx_eq_ulps(rotation_scale, R(0.0), 4 * 8)) return 1;   /* unimplemented!() */
            v_copy(D, nl, cam.location);
        }
    }
    *camera = cam;
    return 0;
}

/* camera ray: d3/entity/camera.rs:155-185 (identical in FreeCamera3 :360-390), d4/entity/camera.rs:146-176 */
static void camera_ray(const eo_camera *cam, int sx, int sy, int sw, int sh, double *point, double *vector) {
    int D = cam->dim;
    double rel_x = (double)(sx - sw / 2) + (double)(1 - sw % 2) / R(2.0);
    double rel_y = (double)(sy - sh / 2) + (double)(1 - sh % 2) / R(2.0);
    double w = (double)sw, h = (double)sh;
    double right[MAXD];
    if (D == 3) { double cr[MAXD]; v_cross3(cam->forward, cam->up, cr); v_normalize(3, cr, right); }
    else v_neg(D, cam->left, right);
    FL(2 + 1 + 3 + 1 + 6 * D); FLD(2 + 2 + 2);
    double fov_rad = EO_PI_C * (double)cam->fov_deg / R(180.0);
    double dist = sqrt(w * w + h * h) / (R(2.0) * eo_tan(fov_rad / R(2.0)));
    double p[MAXD];
    for (int i = 0; i < D; i++) {
        double center = cam->location[i] + cam->forward[i] * dist;
        p[i] = center + (cam->up[i] * rel_y) + (right[i] * rel_x);
    }
    double dlt[MAXD];
    v_sub(D, p, cam->location, dlt);
    v_normalize(D, dlt, vector);
    v_copy(D, cam->location, point);
}

/* trace_screen_point + trace_unknown (universe/mod.rs:253-271, 371-397) */
static void trace_screen_point(tctx *t, const eo_camera *cam, int x, int y, int w, int h, double *rgb, double *hit) {
    double point[MAXD], vector[MAXD];
    camera_ray(cam, x, y, w, h, point, vector);
    const obj *belongs_to = material_at(t, point);
    if (!belongs_to) {
        if (hit) *hit = -R(1.0);
        if ((x / 8 + y / 8) % 2 == 0) { rgb[0] = R(0.0); rgb[1] = R(0.0); rgb[2] = R(0.0); }
        else { rgb[0] = R(1.0); rgb[1] = R(0.0); rgb[2] = R(1.0); }
        return;
    }
    double dir[MAXD];
    v_copy(t->D, vector, dir);
    material_enter(t, belongs_to->o1, dir);
    rgba_t white = { R(1.0), R(1.0), R(1.0), R(1.0) };
    pre_t background = into_premultiplied(white);
    pre_t foreground = into_premultiplied(trace(t, cam->max_depth, belongs_to, point, dir, hit));
    rgba_t out = from_premultiplied(blend_pre(BL_OVER, foreground, background));
    rgb[0] = out.r; rgb[1] = out.g; rgb[2] = out.b;
}

typedef struct {
    const eo_scene *scene; const eo_camera *cam; const eo_frame *frame;
    uint8_t *rgb; double *hit_t;
    atomic_uint next_row;
    eo_stats stats; pthread_mutex_t mu;
} job_t;

static void *worker(void *arg) {
    job_t *j = arg;
    tctx t; memset(&t, 0, sizeof t);
    t.scene = j->scene; t.D = j->scene->dim; t.time_ms = j->frame->time_ms;
    t.arena_cap = EO_ARENA_BYTES; t.arena = malloc(t.arena_cap);
    uint32_t W = j->frame->width, H = j->frame->height;
    uint32_t hw = W / 2, hh = H / 2;
    for (;;) {
        uint32_t y = atomic_fetch_add(&j->next_row, 1);
        if (y >= j->frame->row_end) break;
        for (uint32_t x = 0; x < W; x++) {
            size_t idx = (size_t)(y - j->frame->row_begin) * W + x;
            double rgb[3], hit = -R(1.0);
            int surrounding = j->frame->debug_crosshair &&
                ((x == hw && (y == hh - 1 || y == hh + 1)) || (y == hh && (x == hw - 1 || x == hw + 1)));
            if (surrounding) { rgb[0] = R(1.0); rgb[1] = R(0.0); rgb[2] = R(0.0); }   /* Rgb::new_u8(255,0,0) */
            else trace_screen_point(&t, j->cam, (int)x, (int)y, (int)W, (int)H, rgb, &hit);
            j->rgb[idx * 3 + 0] = to_u8(&t, rgb[0]);
            j->rgb[idx * 3 + 1] = to_u8(&t, rgb[1]);
            j->rgb[idx * 3 + 2] = to_u8(&t, rgb[2]);
            if (j->hit_t) j->hit_t[idx] = hit;
        }
    }
#ifdef EO_COUNT_FLOPS
    pthread_mutex_lock(&eo_fl_mu);
    for (int k = 0; k < 4; k++) { eo_fl_total[k] += eo_fl[k]; eo_fl[k] = 0; }
    pthread_mutex_unlock(&eo_fl_mu);
#endif
    pthread_mutex_lock(&j->mu);
    j->stats.rays += t.stats.rays; j->stats.bg_samples += t.stats.bg_samples;
    j->stats.nan_pixels += t.stats.nan_pixels; j->stats.errors += t.stats.errors; j->stats.spins += t.stats.spins;
    pthread_mutex_unlock(&j->mu);
    free(t.arena);
    return NULL;
}

/* -DEO_COUNT_FLOPS builds: operations counted since the last call (add/sub/mul, div, sqrt, transcendental calls); zeros otherwise */
void eo_flops_take(unsigned long long out[4]) {
    pthread_mutex_lock(&eo_fl_mu);
    for (int k = 0; k < 4; k++) { out[k] = eo_fl_total[k]; eo_fl_total[k] = 0; }
    pthread_mutex_unlock(&eo_fl_mu);
}
int eo_build_flags(void) {
    int f = 0;
#ifdef EO_COUNT_FLOPS
    f |= 1;
#endif
#ifdef EO_USE_LIBM
    f |= 2;
#endif
    return f;
}

int eo_render(const eo_scene *s, const eo_camera *cam, const eo_frame *f, int threads, uint8_t *rgb, double *hit_t, eo_stats *stats) {
    if (!s->background || cam->dim != s->dim) return -1;
    if (f->row_end > f->height || f->row_begin > f->row_end) return -2;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    job_t j; memset(&j, 0, sizeof j);
    j.scene = s; j.cam = cam; j.frame = f; j.rgb = rgb; j.hit_t = hit_t;
    atomic_init(&j.next_row, f->row_begin);
    pthread_mutex_init(&j.mu, NULL);
    pthread_t th[256];
    for (int i = 0; i < threads; i++) pthread_create(&th[i], NULL, worker, &j);
    for (int i = 0; i < threads; i++) pthread_join(th[i], NULL);
    pthread_mutex_destroy(&j.mu);
    if (stats) *stats = j.stats;
    return 0;
}

/* ------------------------------------------------------------------ white-box test hooks */

static void tctx_init(tctx *t, const eo_scene *s, int D) {
    memset(t, 0, sizeof *t);
    t->scene = s; t->D = D; t->arena_cap = EO_ARENA_BYTES; t->arena = malloc(t->arena_cap);
}
int eo_test_intersect(const eo_scene *s, int shape, const double *loc, const double *dir, eo_intersection *out, int max_out) {
    obj *sh = get_obj(s, shape, K_SHAPE);
    if (!sh) return -1;
    tctx t; tctx_init(&t, s, s->dim);
    provider *p = universe_intersect(&t, loc, dir, sh);
    int n = 0;
    hit_t h;
    while (n < max_out && provider_get(&t, p, n, &h)) out[n++] = h;
    free(t.arena);
    return n;
}
/* like eo_test_intersect, for at most max_out elements; *spins = CSG streams that hit the runaway guard on the way */
int eo_test_intersect_spins(const eo_scene *s, int shape, const double *loc, const double *dir, eo_intersection *out, int max_out, uint64_t *spins) {
    obj *sh = get_obj(s, shape, K_SHAPE);
    if (!sh) return -1;
    tctx t; tctx_init(&t, s, s->dim);
    provider *p = universe_intersect(&t, loc, dir, sh);
    int n = 0;
    hit_t h;
    while (n < max_out && provider_get(&t, p, n, &h)) out[n++] = h;
    *spins = t.stats.spins;
    free(t.arena);
    return n;
}
int eo_test_is_point_inside(const eo_scene *s, int shape, const double *point) {
    obj *sh = get_obj(s, shape, K_SHAPE);
    if (!sh) return -1;
    return shape_inside(s->dim, sh, point);
}
double eo_test_angle_between(int dim, const double *a, const double *b) { return angle_between(dim, a, b); }
void eo_test_combine_palette_color(const double *a, const double *b, double ratio, double *out) {
    rgba_t o = combine_palette_color(rgba_of(a), rgba_of(b), ratio);
    out[0] = o.r; out[1] = o.g; out[2] = o.b; out[3] = o.a;
}
double eo_test_remainder_f(double a, double b) { return remainder_f(a, b); }
int64_t eo_test_remainder_i(int64_t a, int64_t b) { return remainder_i(a, b); }
void eo_test_material_enter(const eo_scene *s, int material, double *direction, int exit_) {
    obj *m = get_obj(s, material, K_MATERIAL);
    if (!m) return;
    tctx t; tctx_init(&t, s, s->dim);
    if (exit_) material_exit(&t, m, direction); else material_enter(&t, m, direction);
    free(t.arena);
}
void eo_test_general_rotation(int dim, const double *self, const double *other, double angle, double *vec) { general_rotation(dim, self, other, angle, vec); }
/* Fresnel ratio / Snell direction for one hit: `direction` is the ray's direction at the hit, `normal_closer` the surface normal on the
 * side the ray comes from (TracingContext::normal_closer, shape.rs:111-125) */
double eo_test_fresnel(int dim, double index_inside, double index_outside, const double *direction, const double *normal_closer, int exiting) {
    tctx t; memset(&t, 0, sizeof t); t.D = dim;
    obj p; memset(&p, 0, sizeof p); p.sub = RATIO_FRESNEL; p.p0 = index_inside; p.p1 = index_outside;
    trace_ctx c; memset(&c, 0, sizeof c);
    for (int i = 0; i < dim; i++) { c.intersection.direction[i] = direction[i]; c.normal_closer[i] = normal_closer[i]; }
    c.exiting = exiting;
    return reflection_ratio(&t, &p, &c);
}
void eo_test_snell(int dim, double index, const double *direction, const double *normal_closer, int exiting, double *out) {
    tctx t; memset(&t, 0, sizeof t); t.D = dim;
    obj p; memset(&p, 0, sizeof p); p.sub = THR_SNELL; p.p0 = index;
    trace_ctx c; memset(&c, 0, sizeof c);
    for (int i = 0; i < dim; i++) { c.intersection.direction[i] = direction[i]; c.normal_closer[i] = normal_closer[i]; }
    c.exiting = exiting;
    double o[MAXD];
    threshold_direction(&t, &p, &c, o);
    for (int i = 0; i < dim; i++) out[i] = o[i];
}
void eo_test_to_pixel(const double *rgba, uint8_t *px) { rgba_t c = rgba_of(rgba); to_pixel4(NULL, c, px); }
void eo_test_blend(const char *name, const double *src, const double *dst, double *out) {
    int fn = -1;
    for (int i = 0; i < BL_COUNT; i++) if (!strcmp(name, BLEND_NAMES[i])) fn = i;
    rgba_t o = { NAN, NAN, NAN, NAN };
    if (fn >= 0 && fn != BL_RATIO) o = blend_rgba(fn, rgba_of(src), rgba_of(dst));
    out[0] = o.r; out[1] = o.g; out[2] = o.b; out[3] = o.a;
}
void eo_test_math(int fn, const double *x, const double *y, double *out, int n) {
    for (int i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = eo_acos(x[i]); break;
        case 1: out[i] = eo_asin(x[i]); break;
        case 2: out[i] = eo_sin(x[i]); break;
        case 3: out[i] = eo_cos(x[i]); break;
        case 4: out[i] = eo_tan(x[i]); break;
        case 5: out[i] = eo_atan2(x[i], y[i]); break;
        case 6: out[i] = sqrt(x[i]); break;
        case 7: out[i] = x[i] / y[i]; break;
        case 8: out[i] = fmod(x[i], y[i]); break;
        default: out[i] = NAN;
        }
    }
}
double eo_test_perlin(uint32_t seed, const double *p) { uint8_t perm[512]; perlin_build_perm(seed, perm); return perlin4(perm, p[0], p[1], p[2], p[3]); }
void eo_test_ray(const eo_camera *cam, int x, int y, int w, int h, double *point, double *vector) { camera_ray(cam, x, y, w, h, point, vector); }
