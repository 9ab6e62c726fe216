/*
 * flat_scene.h -- the device-side scene: one contiguous blob of 8-byte words.
 *
 * The reference keeps a universe as boxed trait objects (Vec<Box<Entity>>, Arc<Box<Shape>> CSG
 * trees, Arc<dyn Fn> surface providers: /root/reference/src/universe/d3/mod.rs:24-29,
 * universe/entity/shape.rs:499-505, universe/entity/surface.rs:39-44).  For the GPU the host
 * flattens it once into tables of fixed-size records addressed by word offsets, so that a
 * workgroup can stage the whole scene into LDS with one coalesced copy and every lane of a
 * wave reads the same record at the same time (LDS broadcast, no bank conflicts).
 *
 *  - shapes      : each entity's CSG tree in POST-ORDER ("shape program"); op i's subtree is
 *                  ops[first..i]; composite children: b = i-1, a = ops[b].first-1.  Half-space
 *                  chains are collapsed into single ops (EU_SH_CHAIN_*).
 *  - materials   : Vacuum | LinearSpace (list of per-component RPN programs compiled from the
 *                  meval expressions, material.rs:59-163)
 *  - surfaces    : ComposableSurface = ratio/direction provider ids + a post-order colour program
 *  - textures    : RGBA8 texels stay in HBM; the blob holds the device pointer
 *
 * Everything is plain data: no pointers except texel base addresses patched at upload.
 */
#ifndef EU_FLAT_SCENE_H
#define EU_FLAT_SCENE_H

#include "eu_platform.h"

#define EU_FLAT_MAGIC 0x45554346u /* "EUCF" */
#define EU_FLAT_VERSION 1u

enum EuShapeKind : uint32_t {
    EU_SH_VOID = 0, EU_SH_SPHERE = 1, EU_SH_PLANE = 2, EU_SH_HALFSPACE = 3, EU_SH_CYLINDER = 4,
    EU_SH_UNION = 8, EU_SH_INTERSECTION = 9, EU_SH_COMPLEMENT = 10, EU_SH_SYMDIFF = 11,
    /* a left-fold Union / Intersection whose leaves are all half-spaces or hyperplanes (cuboid,
     * hypercuboid, wall sets), collapsed into ONE op: `count` leaves, parameters contiguous with
     * stride 2*D+2 (HALFSPACE layout; a Hyperplane is stored with signum = NaN, nflip = n).
     * Behaves like a leaf that produces up to `count` hits. */
    EU_SH_CHAIN_UNION = 16, EU_SH_CHAIN_INTERSECTION = 17,
    /* an Intersection chain of exactly 2*D half-spaces whose leaf k has the normal +-e_(k/2) exactly (components +-1 and
     * +-0) and a finite non-zero constant: what HalfSpace3::cuboid / HalfSpace4::hypercuboid build (d3/entity/shape.rs:17-66).
     * Same semantics as EU_SH_CHAIN_INTERSECTION; the device may replace every dot product with the normal by one
     * multiplication (trace_device.h, chain_matrices_box). */
    EU_SH_CHAIN_BOX = 18,
    /* the same with at least one constant that is +-0 (a face through a coordinate plane): the one-product form differs from the dot
     * product only when the product with the normal's +-1 component is -0 next to a zero constant; the device watches for exactly that
     * (chain_matrices_box<D, true>) */
    EU_SH_CHAIN_BOX0 = 19,
    /* Not a shape: a guard in front of a bounded composite subtree that is not an entity's root (the loader puts one where the
     * subtree's bounding sphere is clearly smaller than its parent's).  `first` holds the index of the subtree's root op, `param` its
     * entry in the bounds table.  A wave whose rays all miss the sphere pushes an empty list and continues behind the root op (the
     * subtree's stream is empty: same argument as for an entity's bound); the containment test does the same for points outside. */
    EU_SH_SKIP = 24
};
#define EU_CHAIN_MAX 8
enum EuMaterialKind : uint32_t { EU_MAT_VACUUM = 0, EU_MAT_LINEAR = 1 };
enum EuRatioKind : uint32_t { EU_RATIO_UNIFORM = 0, EU_RATIO_FRESNEL = 1 };
enum EuThresholdKind : uint32_t { EU_THR_IDENTITY = 0, EU_THR_SNELL = 1 };
enum EuColorKind : uint32_t { EU_COL_UNIFORM = 0, EU_COL_BLEND = 1, EU_COL_ILLUM_GLOBAL = 2, EU_COL_ILLUM_DIR = 3,
                              EU_COL_PERLIN = 4, EU_COL_TEXTURE = 5 };
enum EuBlend : uint32_t { EU_BL_OVER = 0, EU_BL_INSIDE, EU_BL_OUTSIDE, EU_BL_ATOP, EU_BL_XOR, EU_BL_PLUS, EU_BL_MULTIPLY,
                          EU_BL_SCREEN, EU_BL_OVERLAY, EU_BL_DARKEN, EU_BL_LIGHTEN, EU_BL_DODGE, EU_BL_BURN,
                          EU_BL_HARD_LIGHT, EU_BL_SOFT_LIGHT, EU_BL_DIFFERENCE, EU_BL_EXCLUSION, EU_BL_RATIO, EU_BL_COUNT };
enum EuTexKind : uint32_t { EU_TEX_NEAREST = 0, EU_TEX_LINEAR = 1 };
enum EuRpn : uint32_t { EU_RPN_CONST = 0, EU_RPN_VAR, EU_RPN_ADD, EU_RPN_SUB, EU_RPN_MUL, EU_RPN_DIV, EU_RPN_REM,
                        EU_RPN_POW, EU_RPN_NEG, EU_RPN_FN };
enum EuFn : uint32_t { EU_FN_SQRT = 0, EU_FN_ABS, EU_FN_FLOOR, EU_FN_CEIL, EU_FN_MIN, EU_FN_MAX, EU_FN_SIN, EU_FN_COS,
                       EU_FN_TAN, EU_FN_ASIN, EU_FN_ACOS, EU_FN_ATAN, EU_FN_ATAN2, EU_FN_SIGNUM, EU_FN_COUNT };

/* header: 16 words */
struct EuFlatHeader {
    uint32_t magic, version;
    uint32_t dim, n_words;
    uint32_t n_ops, off_ops;                 /* EuShapeOp, 1 word each */
    uint32_t n_entities, off_entities;       /* EuFlatEntity, 2 words each */
    uint32_t n_materials, off_materials;     /* EuFlatMaterial, 1 word each */
    uint32_t n_transforms, off_transforms;   /* 8 words each: fwd[4], inv[4] = code_off | len<<32 */
    uint32_t n_code, off_code;               /* RPN words */
    uint32_t n_surfaces, off_surfaces;       /* EuFlatSurface, 8 words each */
    uint32_t n_color_ops, off_color_ops;     /* EuFlatColorOp, 16 words each */
    uint32_t n_mapped, off_mapped;           /* EuFlatMapped, 8 words each */
    uint32_t n_perlin, off_perlin;           /* 64 words (512 B permutation) each */
    uint32_t background, hit_cap;            /* background mapped-texture id; per-ray hit-stack entries: bits 0..15 what the wavefront kernels reserve, bits 16..31 the strict worst case (scene_host.cpp: HitUse) */
    uint32_t list_depth, color_depth;        /* max simultaneous hit lists / colour stack depth */
    uint32_t rpn_depth, flags;
    uint32_t n_params, off_params;           /* leaf parameter doubles */
    uint32_t n_bounds, off_bounds;           /* bounding spheres of bounded entities: c[D], r2, far2 (D+2 doubles each) */
};
#define EU_FLAT_HEADER_WORDS 16

/* one shape-program op (1 word).  Leaf parameter layouts (doubles at params+param):
 *   SPHERE    c[D], r, r*r
 *   PLANE     n[D], constant
 *   HALFSPACE n[D], constant, signum, nflip[D] (= n * -signum, shape.rs:860)
 *   CYLINDER  c[D], axis[D] (normalised), r, r*r                                           */
struct EuShapeOp {
    uint8_t kind;
    uint8_t count;       /* chain ops: number of leaves (2..EU_CHAIN_MAX); composite ops: 1 if the subtree's first op is its guard (EU_SH_SKIP) */
    uint16_t first;      /* index (within the ops table) of the first op of this subtree */
    uint32_t param;      /* word offset into the params table */
};

struct EuFlatEntity {
    uint16_t shape_first, shape_root;
    uint16_t material;
    int16_t surface;     /* -1: no surface (Void / new_without_surface) */
    uint32_t max_hits;   /* static bound of hit-stack use for this entity */
    uint32_t bound;      /* index into the bounds table, 0xffffffff: unbounded (never culled) */
};

struct EuFlatMaterial {
    uint32_t kind;       /* EuMaterialKind | n_transforms << 8 */
    uint32_t first_transform;
};

struct EuFlatSurface {
    uint32_t ratio_kind, thr_kind;
    uint32_t color_first, color_root;
    double ratio_p0, ratio_p1;   /* uniform: ratio ; fresnel: index_inside, index_outside */
    double thr_p0, thr_p0_inv;   /* snell: n, 1/n */
    double reserved[2];
};

struct EuFlatColorOp {
    uint32_t kind, fn;           /* EuColorKind ; EuBlend for BLEND */
    uint32_t aux, reserved;      /* mapped-texture id / perlin table id */
    double c0[4];                /* uniform colour | light colour */
    double c1[4];                /* dark colour */
    double v[4];                 /* light direction | (size, speed) | (ratio) */
    double pad[2];
};

struct EuFlatMapped {
    uint32_t tex_kind, uv_kind;
    uint32_t w, h;
    uint64_t texels;             /* device address of RGBA8 texels (patched at upload) */
    double center[3];
    double wd, hd;               /* (double)w, (double)h */
};

static_assert(sizeof(EuShapeOp) == 8, "1 word");
static_assert(sizeof(EuFlatEntity) == 16, "2 words");
static_assert(sizeof(EuFlatMaterial) == 8, "1 word");
static_assert(sizeof(EuFlatSurface) == 64, "8 words");
static_assert(sizeof(EuFlatColorOp) == 128, "16 words");
static_assert(sizeof(EuFlatMapped) == 64, "8 words");
static_assert(sizeof(EuFlatHeader) == EU_FLAT_HEADER_WORDS * 8, "16 words");

#endif
