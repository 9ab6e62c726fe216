/*
 * eo_oracle.h -- ORACLE (test infrastructure, NOT product code).
 *
 * CPU restatement, in plain C, of euclider's per-pixel trace loop
 * (Environment::render -> Universe::trace / trace_closest / intersect -> shape intersectors ->
 * Surface::get_color).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (euclider_amd/) never includes, links or calls it.
 *
 * PARITY STATUS: pinned against the reference's own known-answer tests
 * (/root/reference/src/universe/entity/shape.rs:1048-1148, src/util.rs:947-969,1007-1037) in
 * tests/test_oracle_kat.py.  The Rust reference cannot be built here (no cargo/rustc), and the
 * arithmetic that lives in un-vendored crates (nalgebra 0.8.2, palette 0.2.1, noise 0.4.1,
 * meval 0.1.0, image 0.18.0 -- Cargo.lock) is restated from their published algorithms; those
 * parts are "parity unpinned" (list: DESIGN.md "Unverified third-party semantics").
 *
 * The scene is built through constructor calls that mirror the reference's JSON constructor
 * registry (src/scene.rs:620-1408); oracle/scene_loader.py walks a scene JSON and issues them.
 * Every object is referred to by an integer handle (>= 0); negative returns are errors.
 */
#ifndef EO_ORACLE_H
#define EO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct eo_scene eo_scene;

enum { EO_OP_UNION = 0, EO_OP_INTERSECTION = 1, EO_OP_COMPLEMENT = 2, EO_OP_SYMDIFF = 3 };
enum { EO_TEX_NEAREST = 0, EO_TEX_LINEAR = 1 };

typedef struct {
    int dim;
    double location[4], forward[4], up[4], left[4];
    uint32_t fov_deg;       /* u8 in the reference (d3/entity/camera.rs:37,49) */
    uint32_t max_depth;     /* 10 in the reference (d3/entity/camera.rs:50) */
} eo_camera;

typedef struct {
    uint32_t width, height;       /* buffer dims (window / resolution), universe/mod.rs:308-309 */
    uint32_t row_begin, row_end;  /* rows [row_begin,row_end) are rendered */
    uint64_t time_ms;             /* (time*1000).as_secs(), d3/entity/surface.rs:32 */
    int debug_crosshair;          /* context.debugging, universe/mod.rs:321-333 */
} eo_frame;

typedef struct {
    uint64_t rays;          /* trace() calls with depth>0 (reach trace_closest) */
    uint64_t bg_samples;    /* background().get_color calls */
    uint64_t nan_pixels;    /* float->u8 casts that would panic in the reference */
    uint64_t errors;        /* other would-panic conditions (no material, csg runaway, ...) */
    uint64_t spins;         /* of those: CSG streams the reference would never finish computing (its render would hang) */
} eo_stats;

typedef struct { double location[4], direction[4], normal[4], distance; } eo_intersection;

eo_scene *eo_scene_new(int dim);
void eo_scene_free(eo_scene *);
const char *eo_last_error(const eo_scene *);

/* shapes (universe/entity/shape.rs, d3/entity/shape.rs, d4/entity/shape.rs) */
int eo_shape_void(eo_scene *);
int eo_shape_sphere(eo_scene *, const double *center, double radius);
int eo_shape_hyperplane(eo_scene *, const double *normal, double constant);
int eo_shape_hyperplane_with_point(eo_scene *, const double *normal, const double *point);
int eo_shape_hyperplane_with_vectors(eo_scene *, const double *a, const double *b, const double *point);
int eo_shape_halfspace(eo_scene *, int plane, double sign);
int eo_shape_halfspace_with_point(eo_scene *, int plane, const double *point);
int eo_shape_cuboid(eo_scene *, const double *center, const double *dimensions);
int eo_shape_cylinder(eo_scene *, const double *center, const double *direction, double radius);
int eo_shape_cylinder_with_height(eo_scene *, const double *center, const double *direction, double radius, double height);
int eo_shape_composable_of(eo_scene *, const int *shapes, int n, int op);

/* materials (universe/entity/material.rs) */
int eo_material_vacuum(eo_scene *);
int eo_transformation_expr(eo_scene *, const char *expression, const char *inverse_expression);
int eo_component_transformation(eo_scene *, const int *exprs, int n);
int eo_material_linear_space(eo_scene *, const char *legend, const int *transformations, int n);

/* surface providers (universe/entity/surface.rs, d3/entity/surface.rs, d4/entity/surface.rs) */
int eo_reflection_ratio_uniform(eo_scene *, double ratio);
int eo_reflection_ratio_fresnel(eo_scene *, double index_inside, double index_outside);
int eo_reflection_direction_specular(eo_scene *);
int eo_threshold_direction_identity(eo_scene *);
int eo_threshold_direction_snell(eo_scene *, double refractive_index);
int eo_blend_function(eo_scene *, const char *name, double ratio); /* "over".."exclusion", "ratio" */
int eo_color_uniform(eo_scene *, const double *rgba);
int eo_color_blend(eo_scene *, int source, int destination, int blend_function);
int eo_color_illumination_global(eo_scene *, const double *light, const double *dark);
int eo_color_illumination_directional(eo_scene *, const double *direction, const double *light, const double *dark);
int eo_color_perlin_hue(eo_scene *, uint32_t seed, double size, double speed);
int eo_color_texture(eo_scene *, int mapped_texture);
int eo_uv_sphere(eo_scene *, const double *center3);
int eo_uv_derank(eo_scene *, int uvfn);
int eo_texture_image(eo_scene *, int kind, uint32_t w, uint32_t h, const uint8_t *rgba8);
int eo_mapped_texture(eo_scene *, int uvfn, int texture);
int eo_surface_composable(eo_scene *, int ratio, int reflection_direction, int threshold_direction, int color);

/* entities / universe (universe/entity/mod.rs, d3/mod.rs, d4/mod.rs) */
int eo_entity(eo_scene *, int shape, int material, int surface /* -1 = none */);
int eo_entity_void(eo_scene *, int material);
int eo_universe(eo_scene *, const eo_camera *camera, const int *entities, int n, int background);

/* colour helpers used by the loader (palette 0.2.1) */
void eo_rgba_from_hsva(double hue, double s, double v, double a, double *rgba_out);

int eo_default_camera(int dim, const double *location_or_null, eo_camera *out);
int eo_scene_camera(const eo_scene *, eo_camera *out);

/* the hot path: Environment::render (universe/mod.rs:300-357) */
/* Universe::trace_path_unknown (universe/mod.rs:273-286): camera motion through surfaces and materials.
 * Returns 1 = Some, 0 = None, -1 = more than 4096 surface crossings. */
int eo_trace_path_unknown(const eo_scene *, const double *location, const double *direction, double distance,
                          double *out_location, double *out_direction);

/* Camera::update (d3/entity/camera.rs:191-245,396-451; d4/entity/camera.rs:182-241). */
enum { EO_CAMERA_PITCH_YAW_3 = 0, EO_CAMERA_FREE_3 = 1, EO_CAMERA_FREE_4 = 2 };
enum { EO_KEY_W = 1, EO_KEY_S = 2, EO_KEY_A = 4, EO_KEY_D = 8, EO_KEY_LSHIFT = 16, EO_KEY_LCONTROL = 32, EO_KEY_Q = 64, EO_KEY_E = 128,
       EO_KEY_C = 256, EO_KEY_M = 512, EO_KEY_I = 1024, EO_KEY_O = 2048, EO_KEY_K = 4096, EO_KEY_L = 8192 };
typedef struct {
    uint32_t keys;                       /* context.pressed_keys() */
    int32_t delta_mouse_x, delta_mouse_y;
    uint64_t delta_time_ms;              /* (delta_time * 1000).as_secs() */
    double mouse_sensitivity, speed;     /* 0 = the reference's 0.01 / 10.0 */
} eo_input;
/* 0 = updated, 1 = the reference reaches unimplemented!() (4-D, direction turned), -1 = path step cap */
int eo_camera_update(const eo_scene *, int kind, eo_camera *, const eo_input *);

/* variant builds (oracle/Makefile): libeo_oracle_flops.so counts the f64 operations of the algorithm,
 * libeo_oracle_libm.so takes the elementary functions from the platform libm.  eo_build_flags: bit 0 flops, bit 1 libm. */
void eo_flops_take(unsigned long long out[4]);
int eo_build_flags(void);

int eo_render(const eo_scene *, const eo_camera *, const eo_frame *, int threads,
              uint8_t *rgb_out /* (row_end-row_begin)*width*3 */,
              double *hit_t /* optional, per pixel: distance of the primary ray's closest hit, -1 if none */,
              eo_stats *stats);

/* white-box entry points for the reference's known-answer tests */
int eo_test_intersect(const eo_scene *, int shape, const double *location, const double *direction,
                      eo_intersection *out, int max_out);
int eo_test_is_point_inside(const eo_scene *, int shape, const double *point);
double eo_test_angle_between(int dim, const double *a, const double *b);
void eo_test_combine_palette_color(const double *a, const double *b, double ratio, double *out);
double eo_test_remainder_f(double a, double b);
int64_t eo_test_remainder_i(int64_t a, int64_t b);
void eo_test_material_enter(const eo_scene *, int material, double *direction, int exit);
void eo_test_general_rotation(int dim, const double *self, const double *other, double angle, double *vec);
double eo_test_fresnel(int dim, double index_inside, double index_outside, const double *direction, const double *normal_closer, int exiting);
void eo_test_snell(int dim, double index, const double *direction, const double *normal_closer, int exiting, double *out);
void eo_test_to_pixel(const double *rgba, uint8_t *px);
void eo_test_blend(const char *name, const double *src, const double *dst, double *out);
void eo_test_math(int fn, const double *x, const double *y, double *out, int n);
double eo_test_perlin(uint32_t seed, const double *xyzw);
void eo_test_ray(const eo_camera *, int x, int y, int w, int h, double *point, double *vector);

#ifdef __cplusplus
}
#endif
#endif
