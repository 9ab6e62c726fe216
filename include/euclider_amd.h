/*
 * euclider_amd.h -- C ABI of the MI355X-native trace path (libeuclider_amd.so).
 *
 * The reference (Limeth/euclider, pure Rust) has no FFI; the narrowest seam around its hot path
 * is the `Environment` trait object the window loop holds (/root/reference/src/universe/mod.rs:289-360):
 *     fn max_depth(&self) -> u32;                                              (:290)
 *     fn trace_screen_point(&self, time, max_depth, x, y, w, h, debug) -> Rgb<F>;  (:291-299)
 *     fn render(&self, dimensions, time, threads, context) -> RawImage2d<u8>;   (:300-357)
 * plus the loader that produces it, `Parser::parse::<Box<Environment>>(&str)`
 * (/root/reference/src/scene.rs:1466-1478).  A Rust `impl Environment for GpuUniverse` would bind
 * exactly the functions below (INTEGRATION.md shows the extern "C" block).
 *
 * Conventions: every function returns 0 on success or a negative EU_ERR_* code and never throws
 * or aborts across the boundary; the caller owns all output buffers; scenes are immutable after
 * construction; a renderer is bound to one HIP device and must not be used from two threads at
 * once.  There is NO CPU fallback: without a usable HIP device the render entry points fail with
 * EU_ERR_NO_DEVICE.
 */
#ifndef EUCLIDER_AMD_H
#define EUCLIDER_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)      /* the library is built with -fvisibility=hidden: only what this header declares is exported */
#endif

#define EU_OK 0
#define EU_ERR_INVALID_ARGUMENT (-1)
#define EU_ERR_PARSE (-2)          /* ParserError, scene.rs:524-552; message says which variant */
#define EU_ERR_NO_DEVICE (-3)
#define EU_ERR_HIP (-4)
#define EU_ERR_CAPACITY (-5)       /* scene exceeds a compiled-in kernel capacity */
#define EU_ERR_TEXTURE (-6)
#define EU_ERR_UNIMPLEMENTED (-7)  /* the reference reaches `unimplemented!()` here (d4/entity/camera.rs:234) */
#define EU_ERR_PATH_STEPS (-8)     /* eu_trace_path: more than 4096 surface crossings in one call */
#define EU_ERR_BUSY (-9)           /* eu_sequence_submit: every slot is in flight */

typedef struct eu_scene eu_scene;        /* a parsed + flattened Universe3 / Universe4 */
typedef struct eu_renderer eu_renderer;  /* a scene resident in one GPU's HBM + work buffers */
typedef struct eu_sequence eu_sequence;  /* frames in flight: trace of frame k+1 overlaps the read-back of frame k */

/* Camera pose.  The reference's cameras (d3/entity/camera.rs:31-52, d4/entity/camera.rs:34-58)
 * only expose `new` / `new_with_location` to the loader; orientation, fov and max_depth are
 * fixed there (forward +x, up +z, left +y, fov 90, max_depth 10) and mutable only through
 * interactive input, so the caller passes the pose explicitly. */
typedef struct {
    int32_t dim;                 /* 3 or 4, must equal the scene's */
    uint32_t fov_deg;            /* diagonal field of view in degrees (u8 in the reference) */
    uint32_t max_depth;          /* Camera::max_depth(), universe/mod.rs:290,310 */
    uint32_t kind;               /* EU_CAMERA_*: which Camera::update applies (only eu_camera_update reads it) */
    double location[4], forward[4], up[4], left[4];
} eu_camera;

#define EU_CAMERA_PITCH_YAW_3 0u   /* PitchYawCamera3, d3/entity/camera.rs:76-245 */
#define EU_CAMERA_FREE_3 1u        /* FreeCamera3,     d3/entity/camera.rs:283-451 */
#define EU_CAMERA_FREE_4 2u        /* FreeCamera4,     d4/entity/camera.rs:34-241 (implied by dim == 4) */

/* One frame = Environment::render's arguments (universe/mod.rs:300-311), plus the multi-GPU partition.
 * Partition: with strip_count <= 1 the renderer traces rows [row_begin,row_end) into a buffer of
 * (row_end-row_begin) rows.  With strip_count = N > 1 the rows [row_begin,row_end) are cut into 8-row
 * strips and this renderer traces the strips s with s % N == strip_index (row tiles dealt round-robin,
 * so that glass objects, which cost hundreds of rays per pixel, are spread over all GPUs); its buffer
 * holds eu_frame_local_rows() rows: local row (s / N) * 8 + (row % 8). */
typedef struct {
    uint32_t width, height;      /* buffer dimensions (window / context.resolution) */
    uint32_t row_begin, row_end;
    uint64_t time_ms;            /* (time * 1000).as_secs(), d3/entity/surface.rs:32 */
    int32_t debug_crosshair;     /* context.debugging, universe/mod.rs:321-333 */
    uint32_t strip_count;        /* 0 or 1: contiguous rows */
    uint32_t strip_index;
    uint32_t reserved;
} eu_frame;

typedef struct {
    uint64_t rays;               /* Universe::trace calls with depth > 0 (reach trace_closest) */
    uint64_t bg_samples;         /* background().get_color calls */
    uint64_t nan_pixels;         /* float->u8 casts of NaN (the reference would panic) */
    uint64_t errors;             /* other would-panic conditions */
} eu_stats;

typedef struct {
    int32_t dim;
    uint32_t n_entities, n_shape_ops, n_leaves, n_materials, n_surfaces, n_color_ops, n_textures;
    uint32_t hit_cap, list_depth, flat_bytes;
} eu_scene_info;

/* Texture decoding is not part of the hot path (image::open, scene.rs:1053,1065); the host
 * supplies decoded RGBA8 texels (image::DynamicImage::get_pixel semantics: alpha 255 for RGB and
 * Luma images, row 0 = top).  The callback allocates *rgba with eu_alloc(); the library frees it.
 * Return 0 on success. */
typedef int (*eu_texture_loader)(void *user, const char *path, uint32_t *width, uint32_t *height, uint8_t **rgba);

typedef struct {
    eu_texture_loader load_texture;
    void *user;
    uint32_t random_seed;        /* stands in for rand::random() in surface_color_perlin_hue_random_3 */
    uint32_t reserved;
} eu_load_opts;

void *eu_alloc(size_t bytes);
void eu_free(void *p);

/* scene.rs:1466 Parser::default().parse::<Box<Environment>>(json) */
int eu_scene_from_json(const char *json, size_t len, const eu_load_opts *opts, eu_scene **out, char *err, size_t errlen);
void eu_scene_free(eu_scene *);
int eu_scene_get_info(const eu_scene *, eu_scene_info *);
int eu_scene_default_camera(const eu_scene *, eu_camera *out);      /* camera as loaded from the JSON */
const void *eu_scene_flat(const eu_scene *, size_t *bytes);         /* flattened blob (flat_scene.h), for inspection */

uint32_t eu_frame_local_rows(const eu_frame *);   /* rows of the output buffer for this frame/partition */

int eu_device_count(void);

/* How a renderer works (every field 0 = the default).  Pass it to eu_renderer_create_opts; eu_renderer_create uses the defaults.
 * (Rounds 1-2 read these from environment variables; a library built with -DEU_DIAGNOSTICS still honours EU_KERNEL,
 * EU_SPECIALIZE, EU_WF_STREAMS, EU_WF_RAY_FACTOR and EU_WF_BAND_PIXELS on top of the struct, for the profiling scripts.) */
#define EU_KERNEL_AUTO 0u          /* the wavefront pipeline; the stack kernel only to re-trace a frame that overflowed the ray queues */
#define EU_KERNEL_WAVEFRONT 1u
#define EU_KERNEL_STACK 2u         /* the persistent stack-based kernel for everything (several times slower; A/B checks) */
#define EU_SPECIALIZE_AUTO 0u      /* currently = OFF */
#define EU_SPECIALIZE_OFF 1u       /* the ahead-of-time kernels, which interpret the flat scene */
#define EU_SPECIALIZE_SYNC 2u      /* eu_renderer_create compiles trace kernels specialised for THIS scene with hiprtc (a few seconds the
                                      first time; the code object is cached on disk by content) and every frame uses them; if the compilation
                                      fails the renderer falls back to the interpreter kernels and eu_renderer_jit_info says so */
#define EU_SPECIALIZE_ASYNC 3u     /* the same without the wait: a worker thread of the library compiles while the renderer traces with the
                                      interpreter kernels, and switches to the specialised ones at the first frame launched after the code object
                                      is ready (eu_renderer_jit_info.active turns 1).  The frames are the same bit for bit either way.  A code
                                      object already in a cache is used from the first frame on. */
#define EU_RENDERER_SHADE_SCENE_GLOBAL 1u   /* flags: the interpreter's shade kernel reads the scene from global memory, not from its LDS copy
                                               (what scenes above ~40 KB get anyway; here so that tests can reach that variant) */
#define EU_RENDERER_NO_FUSE 2u               /* flags: one intersect and one shade launch per generation (rounds 1-3) instead of the fused kernel that shades a
                                               generation and intersects the rays it has just queued (A/B checks; the frames are the same) */
typedef struct {
    uint32_t struct_size;        /* sizeof(eu_renderer_opts) of the caller: the struct may grow */
    uint32_t kernel;             /* EU_KERNEL_* */
    uint32_t specialize;         /* EU_SPECIALIZE_* */
    uint32_t streams;            /* band pipelines in flight per frame, 1..8 (0: the library's choice -- three for a scene whose recursion can branch, else one;
                                    after the first frames it goes by their rays per pixel: three above 2.5, one below 1.6) */
    double ray_factor;           /* ray-queue slots per pixel and generation (0: 4.0); a frame that needs more is reported through
                                    EU_ERR_CAPACITY (asynchronous calls) or traced again by the stack kernel (eu_render, eu_render_multi) */
    uint64_t band_pixels;        /* pixels per wavefront pass (0: 4 Mi): larger frames are traced in bands of whole 8-row tiles */
    const char *cache_dir;       /* where specialised code objects are kept (NULL: $XDG_CACHE_HOME/euclider_amd or ~/.cache/euclider_amd);
                                    the directory jit_cache next to the library is consulted first (read-only) */
    uint32_t flags;              /* EU_RENDERER_* */
    uint32_t reserved;
    const char *jit_flags;       /* extra compiler flags for the specialised kernels, space-separated (tuning experiments, e.g.
                                    "-DEU_SHADE_WAVES=4"); part of the cache key; NULL: none */
    uint32_t band_grid_permille; /* the share of a full-chip grid (in 1/1000) every band pipeline of a frame launches (0: chosen from `streams`) */
    uint32_t split_pixels;       /* frames of at least this many pixels are cut into `streams` concurrent bands (0: 524288) */
} eu_renderer_opts;

typedef struct {
    int32_t requested;           /* the renderer was asked to specialise */
    int32_t active;              /* its frames run on the specialised kernels */
    int32_t from_cache;          /* the code object came from the cache (memory or disk), not from a compilation */
    uint32_t hit_stack_entries;  /* per-ray hit-stack entries the specialised intersect kernel was compiled for */
    double compile_ms;           /* time spent in hiprtc (0 when cached) */
    char key[40];                /* content hash the code object is cached under */
} eu_jit_info;

int eu_renderer_create(const eu_scene *, int device, eu_renderer **out, char *err, size_t errlen);
int eu_renderer_create_opts(const eu_scene *, int device, const eu_renderer_opts *opts, eu_renderer **out, char *err, size_t errlen);
void eu_renderer_destroy(eu_renderer *);
int eu_renderer_jit_info(eu_renderer *, eu_jit_info *out);
/* What the most recent failing call on this renderer had to say (valid until the next call on it; "" if nothing failed yet). */
const char *eu_renderer_error(const eu_renderer *);
/* Why a renderer that was asked to specialise runs the interpreter kernels (eu_jit_info.active == 0): the compiler's log, or what the
 * runtime said about the code object; "" when the specialised kernels are in use.  (No scene is too large: beyond 256 shape operations
 * or 48 surfaces the rest of a scene is traced from the flat scene inside the specialised kernels.) */
const char *eu_renderer_jit_log(const eu_renderer *);
/* The HIP source eu_renderer_create_opts(EU_SPECIALIZE_SYNC) would compile for this scene (no GPU needed): *source is allocated
 * with eu_alloc, NUL-terminated; free it with eu_free.  key (optional, >= 40 bytes): the cache key. */
int eu_scene_jit_source(const eu_scene *, char **source, char *key);
/* The same for a renderer created with eu_renderer_opts.jit_flags / .flags (EU_RENDERER_NO_FUSE selects the two-kernel set): both are
 * part of the cache key. */
int eu_scene_jit_source_opts(const eu_scene *, const char *jit_flags, unsigned renderer_flags, char **source, char *key);
/* Compiles that source for gfx950 into the cache directory without touching a GPU (what a build step runs so that the first
 * renderer does not wait); err receives the compiler's log on failure (on success: its warnings and the generator's notes, or ""). */
int eu_scene_jit_precompile(const eu_scene *, const char *cache_dir, eu_jit_info *info, char *err, size_t errlen);
int eu_scene_jit_precompile_opts(const eu_scene *, const char *cache_dir, const char *jit_flags, unsigned renderer_flags, eu_jit_info *info, char *err, size_t errlen);

/* Environment::render on the GPU, asynchronous on `hip_stream` (a hipStream_t; NULL = default stream).
 * rgba_dev: DEVICE buffer of eu_frame_local_rows()*width uint32 (R | G<<8 | B<<16 | 255<<24), rows in
 * the reference's order (row 0 first = bottom row on screen).  hit_t_dev: optional DEVICE buffer of
 * eu_frame_local_rows()*width doubles, distance of the primary ray's closest hit (-1 if none). */
int eu_render_device(eu_renderer *, const eu_camera *, const eu_frame *, void *hip_stream, void *rgba_dev, double *hit_t_dev);
/* RGBA8 -> packed RGB8 (RawImage2d<u8>, ClientFormat::U8U8U8, universe/mod.rs:351-356), device to device. */
int eu_pack_rgb_device(eu_renderer *, const void *rgba_dev, void *rgb_dev, size_t pixels, void *hip_stream);
/* Counters of the most recent eu_render_device on this renderer (waits for it to finish). */
int eu_renderer_stats(eu_renderer *, eu_stats *);
/* Duration of the most recent trace-kernel launch, measured with HIP events on its own stream. */
int eu_renderer_kernel_ms(eu_renderer *, float *ms);
/* The same for the most recent min(max_n, 64) launches, oldest first; returns how many were written. */
int eu_renderer_kernel_ms_history(eu_renderer *, float *ms, int max_n);
/* Number of frames (or single pixels) eu_render / eu_trace_screen_point / eu_sequence_next / eu_render_multi had to trace a second
 * time since the renderer was created: a frame whose recursion fans out beyond the ray queues (more than eu_renderer_opts.ray_factor
 * rays per pixel in one generation), or one of whose rays needed more hit-stack entries than the wavefront kernels reserve, is
 * traced again by the stack-based kernel, several times slower.  The result is the same either way; this counter is how a caller
 * notices the slow path.  Only the plainly asynchronous eu_render_device cannot retry: eu_renderer_stats then returns
 * EU_ERR_CAPACITY and eu_renderer_error says which of the two it was. */
int eu_renderer_retraces(eu_renderer *, uint64_t *count);

/* Diagnostic builds (-DEU_PROFILE_PHASES) only: summed per-wave cycle shares of the kernel's phases
 * (refill, intersect, shade, return); all zero in the shipped build. */
int eu_renderer_debug_phases(eu_renderer *, unsigned long long out[16]);
/* Diagnostics (kernels built with -DEU_PROFILE_WG among jit_flags): per-workgroup time stamps of every launch.  First call (out NULL):
 * recording on; later calls copy the records (4 words each) written since the previous call. */
int eu_renderer_debug_wg_profile(eu_renderer *, unsigned long long *out, size_t max_records, size_t *n_records);
/* Diagnostics: rays queued per generation by the most recent frame's first band (the last band pipeline of buffer set 0). */
int eu_renderer_debug_generations(eu_renderer *, unsigned long long out[17]);

/* Synchronous convenience = Environment::render: traces the frame and copies RGB8 (and hit_t) to the host.  If the frame's
 * recursion fans out beyond the wavefront queues (reported as EU_ERR_CAPACITY by eu_renderer_stats after the asynchronous
 * eu_render_device), this call traces it again with the stack-based persistent kernel, which has no such limit. */
int eu_render(eu_renderer *, const eu_camera *, const eu_frame *, uint8_t *rgb_host, double *hit_t_host, eu_stats *);
/* Environment::trace_screen_point (universe/mod.rs:371-397): one pixel, un-quantised Rgb<F> (traced again by the stack-based kernel should
 * the wavefront pipeline have dropped one of its rays). */
int eu_trace_screen_point(eu_renderer *, const eu_camera *, const eu_frame *, int32_t x, int32_t y, double rgb[3]);

/* ---- frame sequences ("next" row f4 of the scope table) ---------------------------------------------
 * The frame loop around Environment::render (simulation.rs:93-150): the reference finishes a frame, uploads it and only
 * then starts the next.  A sequence keeps up to `slots` frames in flight on its own streams: eu_sequence_submit queues
 * trace + RGB8 pack + an asynchronous copy into pinned host memory and returns at once; eu_sequence_next waits for the
 * OLDEST submitted frame (tracing it again with the stack-based kernel, synchronously, should the wavefront pipeline have dropped
 * rays: eu_renderer_retraces) and hands out its image (rows of the reference's RawImage2d; the sequence owns slots + 1
 * pinned images, so the one handed out stays valid until the NEXT eu_sequence_next, however many frames are submitted in
 * between) and its counters.  Frames may differ in size (the run-time `resolution` divisor, simulation.rs:284-306), time
 * (time-varying surfaces) and camera.  Every slot traces on its own stream with its own work buffers and counters (the scene
 * and its textures are resident once per slot), so consecutive frames overlap on the GPU.  Do not interleave eu_render* calls
 * on the same renderer while frames are in flight. */
int eu_sequence_create(eu_renderer *, uint32_t max_width, uint32_t max_height, uint32_t slots, eu_sequence **out);
void eu_sequence_destroy(eu_sequence *);
int eu_sequence_submit(eu_sequence *, const eu_camera *, const eu_frame *);
int eu_sequence_next(eu_sequence *, const uint8_t **rgb_host, uint32_t *width, uint32_t *rows, eu_stats *);

/* ---- one frame across the GPUs of a node (scope row e, BASELINE config 5) ------------------------------
 * the frame's rows [row_begin,row_end) are cut into 8-row strips dealt round-robin over `devices`
 * (the partition eu_frame.strip_* describes); every device traces and packs its strips, the packed strips travel to
 * devices[0] over xGMI (one peer transfer per device per frame, landing in place), and a kernel there restores row order =
 * the reference's RawImage2d (universe/mod.rs:351-356).  One process drives all devices: this is what a Rust
 * `impl Environment` (simulation.rs:86 holds a single Box<Environment>) calls to reach 8 GPUs.  The same device may be listed
 * more than once (then its renderers share it).  `f->strip_count` must be 0: the partition is this call's business.
 * rgb_host (optional): rows*width*3 bytes, frame order.  rgb_dev_root (optional): receives the DEVICE address (on
 * devices[0]) of the same image, valid until the next call.  stats: summed over the devices. */
typedef struct eu_multi eu_multi;
int eu_multi_create(const eu_scene *, const int *devices, int n_devices, eu_multi **out, char *err, size_t errlen);
int eu_multi_create_opts(const eu_scene *, const int *devices, int n_devices, const eu_renderer_opts *opts, eu_multi **out, char *err, size_t errlen);
void eu_multi_destroy(eu_multi *);
int eu_render_multi(eu_multi *, const eu_camera *, const eu_frame *, uint8_t *rgb_host, void **rgb_dev_root, eu_stats *stats);
/* The same in two halves, for a frame loop (simulation.rs:93-150): eu_render_multi_begin queues the frame on all devices and
 * returns; eu_render_multi_end waits for the OLDEST frame begun, retraces overflowed strips, and hands out image and stats as
 * above.  Two frames may be in flight (a third begin is EU_ERR_INVALID_ARGUMENT): the strip buffers, the root's gather and
 * output buffers and the per-device counters exist twice, and the pack + peer transfer + row restore of frame k run on
 * separate copy streams while frame k + 1 is traced.  rgb_dev_root stays valid until the second begin after its end.
 * eu_render_multi == begin + end, and refuses while frames begun this way are in flight. */
int eu_render_multi_begin(eu_multi *, const eu_camera *, const eu_frame *);
int eu_render_multi_end(eu_multi *, uint8_t *rgb_host, void **rgb_dev_root, eu_stats *stats);
const char *eu_multi_error(const eu_multi *);

/* ---- camera motion ("next" row f3 of the scope table) ------------------------------------------------
 * Universe::trace_path_unknown (universe/mod.rs:273-286): push a point `distance` along `direction`
 * through surfaces and materials (portals rescale the step; get_path surface.rs:164-197).  Runs on the
 * GPU against the resident scene.  *found = 1: Some((out_location, out_direction)); 0: None (no
 * material at `location`).  Vectors have 4 slots, the first `dim` are used. */
int eu_trace_path(eu_renderer *, const double location[4], const double direction[4], double distance,
                  double out_location[4], double out_direction[4], int32_t *found);

/* SimulationContext as Camera::update sees it (simulation.rs:29-41): pressed keys, mouse delta, and the
 * frame's delta time.  mouse_sensitivity / speed are the Camera3Data / FreeCamera4 fields the loader
 * cannot set (0.01 and 10.0, d3/entity/camera.rs:46-47); 0 selects those defaults. */
#define EU_KEY_W 0x0001u
#define EU_KEY_S 0x0002u
#define EU_KEY_A 0x0004u
#define EU_KEY_D 0x0008u
#define EU_KEY_LSHIFT 0x0010u
#define EU_KEY_LCONTROL 0x0020u
#define EU_KEY_Q 0x0040u
#define EU_KEY_E 0x0080u
#define EU_KEY_C 0x0100u
#define EU_KEY_M 0x0200u
#define EU_KEY_I 0x0400u
#define EU_KEY_O 0x0800u
#define EU_KEY_K 0x1000u
#define EU_KEY_L 0x2000u
typedef struct {
    uint32_t keys;               /* EU_KEY_* bit set = context.pressed_keys() */
    int32_t delta_mouse_x, delta_mouse_y;   /* context.delta_mouse */
    uint32_t reserved;
    uint64_t delta_time_ms;      /* (delta_time * 1000).as_secs() */
    double mouse_sensitivity, speed;
} eu_input;

/* Camera::update = Environment::update (universe/mod.rs:359,399-405; d3/entity/camera.rs:191-245,396-451;
 * d4/entity/camera.rs:182-241): rotation from mouse/keys on the host, translation through eu_trace_path.
 * The renderer may be NULL when the input cannot move the camera (delta_time_ms == 0 or no movement key);
 * otherwise a NULL renderer fails with EU_ERR_NO_DEVICE.  EU_ERR_UNIMPLEMENTED: a 4-D camera crossed a
 * surface that turned its direction -- the reference panics there; the pose is left as it was. */
int eu_camera_update(eu_renderer *, eu_camera *, const eu_input *);

/* Device self-test of the elementary functions (fn: 0 acos 1 asin 2 sin 3 cos 4 tan 5 atan2(x,y) 6 sqrt
 * 7 x/y 8 fmod(x,y)); host buffers. */
int eu_selftest_math(int device, int fn, const double *x, const double *y, double *out, size_t n);

const char *eu_version(void);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
