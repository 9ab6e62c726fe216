/*
 * renderer.hip -- the trace kernels for gfx950 and the host-side renderer behind the C ABI.
 *
 * Kernel shape (replaces Environment::render's "one thread-pool job per pixel",
 * /root/reference/src/universe/mod.rs:300-357):
 *   - persistent wavefronts: the grid is sized to the chip (CUs x resident blocks), not to the
 *     image; each lane owns one primary ray at a time and, when its pixel is finished, pulls the
 *     next pixel index from a global counter with ONE wave-aggregated atomic (ballot + mbcnt).
 *     A lane's step is "trace one ray segment", so lanes that sit at different recursion depths
 *     of different pixels still execute the same code: secondary-ray divergence (1 ray for a wall
 *     pixel, hundreds for a glass pixel) costs idle lanes only at the very end of the frame.
 *   - pixels are handed out in 8x8 tiles so the 64 lanes of a wave start on neighbouring pixels.
 *   - the flattened scene (a few KB) is staged once per workgroup into LDS; every lane reads the
 *     same record at the same time (broadcast, conflict-free).
 *   - each pixel is written once as RGBA8 (one dword store); textures are read-only RGBA8 in HBM.
 * No MFMA: there is no dense contraction anywhere in this path; it is f64 VALU + divergence bound.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/euclider_amd.h"
#include "camera_host.hpp"      /* Camera::update stays f64 in every build: before eu_real.h */
#include "scene_host.hpp"       /* ends with eu_real.h: from here on `double` is the path's F */
#include "trace_device.h"
#include "trace_megakernel.h"
#include "trace_wavefront.h"
#include "trace_stream.h"
#include "trace_path.h"

/* RGBA8 -> packed RGB8 (RawImage2d U8U8U8, universe/mod.rs:351-356): 4 pixels (16 B in, 12 B out) per thread */
__global__ void eu_pack_rgb_kernel(const uint32_t *__restrict__ rgba, uint8_t *__restrict__ rgb, size_t pixels) {
    size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t nq = pixels / 4;
    if (q < nq) {
        const uint4 v = ((const uint4 *)rgba)[q];
        uint32_t w0 = (v.x & 0xffffffu) | (v.y << 24);
        uint32_t w1 = ((v.y >> 8) & 0xffffu) | (v.z << 16);
        uint32_t w2 = ((v.z >> 16) & 0xffu) | (v.w << 8);
        uint32_t *out = (uint32_t *)(rgb + q * 12);
        out[0] = w0; out[1] = w1; out[2] = w2;
    }
    if (q == 0) {
        for (size_t p = nq * 4; p < pixels; p++) { uint32_t v = rgba[p]; rgb[p * 3] = (uint8_t)v; rgb[p * 3 + 1] = (uint8_t)(v >> 8); rgb[p * 3 + 2] = (uint8_t)(v >> 16); }
    }
}

__global__ void eu_math_kernel(int fn, const eu_f64 *x, const eu_f64 *y, eu_f64 *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    eu_f64 a = x[i], b = y ? y[i] : 0.0, r;
    switch (fn) {
    case 0: r = eu_acos_f64(a); break;
    case 1: r = eu_asin_f64(a); break;
    case 2: r = eu_sin_f64(a); break;
    case 3: r = eu_cos_f64(a); break;
    case 4: r = eu_tan_f64(a); break;
    case 5: r = eu_atan2_f64(a, b); break;
    case 6: r = sqrt((eu_f64)a); break;
    case 7: r = a / b; break;
    case 8: r = fmod((eu_f64)a, (eu_f64)b); break;
    default: r = 0.0;
    }
    out[i] = r;
}

/* ------------------------------------------------------------------ host side */
struct eu_renderer {
    std::shared_ptr<const euclider::FlatScene> flat;     /* host copy of the scene: frame sequences clone the renderer per slot */
    int device = 0;
    int dim = 3;
    uint32_t hit_cap = 0;
    uint32_t color_depth = 0;
    uint32_t scene_words = 0;
    uint64_t *d_scene = nullptr;
    std::vector<void *> d_textures;
    EuDevCounters *d_counters = nullptr;
    uint32_t *d_rgba = nullptr; size_t rgba_pixels = 0;      /* internal frame buffers for eu_render */
    uint8_t *d_rgb = nullptr;
    eu_f64 *d_hit = nullptr;
    eu_f64 *d_point = nullptr;
    double *d_path_in = nullptr;            /* eu_trace_path: location, direction, distance */
    EuPathResult *d_path_out = nullptr;
    static constexpr int EV_RING = 64;        /* per-launch HIP event pairs, on the launch stream */
    hipEvent_t ev_start[EV_RING] = {}, ev_stop[EV_RING] = {};
    unsigned long long launches = 0;
    hipStream_t last_stream = nullptr;
    bool have_timing = false;
    int num_cus = 0;
    bool scene_in_lds = true;
    /* which kernels trace a frame: the generation-synchronous wavefront pipeline (trace_wavefront.h) unless EU_KERNEL says
     * otherwise ("stream": the persistent one-launch kernel of trace_stream.h, "mega": the stack-based kernel) */
    enum { PATH_STREAM = 0, PATH_WAVEFRONT = 1, PATH_MEGA = 2 };
    int path = PATH_WAVEFRONT;
    bool use_wavefront = true;               /* among the two older paths */
    /* stream kernel: per-workgroup ray chunks + the shared node pool, grown on demand */
    EuTsPool ts = {};
    std::vector<void *> ts_allocs;
    unsigned ts_grid_cap = 0;                /* workgroups the ray chunks / hit rows / counter rows are sized for */
    uint32_t ts_nch = 0;
    size_t ts_node_chunks = 0;
    unsigned ts_grid_last = 0;               /* grid of the most recent launch: that many counter rows are valid */
    bool ts_last = false;                    /* the most recent frame went through the stream kernel */
    uint64_t retraces = 0;                   /* frames eu_render traced a second time (queue / node-pool overflow): eu_renderer_retraces */
    double ts_node_factor = R(6.0);             /* node slots per pixel (EU_TS_NODE_FACTOR); eu_render doubles it after an overflow */
    unsigned ts_grid_limit = 0;              /* EU_TS_GRID: fewer workgroups than the chip holds (diagnostics) */
    /* wavefront pipeline buffers (HBM), sized for the largest frame seen so far */
    static constexpr int WF_MAX_STREAMS = 4;
    EuWfBuffers wf[WF_MAX_STREAMS] = {};     /* band pipelines run concurrently on side streams */
    size_t wf_pixels = 0;
    uint32_t wf_depth = 0;                   /* deepest max_depth the node slots are sized for */
    int wf_sets = 0;                         /* buffer sets allocated */
    bool prepare_only = false;               /* render_device_impl: size the work buffers for the frame, launch nothing */
    /* FINISH step: once a generation holds fewer than wf_finish_rays rays, the stream kernel (trace_stream.h, import mode) takes
     * that generation's queue over and finishes those rays and all their descendants in ONE launch; the generation to hand over at
     * is learnt from the previous frame's queue lengths (read back asynchronously), so a wrong guess costs time, never correctness */
    EuTsPool wf_fin[WF_MAX_STREAMS] = {};
    unsigned wf_fin_grid = 0;
    uint32_t wf_finish_rays = 0;             /* EU_WF_FINISH_RAYS; 0 = never hand over, the default: measured slower */
    uint32_t wf_handover[WF_MAX_STREAMS] = {};      /* generation the next frame hands over at (>= max_depth: no finish step) */
    EuDevCounters *wf_fin_counters[WF_MAX_STREAMS] = {};   /* the finish kernel's own segment / node-chunk counters */
    uint32_t *wf_d_totals[WF_MAX_STREAMS] = {};     /* rays per generation of the band most recently traced with this buffer set */
    uint32_t *wf_h_totals[WF_MAX_STREAMS] = {};     /* pinned host copy */
    hipEvent_t wf_totals_ready[WF_MAX_STREAMS] = {};
    bool wf_totals_pending[WF_MAX_STREAMS] = {};
    uint32_t wf_totals_handover[WF_MAX_STREAMS] = {};   /* the hand-over generation of the frame the pending totals describe */
    size_t wf_totals_pixels[WF_MAX_STREAMS] = {};
    hipStream_t wf_stream[WF_MAX_STREAMS] = {};
    hipEvent_t wf_fork = nullptr, wf_join[WF_MAX_STREAMS] = {};
    int wf_n_streams = 2;                    /* EU_WF_STREAMS (1 = everything on the caller's stream) */
    double wf_ray_factor = R(4.0);
    uint64_t wf_band_pixels = 4u << 20;      /* pixels traced per wavefront pass (EU_WF_BAND_PIXELS) */
    /* diagnostic switches, read from the environment once, when the renderer is created */
    uint32_t dbg_hs_cap = 0;                 /* EU_HS_CAP */
    bool dbg_hs_private = false;             /* EU_HS_PRIVATE */
    bool dbg_shade_scene_global = false;     /* EU_SHADE_SCENE_GLOBAL */
    uint32_t dbg_skip_entities = 0;          /* EU_DEBUG_SKIP_ENTITIES (-DEU_PROFILE_ISECT / -DEU_DEBUG_SKIP builds) */
    uint32_t dbg_skip_shade = 0;             /* EU_DEBUG_SKIP_SHADE (-DEU_DEBUG_SKIP builds) */
    std::vector<void *> wf_allocs;
    std::string err;
};

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            r->err = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            return EU_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

static void set_err(char *err, size_t errlen, const std::string &msg) {
    if (err && errlen) { snprintf(err, errlen, "%s", msg.c_str()); }
}

extern "C" int eu_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static int renderer_create_impl(std::shared_ptr<const euclider::FlatScene> flat_scene, int device, eu_renderer **out, char *err, size_t errlen) {
    struct { const euclider::FlatScene &flat; } scene_ref{*flat_scene}, *scene = &scene_ref;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        set_err(err, errlen, "no usable HIP device (this library has no CPU fallback)");
        return EU_ERR_NO_DEVICE;
    }
    eu_renderer *r = new eu_renderer();
    r->flat = flat_scene;
    r->device = device;
    const EuFlatHeader &h = scene->flat.header();
    r->dim = (int)h.dim;
    r->hit_cap = h.hit_cap;
    r->color_depth = h.color_depth;
    auto failhip = [&](int code) { set_err(err, errlen, r->err); eu_renderer_destroy(r); return code; };
    if (h.hit_cap > 96) { r->err = "scene needs a per-ray hit stack of " + std::to_string(h.hit_cap) + " entries (compiled maximum 96)"; return failhip(EU_ERR_CAPACITY); }
    if (h.list_depth > 8 || h.color_depth > 4 || h.rpn_depth > 8) { r->err = "scene exceeds a compiled stack depth (csg lists 8, colour 4, rpn 8)"; return failhip(EU_ERR_CAPACITY); }
    auto body = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        r->num_cus = prop.multiProcessorCount;
        std::vector<uint64_t> blob = scene->flat.words;
        EuFlatHeader *bh = reinterpret_cast<EuFlatHeader *>(blob.data());
        for (uint32_t m = 0; m < bh->n_mapped; m++) {
            const auto &tex = scene->flat.textures[m];
            void *dptr = nullptr;
            size_t bytes = (size_t)tex->w * tex->h * 4;
            HIP_TRY(hipMalloc(&dptr, bytes));
            r->d_textures.push_back(dptr);
            HIP_TRY(hipMemcpy(dptr, tex->rgba->data(), bytes, hipMemcpyHostToDevice));
            EuFlatMapped *fm = reinterpret_cast<EuFlatMapped *>(blob.data() + bh->off_mapped + 8 * m);
            fm->texels = (uint64_t)(uintptr_t)dptr;
        }
        r->scene_words = (uint32_t)blob.size();
        r->scene_in_lds = blob.size() * 8 <= 60 * 1024;
        if (const char *k = getenv("EU_KERNEL")) {
            const std::string ks(k);
            r->path = ks == "mega" ? eu_renderer::PATH_MEGA : (ks == "stream" ? eu_renderer::PATH_STREAM : eu_renderer::PATH_WAVEFRONT);
            r->use_wavefront = ks != "mega";
        }
        if (const char *k = getenv("EU_TS_NODE_FACTOR")) r->ts_node_factor = atof(k);
        if (const char *k = getenv("EU_TS_GRID")) r->ts_grid_limit = (unsigned)atoi(k);
        if (const char *k = getenv("EU_WF_FINISH_RAYS")) r->wf_finish_rays = (uint32_t)strtoul(k, nullptr, 10);
        if (const char *k = getenv("EU_WF_RAY_FACTOR")) r->wf_ray_factor = atof(k);
        if (const char *k = getenv("EU_WF_BAND_PIXELS")) r->wf_band_pixels = strtoull(k, nullptr, 10);
        r->wf_n_streams = (h.flags & 1u) ? 2 : 1;      /* branching scenes: two band pipelines fill each other's kernel tails (measured: +8 %); others: -5 % */
        if (const char *k = getenv("EU_HS_CAP")) r->dbg_hs_cap = (uint32_t)atoi(k);
        r->dbg_hs_private = getenv("EU_HS_PRIVATE") != nullptr;
        r->dbg_shade_scene_global = getenv("EU_SHADE_SCENE_GLOBAL") != nullptr;
#if defined(EU_DEBUG_SKIP) || defined(EU_PROFILE_ISECT)      /* only the diagnostic builds' kernels mask these bits back out */
        if (const char *k = getenv("EU_DEBUG_SKIP_ENTITIES")) r->dbg_skip_entities = (uint32_t)strtoul(k, nullptr, 0) << 24;
#endif
#if defined(EU_DEBUG_SKIP)
        if (const char *k = getenv("EU_DEBUG_SKIP_SHADE")) r->dbg_skip_shade = (uint32_t)strtoul(k, nullptr, 0) << 16;
#endif
        if (const char *k = getenv("EU_WF_STREAMS")) { int v = atoi(k); r->wf_n_streams = v < 1 ? 1 : (v > eu_renderer::WF_MAX_STREAMS ? eu_renderer::WF_MAX_STREAMS : v); }
        HIP_TRY(hipMalloc((void **)&r->d_scene, blob.size() * 8));
        HIP_TRY(hipMemcpy(r->d_scene, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&r->d_counters, sizeof(EuDevCounters)));
        HIP_TRY(hipMemset(r->d_counters, 0, sizeof(EuDevCounters)));
        HIP_TRY(hipDeviceSynchronize());      /* (hipMemset is asynchronous to the host; see wf_ensure) */
        HIP_TRY(hipMalloc((void **)&r->d_point, 3 * sizeof(eu_f64)));
        for (int i = 0; i < eu_renderer::EV_RING; i++) { HIP_TRY(hipEventCreate(&r->ev_start[i])); HIP_TRY(hipEventCreate(&r->ev_stop[i])); }
        return EU_OK;
    };
    int rc = body();
    if (rc != EU_OK) return failhip(rc);
    *out = r;
    return EU_OK;
}

extern "C" int eu_renderer_create(const eu_scene *scene, int device, eu_renderer **out, char *err, size_t errlen) {
    if (!scene || !out) return EU_ERR_INVALID_ARGUMENT;
    return renderer_create_impl(std::make_shared<const euclider::FlatScene>(scene->flat), device, out, err, errlen);
}

extern "C" void eu_renderer_destroy(eu_renderer *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    for (void *p : r->d_textures) (void)hipFree(p);
    for (void *p : r->wf_allocs) (void)hipFree(p);
    for (void *p : r->ts_allocs) (void)hipFree(p);
    for (int k = 0; k < eu_renderer::WF_MAX_STREAMS; k++) {
        if (r->wf_h_totals[k]) (void)hipHostFree(r->wf_h_totals[k]);
        if (r->wf_totals_ready[k]) (void)hipEventDestroy(r->wf_totals_ready[k]);
    }
    for (int k = 0; k < eu_renderer::WF_MAX_STREAMS; k++) { if (r->wf_stream[k]) (void)hipStreamDestroy(r->wf_stream[k]); if (r->wf_join[k]) (void)hipEventDestroy(r->wf_join[k]); }
    if (r->wf_fork) (void)hipEventDestroy(r->wf_fork);
    if (r->d_scene) (void)hipFree(r->d_scene);
    if (r->d_counters) (void)hipFree(r->d_counters);
    if (r->d_rgba) (void)hipFree(r->d_rgba);
    if (r->d_rgb) (void)hipFree(r->d_rgb);
    if (r->d_hit) (void)hipFree(r->d_hit);
    if (r->d_point) (void)hipFree(r->d_point);
    if (r->d_path_in) (void)hipFree(r->d_path_in);
    if (r->d_path_out) (void)hipFree(r->d_path_out);
    for (int i = 0; i < eu_renderer::EV_RING; i++) { if (r->ev_start[i]) (void)hipEventDestroy(r->ev_start[i]); if (r->ev_stop[i]) (void)hipEventDestroy(r->ev_stop[i]); }
    delete r;
}

static int make_dev_camera(const eu_camera *cam, const eu_frame *f, EuDevCamera &dc) {
    const int D = cam->dim;
    memset(&dc, 0, sizeof dc);
    for (int i = 0; i < D; i++) { dc.location[i] = cam->location[i]; dc.forward[i] = cam->forward[i]; dc.up[i] = cam->up[i]; }
    if (D == 3) {   /* get_right = cross(forward, up).normalize(), d3/entity/camera.rs:62-64 */
        double cr[3];      /* from the pose already rounded to F: the reference's camera holds F values (with the ABI's f64 fields
                             * the low_precision build used to form these products in f64: one ulp off for a pose that is not f32-exact) */
        cr[0] = dc.forward[1] * dc.up[2] - dc.forward[2] * dc.up[1];
        cr[1] = dc.forward[2] * dc.up[0] - dc.forward[0] * dc.up[2];
        cr[2] = dc.forward[0] * dc.up[1] - dc.forward[1] * dc.up[0];
        double n = sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
        for (int i = 0; i < 3; i++) dc.right[i] = cr[i] / n;
    } else {        /* right = -left, d4/entity/camera.rs:167 */
        for (int i = 0; i < D; i++) dc.right[i] = -cam->left[i];
    }
    const double w = (double)f->width, h = (double)f->height;
    const double fov_rad = EU_PI_C * (double)cam->fov_deg / R(180.0);
    dc.dist = sqrt(w * w + h * h) / (R(2.0) * eu_tan(fov_rad / R(2.0)));
    dc.max_depth = cam->max_depth;
    return EU_OK;
}

template <int D, int HSCAP, bool LDS>
static hipError_t launch_trace(eu_renderer *r, hipStream_t stream, const EuDevCamera &dc, const EuDevFrame &df, uint32_t *rgba, eu_f64 *hit_t, eu_f64 *point) {
    auto kern = eu_trace_kernel<D, HSCAP, LDS>;
    const uint32_t hs_cap = HSCAP ? (uint32_t)HSCAP : (r->hit_cap < 8 ? 8u : ((r->hit_cap + 3u) & ~3u));
    size_t lds_bytes = LDS ? (size_t)r->scene_words * 8 : 0;
    if (HSCAP == 0) lds_bytes += (size_t)(EU_BLOCK / 64) * hs_cap * 64 * 12;
    int blocks_per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kern, EU_BLOCK, lds_bytes);
    if (e != hipSuccess) return e;
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    unsigned long long total_waves = ((unsigned long long)df.n_tiles);
    unsigned long long want_blocks = (total_waves + (EU_BLOCK / 64) - 1) / (EU_BLOCK / 64);
    unsigned long long grid = (unsigned long long)r->num_cus * (unsigned long long)blocks_per_cu;
    if (grid > want_blocks) grid = want_blocks;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(EU_BLOCK), lds_bytes, stream, r->d_scene, r->scene_words, hs_cap, dc, df, r->d_counters, rgba, hit_t, point);
    return hipGetLastError();
}


/* ------------------------------------------------------------------ wavefront pipeline (trace_wavefront.h) */
static int wf_ensure(eu_renderer *r, size_t pixels, uint32_t max_depth, int n_sets) {
    if (pixels <= r->wf_pixels && max_depth <= r->wf_depth && n_sets <= r->wf_sets) return EU_OK;
    if (pixels < r->wf_pixels) pixels = r->wf_pixels;
    if (max_depth > r->wf_depth) r->wf_depth = max_depth;
    if (n_sets > r->wf_sets) r->wf_sets = n_sets;      /* the second buffer set exists only once a frame is traced as two concurrent bands */
    for (void *p : r->wf_allocs) (void)hipFree(p);
    r->wf_allocs.clear();
    r->wf_pixels = 0;
    const int D = r->dim;
    for (int set = 0; set < r->wf_sets; set++) {
    EuWfBuffers &B = r->wf[set];
    memset(&B, 0, sizeof B);
    /* one queue segment per producer workgroup; a generation's queue holds n_seg * seg_cap ray slots */
    uint32_t n_seg = (uint32_t)r->num_cus * 3u;
    if (n_seg > EU_WF_MAX_SEG) n_seg = EU_WF_MAX_SEG;
    if (pixels / 16 < n_seg) n_seg = pixels / 16 < 16 ? 16u : (uint32_t)(pixels / 16);      /* tiny frames, single pixels: fewer producers, small buffers */
    size_t seg_cap = ((size_t)((double)pixels * r->wf_ray_factor) + n_seg - 1) / n_seg;
    if (seg_cap < 1024) seg_cap = 1024;        /* small frames: absorb uneven segments */
    seg_cap = (seg_cap + 255) & ~(size_t)255;
    const size_t ray_cap = seg_cap * n_seg;
    const size_t node_cap = ray_cap * (size_t)(r->wf_depth ? r->wf_depth : 1u);      /* one slot per ray of every generation the deepest frame so far has */
    if (ray_cap > 0x7ffffff0ull || node_cap > 0xfffffff0ull) { r->err = "frame too large for 32-bit queue indices; render it in row tiles"; return EU_ERR_CAPACITY; }
    static const bool poison = getenv("EU_DEBUG_POISON") != nullptr;      /* diagnostics: no buffer may rely on fresh memory being zero */
    auto alloc = [&](void **p, size_t bytes) -> int {
        HIP_TRY(hipMalloc(p, bytes));
        r->wf_allocs.push_back(*p);
        if (poison) { HIP_TRY(hipMemset(*p, 0xAB, bytes)); HIP_TRY(hipDeviceSynchronize()); }
        return EU_OK;
    };
    int rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = alloc((void **)&B.ray[k], ray_cap * 2 * D * sizeof(double)))) return rc;
        if ((rc = alloc((void **)&B.ray_pa[k], ray_cap * sizeof(uint2)))) return rc;
    }
    if ((rc = alloc((void **)&B.hit, ray_cap * sizeof(EuWfHit)))) return rc;
    /* node ids are static: pixel roots, then generation g's queue slot q at pixels + g*ray_cap + q
     * (only the slots that hold rays are ever touched) */
    /* the finish step's pool: its node chunks lie behind the pipeline's slots in the same array (one id space for deliveries) */
    EuTsPool &F = r->wf_fin[set];
    memset(&F, 0, sizeof F);
    const unsigned fin_grid = (unsigned)r->num_cus * 2u;
    const uint32_t fin_nch = EU_TS_NCH;
    const size_t chunk_base = (node_cap + EU_TS_NCN - 1) / EU_TS_NCN;
    const size_t fin_rays = (size_t)r->wf_finish_rays < ray_cap ? (size_t)r->wf_finish_rays : ray_cap;      /* the hand-over generation holds at most this many */
    const size_t fin_chunks = (fin_rays * 6) / EU_TS_NCN + (size_t)fin_grid * (EU_MAX_DEPTH + 2) + 1;
    if ((chunk_base + fin_chunks) * (size_t)EU_TS_NCN > 0xfffffff0ull) { r->err = "frame too large for 32-bit node indices; render it in row tiles"; return EU_ERR_CAPACITY; }
    if ((rc = alloc((void **)&B.nodes, (chunk_base + fin_chunks) * EU_TS_NCN * sizeof(EuTsNode)))) return rc;
    if ((rc = alloc((void **)&B.node_kind, node_cap))) return rc;
    if ((rc = alloc((void **)&B.seg_count, (size_t)(EU_MAX_DEPTH + 2) * n_seg * 4))) return rc;
    HIP_TRY(hipMemset(B.seg_count, 0, (size_t)(EU_MAX_DEPTH + 2) * n_seg * 4));
    /* hipMemset on device memory is asynchronous to the host and ordered on the NULL stream only, which the (non-blocking) trace
     * streams do not synchronise with: without this wait the clear can land after the first frame's gen kernel has published its
     * queue lengths (seen as whole strips of unwritten pixels, first frame of a fresh renderer, only with several hardware queues) */
    HIP_TRY(hipDeviceSynchronize());
    B.ray_cap = (uint32_t)ray_cap; B.node_cap = (uint32_t)node_cap;
    B.n_seg = n_seg; B.seg_cap = (uint32_t)seg_cap;
    if (r->wf_finish_rays) {
        const size_t chunks = (size_t)fin_grid * fin_nch;
        if ((rc = alloc((void **)&F.ray_od, chunks * 2 * D * EU_TS_CH * sizeof(double)))) return rc;
        if ((rc = alloc((void **)&F.ray_parent, chunks * EU_TS_CH * 4))) return rc;
        if ((rc = alloc((void **)&F.ray_aux, chunks * EU_TS_CH * 4))) return rc;
        if ((rc = alloc((void **)&F.hit_t, (size_t)fin_grid * EU_TS_CH * 8))) return rc;
        if ((rc = alloc((void **)&F.hit_code, (size_t)fin_grid * EU_TS_CH * 4))) return rc;
        if ((rc = alloc((void **)&F.nchunk_prev, fin_chunks * 4))) return rc;
        if ((rc = alloc((void **)&F.wg_counters, (size_t)fin_grid * EU_TS_ROW * sizeof(unsigned long long)))) return rc;
        if ((rc = alloc((void **)&r->wf_d_totals[set], (EU_MAX_DEPTH + 2) * 4))) return rc;
        if ((rc = alloc((void **)&r->wf_fin_counters[set], sizeof(EuDevCounters)))) return rc;
        F.nodes = B.nodes; F.n_node_chunks = (uint32_t)fin_chunks; F.node_chunk_base = (uint32_t)chunk_base; F.n_wg = fin_grid; F.nch = fin_nch;
        /* nchunk_prev is indexed by the global chunk id */
        F.nchunk_prev -= chunk_base;
        r->wf_fin_grid = fin_grid;
        if (!r->wf_h_totals[set]) {
            HIP_TRY(hipHostMalloc((void **)&r->wf_h_totals[set], (EU_MAX_DEPTH + 2) * 4, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&r->wf_totals_ready[set], hipEventDisableTiming));
        }
        r->wf_totals_pending[set] = false;
        r->wf_handover[set] = EU_MAX_DEPTH + 1;      /* the first frame runs every generation and learns their sizes */
    }
    }
    if (!r->wf_stream[0]) {
        for (int k = 0; k < eu_renderer::WF_MAX_STREAMS; k++) { HIP_TRY(hipStreamCreateWithFlags(&r->wf_stream[k], hipStreamNonBlocking)); HIP_TRY(hipEventCreateWithFlags(&r->wf_join[k], hipEventDisableTiming)); }
        HIP_TRY(hipEventCreateWithFlags(&r->wf_fork, hipEventDisableTiming));
    }
    r->wf_pixels = pixels;
    return EU_OK;
}

template <class K> static int wf_grid(eu_renderer *r, K kern, size_t lds_bytes, unsigned &grid) {
    int blocks_per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kern, EU_WF_BLOCK, lds_bytes));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    grid = (unsigned)(r->num_cus * blocks_per_cu);
    return EU_OK;
}

/* rays per generation of one buffer set's queues (read back asynchronously: the next frame's hand-over generation) */
__global__ void eu_wf_totals_kernel(EuWfBuffers B, uint32_t *__restrict__ out) {
    __shared__ uint32_t acc[EU_MAX_DEPTH + 2];
    if (threadIdx.x < EU_MAX_DEPTH + 2) acc[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t g = 0; g <= EU_MAX_DEPTH; g++) {
        uint32_t s = 0;
        for (uint32_t k = threadIdx.x; k < B.n_seg; k += blockDim.x) s += B.seg_count[g * B.n_seg + k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&acc[g], s);
    }
    __syncthreads();
    if (threadIdx.x <= EU_MAX_DEPTH) out[threadIdx.x] = acc[threadIdx.x];
}

/* the FINISH step: the stream kernel takes over the queue of generation `gen` and everything below it */
template <int D, int HSCAP, bool LDS>
static int wf_launch_finish(eu_renderer *r, hipStream_t stream, int set, uint32_t gen, const EuDevCamera &dc, const EuDevFrame &df, const EuWfBuffers &B, uint32_t *rgba, eu_f64 *point) {
    auto kern = eu_ts_kernel<D, HSCAP, LDS>;
    uint32_t hs_cap = r->hit_cap < 8 ? 8u : ((r->hit_cap + 3u) & ~3u);
    const size_t hs_bytes = HSCAP == 0 ? (size_t)(EU_TS_BLOCK / 64) * hs_cap * 64 * 12 : 0;
    const size_t color_lds = (size_t)(r->color_depth ? r->color_depth : 1u) * 4 * sizeof(double) * EU_TS_BLOCK;
    const size_t shade_bytes = (LDS ? (size_t)r->scene_words * 8 : 0) + color_lds;
    const size_t lds_bytes = hs_bytes > shade_bytes ? hs_bytes : shade_bytes;
    int blocks_per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kern, EU_TS_BLOCK, lds_bytes));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    unsigned grid = (unsigned)(r->num_cus * blocks_per_cu);
    if (grid > r->wf_fin_grid) grid = r->wf_fin_grid;
    EuTsParams prm;
    memset(&prm, 0, sizeof prm);
    prm.scene_g = r->d_scene; prm.scene_words = r->scene_words; prm.hs_cap = hs_cap;
    prm.cam = dc; prm.fr = df; prm.P = r->wf_fin[set];
    prm.counters = r->wf_fin_counters[set]; prm.rgba = rgba; prm.hit_t_aov = nullptr; prm.point_rgb = point;
    prm.import_gen = gen; prm.imp_n_seg = B.n_seg; prm.imp_seg_cap = B.seg_cap; prm.imp_ray_cap = B.ray_cap;
    prm.imp_ray = B.ray[gen & 1u]; prm.imp_ray_pa = B.ray_pa[gen & 1u];
    prm.imp_seg_count = B.seg_count + (size_t)gen * B.n_seg; prm.imp_seg_count_rows = B.seg_count;
    prm.stats_counters = r->d_counters;
    HIP_TRY(hipMemsetAsync(r->wf_fin_counters[set], 0, sizeof(EuDevCounters), stream));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(EU_TS_BLOCK), lds_bytes, stream, prm);
    HIP_TRY(hipGetLastError());
    return EU_OK;
}

template <int D>
static int wf_launch_frame(eu_renderer *r, hipStream_t caller_stream, const EuDevCamera &dc, const EuDevFrame &df_in, uint32_t *rgba, eu_f64 *hit_t, eu_f64 *point) {
    /* Large frames are traced in bands of whole 8-row tiles so that the queue and node buffers stay bounded
     * (a band of 4 Mpixel needs ~25 GB at depth 16; an 8K frame goes through in 8 passes). */
    uint32_t band_rows = df_in.local_rows;
    bool two_streams = false;
    if (!df_in.single_pixel) {
        uint64_t rows_fit = r->wf_band_pixels / (df_in.width ? df_in.width : 1);
        rows_fit = rows_fit / 8 * 8;
        if (rows_fit < 8) rows_fit = 8;
        if (rows_fit < band_rows) band_rows = (uint32_t)rows_fit;
        /* at least two bands for anything but tiny frames: they run on two streams, so the tail and the launch gap
         * of one band's kernel are filled by the other band's kernel */
        if ((uint64_t)df_in.local_rows * df_in.width >= (1u << 19) && r->wf_n_streams > 1) {
            const uint32_t ns = (uint32_t)r->wf_n_streams;
            const uint32_t part = (((df_in.local_rows + ns - 1) / ns) + 7) / 8 * 8;
            if (part < band_rows) band_rows = part;
            two_streams = true;
        }
    }
    const size_t band_pixels = (size_t)band_rows * df_in.width;
    int rc = wf_ensure(r, df_in.single_pixel ? 64 : band_pixels, dc.max_depth, two_streams ? r->wf_n_streams : 1);
    if (rc != EU_OK) return rc;
    if (r->prepare_only) return EU_OK;      /* buffers, streams and events exist now: nothing is allocated while the frame is in flight */
    uint32_t hs_cap = r->hit_cap < 8 ? 8u : ((r->hit_cap + 3u) & ~3u);
    if (r->dbg_hs_cap) hs_cap = r->dbg_hs_cap;      /* diagnostics only */
    const size_t isect_lds = (size_t)(EU_WF_BLOCK / 64) * hs_cap * 64 * (sizeof(double) + 4);     /* (t, code) per entry; the intersect kernel reads the scene through scalar loads */
    unsigned g_isect, g_res;
    const bool hs_lds = r->hit_cap <= 32 && !r->dbg_hs_private;      /* else: private (scratch) hit stack */
    const bool hs_small = !hs_lds && r->hit_cap <= 16;
    if (hs_small) { if ((rc = wf_grid(r, eu_wf_intersect_kernel<D, 16>, 0, g_isect))) return rc; }
    else if (!hs_lds) { if ((rc = wf_grid(r, eu_wf_intersect_kernel<D, 96>, 0, g_isect))) return rc; }
    else if ((rc = wf_grid(r, eu_wf_intersect_kernel<D, 0>, isect_lds, g_isect))) return rc;
    if ((rc = wf_grid(r, eu_wf_resolve_kernel<D>, 0, g_res))) return rc;
    /* shade kernel's dynamic LDS: the colour-operand stack (color_depth RGBA doubles per lane) and, when three workgroups
     * per CU still fit (160 KB / 3, minus ~8 KB static), a copy of the flat scene */
    const size_t color_lds = (size_t)(r->color_depth ? r->color_depth : 1u) * 4 * sizeof(double) * EU_WF_BLOCK;
    const bool shade_lds = (size_t)r->scene_words * 8 + color_lds <= 44 * 1024 && !r->dbg_shade_scene_global;
    if (two_streams) {      /* fork: both side streams wait for everything queued on the caller's stream so far */
        HIP_TRY(hipEventRecord(r->wf_fork, caller_stream));
        for (int k = 0; k < r->wf_n_streams; k++) HIP_TRY(hipStreamWaitEvent(r->wf_stream[k], r->wf_fork, 0));
    }
    const uint32_t dbg_skip = r->dbg_skip_entities;   /* -DEU_PROFILE_ISECT / -DEU_DEBUG_SKIP builds only */
    uint32_t band_no = 0;
    for (uint32_t row0 = 0; row0 < df_in.local_rows; row0 += band_rows, band_no++) {
        EuDevFrame df = df_in;
        const int set = two_streams ? (int)(band_no % (uint32_t)r->wf_n_streams) : 0;
        hipStream_t stream = two_streams ? r->wf_stream[set] : caller_stream;
        EuWfBuffers B = r->wf[set];
        if (df.single_pixel) { df.band_row0 = 0; df.band_rows = 1; df.root_base = 0; B.npix = 1u; }
        else {
            df.band_row0 = row0;
            df.band_rows = df_in.local_rows - row0 < band_rows ? df_in.local_rows - row0 : band_rows;
            df.root_base = row0 * df.width;
            df.n_tiles = df.tiles_x * ((df.band_rows + 7) / 8);
            B.npix = df.band_rows * df.width;
        }
        const unsigned g_prod = B.n_seg;      /* producers: one output segment per workgroup */
        /* hand-over generation of this band: from the queue lengths of the last frame traced with this buffer set, if they have arrived */
        uint32_t handover = dc.max_depth;
        if (r->wf_finish_rays && !df.single_pixel) {
            if (r->wf_totals_pending[set] && hipEventQuery(r->wf_totals_ready[set]) == hipSuccess) {
                r->wf_totals_pending[set] = false;
                const uint32_t known = r->wf_totals_handover[set];      /* generations 0..known were traced (or handed over) by the pipeline: their lengths are exact */
                uint32_t hnew = known + 1;
                for (uint32_t g = 1; g <= known && g <= EU_MAX_DEPTH; g++) if (r->wf_h_totals[set][g] < r->wf_finish_rays) { hnew = g; break; }
                r->wf_handover[set] = r->wf_totals_pixels[set] == (size_t)B.npix ? hnew : EU_MAX_DEPTH + 1;
            } else if (hipGetLastError() != hipSuccess) { /* hipErrorNotReady from the query: nothing to report */ }
            handover = r->wf_handover[set] < dc.max_depth ? r->wf_handover[set] : dc.max_depth;
            if (handover < 1) handover = 1;
        }
        hipLaunchKernelGGL(eu_wf_gen_kernel<D>, dim3(g_prod), dim3(EU_WF_BLOCK), 0, stream, r->d_scene, r->scene_words, dc, df, B, r->d_counters, rgba, hit_t, point);
        for (uint32_t g = 0; g < handover; g++) {
            if (hs_lds) hipLaunchKernelGGL((eu_wf_intersect_kernel<D, 0>), dim3(g_isect), dim3(EU_WF_BLOCK), isect_lds, stream, r->d_scene, r->scene_words | dbg_skip, hs_cap, g, df.root_base, B, r->d_counters, hit_t);
            else if (hs_small) hipLaunchKernelGGL((eu_wf_intersect_kernel<D, 16>), dim3(g_isect), dim3(EU_WF_BLOCK), 0, stream, r->d_scene, r->scene_words, 16u, g, df.root_base, B, r->d_counters, hit_t);
            else hipLaunchKernelGGL((eu_wf_intersect_kernel<D, 96>), dim3(g_isect), dim3(EU_WF_BLOCK), 0, stream, r->d_scene, r->scene_words, 96u, g, df.root_base, B, r->d_counters, hit_t);
            if (shade_lds) hipLaunchKernelGGL((eu_wf_shade_kernel<D, true>), dim3(g_prod), dim3(EU_WF_BLOCK), (size_t)r->scene_words * 8 + color_lds, stream, r->d_scene, r->scene_words, g, dc.max_depth | r->dbg_skip_shade, df.time_s, B, r->d_counters, rgba, point);
            else hipLaunchKernelGGL((eu_wf_shade_kernel<D, false>), dim3(g_prod), dim3(EU_WF_BLOCK), color_lds, stream, r->d_scene, r->scene_words, g, dc.max_depth | r->dbg_skip_shade, df.time_s, B, r->d_counters, rgba, point);
        }
        if (handover < dc.max_depth) {
            int frc;
            if (hs_lds) frc = shade_lds ? wf_launch_finish<D, 0, true>(r, stream, set, handover, dc, df, B, rgba, point) : wf_launch_finish<D, 0, false>(r, stream, set, handover, dc, df, B, rgba, point);
            else frc = shade_lds ? wf_launch_finish<D, 96, true>(r, stream, set, handover, dc, df, B, rgba, point) : wf_launch_finish<D, 96, false>(r, stream, set, handover, dc, df, B, rgba, point);
            if (frc != EU_OK) return frc;
        }
        for (uint32_t g = handover; g-- > 0;)
            hipLaunchKernelGGL(eu_wf_resolve_kernel<D>, dim3(g_res), dim3(EU_WF_BLOCK), 0, stream, g, B, r->d_counters, rgba, point);
        if (r->wf_finish_rays && !df.single_pixel && !r->wf_totals_pending[set]) {      /* this band's queue lengths, for the next frame */
            hipLaunchKernelGGL(eu_wf_totals_kernel, dim3(1), dim3(256), 0, stream, B, r->wf_d_totals[set]);
            HIP_TRY(hipMemcpyAsync(r->wf_h_totals[set], r->wf_d_totals[set], (EU_MAX_DEPTH + 1) * 4, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipEventRecord(r->wf_totals_ready[set], stream));
            r->wf_totals_pending[set] = true;
            r->wf_totals_handover[set] = handover;
            r->wf_totals_pixels[set] = (size_t)B.npix;
        }
        if (df.single_pixel) break;
    }
    if (two_streams) {      /* join */
        for (int k = 0; k < r->wf_n_streams; k++) { HIP_TRY(hipEventRecord(r->wf_join[k], r->wf_stream[k])); HIP_TRY(hipStreamWaitEvent(caller_stream, r->wf_join[k], 0)); }
    }
    HIP_TRY(hipGetLastError());
    return EU_OK;
}

/* ------------------------------------------------------------------ stream kernel (trace_stream.h) */
static int ts_ensure(eu_renderer *r, unsigned grid, uint32_t nch, size_t node_chunks) {
    if (grid <= r->ts_grid_cap && nch <= r->ts_nch && node_chunks <= r->ts_node_chunks) return EU_OK;
    if (grid < r->ts_grid_cap) grid = r->ts_grid_cap;
    if (nch < r->ts_nch) nch = r->ts_nch;
    if (node_chunks < r->ts_node_chunks) node_chunks = r->ts_node_chunks;
    for (void *p : r->ts_allocs) (void)hipFree(p);      /* (hipFree waits for the device) */
    r->ts_allocs.clear();
    r->ts_grid_cap = 0; r->ts_nch = 0; r->ts_node_chunks = 0;
    if (node_chunks * (size_t)EU_TS_NCN > 0xfffffff0ull) { r->err = "frame too large for 32-bit node indices; render it in row tiles"; return EU_ERR_CAPACITY; }
    EuTsPool &P = r->ts;
    memset(&P, 0, sizeof P);
    auto alloc = [&](void **p, size_t bytes) -> int {
        HIP_TRY(hipMalloc(p, bytes));
        r->ts_allocs.push_back(*p);
        return EU_OK;
    };
    const size_t D = (size_t)r->dim, chunks = (size_t)grid * nch;
    int rc;
    if ((rc = alloc((void **)&P.ray_od, chunks * 2 * D * EU_TS_CH * sizeof(double)))) return rc;
    if ((rc = alloc((void **)&P.ray_parent, chunks * EU_TS_CH * 4))) return rc;
    if ((rc = alloc((void **)&P.ray_aux, chunks * EU_TS_CH * 4))) return rc;
    if ((rc = alloc((void **)&P.hit_t, (size_t)grid * EU_TS_CH * 8))) return rc;
    if ((rc = alloc((void **)&P.hit_code, (size_t)grid * EU_TS_CH * 4))) return rc;
    if ((rc = alloc((void **)&P.nodes, node_chunks * EU_TS_NCN * sizeof(EuTsNode)))) return rc;
    if ((rc = alloc((void **)&P.nchunk_prev, node_chunks * 4))) return rc;
    if ((rc = alloc((void **)&P.wg_counters, (size_t)grid * EU_TS_ROW * sizeof(unsigned long long)))) return rc;
    P.n_node_chunks = (uint32_t)node_chunks; P.n_wg = grid; P.nch = nch;
    r->ts_grid_cap = grid; r->ts_nch = nch; r->ts_node_chunks = node_chunks;
    return EU_OK;
}

template <int D, int HSCAP, bool LDS>
static int ts_launch(eu_renderer *r, hipStream_t stream, const EuDevCamera &dc, const EuDevFrame &df, uint32_t *rgba, eu_f64 *hit_t, eu_f64 *point) {
    auto kern = eu_ts_kernel<D, HSCAP, LDS>;
    uint32_t hs_cap = r->hit_cap < 8 ? 8u : ((r->hit_cap + 3u) & ~3u);
    const size_t hs_bytes = HSCAP == 0 ? (size_t)(EU_TS_BLOCK / 64) * hs_cap * 64 * 12 : 0;
    const size_t color_lds = (size_t)(r->color_depth ? r->color_depth : 1u) * 4 * sizeof(double) * EU_TS_BLOCK;
    const size_t shade_bytes = (LDS ? (size_t)r->scene_words * 8 : 0) + color_lds;
    const size_t lds_bytes = hs_bytes > shade_bytes ? hs_bytes : shade_bytes;      /* the two phases share the space */
    int blocks_per_cu = 0;
    HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, kern, EU_TS_BLOCK, lds_bytes));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    unsigned grid = (unsigned)(r->num_cus * blocks_per_cu);
    if (r->ts_grid_limit && grid > r->ts_grid_limit) grid = r->ts_grid_limit;
    const unsigned work_tiles = (df.n_tiles + 3u) / 4u;
    if (grid > work_tiles) grid = work_tiles;
    if (grid < 1) grid = 1;
    uint32_t nch = 3u * dc.max_depth + 4u;      /* <= 2 full chunks per generation + 1 open, the one in work and 3 spares */
    if (nch > EU_TS_NCH) nch = EU_TS_NCH;
    const size_t pixels = df.single_pixel ? 64 : (size_t)df.local_rows * df.width;
    const size_t node_chunks = (size_t)((double)pixels * r->ts_node_factor) / EU_TS_NCN + (size_t)grid * (dc.max_depth + 2u) + 1;
    int rc = ts_ensure(r, grid, nch, node_chunks);
    if (rc != EU_OK) return rc;
    EuTsParams prm;
    memset(&prm, 0, sizeof prm);
    prm.scene_g = r->d_scene; prm.scene_words = r->scene_words; prm.hs_cap = hs_cap;
    prm.cam = dc; prm.fr = df; prm.P = r->ts;
    prm.counters = r->d_counters; prm.rgba = rgba; prm.hit_t_aov = hit_t; prm.point_rgb = point;
    prm.import_gen = 0xffffffffu;      /* frame mode: the rays come from the camera */
    hipLaunchKernelGGL(kern, dim3(grid), dim3(EU_TS_BLOCK), lds_bytes, stream, prm);
    HIP_TRY(hipGetLastError());
    r->ts_grid_last = grid;
    return EU_OK;
}

template <int D>
static int ts_launch_frame(eu_renderer *r, hipStream_t stream, const EuDevCamera &dc, const EuDevFrame &df_in, uint32_t *rgba, eu_f64 *hit_t, eu_f64 *point) {
    EuDevFrame df = df_in;
    df.band_row0 = 0; df.band_rows = df.local_rows; df.root_base = 0;
    const bool hs_lds = r->hit_cap <= 32;      /* else: private (scratch) hit stack */
    const size_t color_lds = (size_t)(r->color_depth ? r->color_depth : 1u) * 4 * sizeof(double) * EU_TS_BLOCK;
    const bool scene_lds = (size_t)r->scene_words * 8 + color_lds <= 48 * 1024 && !r->dbg_shade_scene_global;
    if (hs_lds) return scene_lds ? ts_launch<D, 0, true>(r, stream, dc, df, rgba, hit_t, point) : ts_launch<D, 0, false>(r, stream, dc, df, rgba, hit_t, point);
    return scene_lds ? ts_launch<D, 96, true>(r, stream, dc, df, rgba, hit_t, point) : ts_launch<D, 96, false>(r, stream, dc, df, rgba, hit_t, point);
}

/* rays, background samples, would-panic counters of a stream-kernel frame: one row per workgroup */
static void ts_sum_rows(const unsigned long long *rows, unsigned n, eu_stats *out, unsigned long long *overflow) {
    unsigned long long v[5] = {0, 0, 0, 0, 0};
    for (unsigned w = 0; w < n; w++) for (int k = 0; k < 5; k++) v[k] += rows[(size_t)w * EU_TS_ROW + k];
    if (out) { out->rays = v[0]; out->bg_samples = v[1]; out->nan_pixels = v[2]; out->errors = v[3]; }
    *overflow = v[4];
}

static int render_device_impl(eu_renderer *r, const eu_camera *cam, const eu_frame *f, hipStream_t stream, uint32_t *rgba, eu_f64 *hit_t, eu_f64 *point,
                              bool single = false, uint32_t single_x = 0) {
    if (!r || !cam || !f || !rgba) return EU_ERR_INVALID_ARGUMENT;
    if (cam->dim != r->dim) { r->err = "camera dimension does not match the scene"; return EU_ERR_INVALID_ARGUMENT; }
    if (f->width == 0 || f->height == 0 || f->row_begin > f->row_end || f->row_end > f->height) { r->err = "bad frame"; return EU_ERR_INVALID_ARGUMENT; }
    if (cam->max_depth > EU_MAX_DEPTH) { r->err = "max_depth exceeds the compiled frame-stack depth (16)"; return EU_ERR_CAPACITY; }
    HIP_TRY(hipSetDevice(r->device));
    EuDevCamera dc;
    make_dev_camera(cam, f, dc);
    EuDevFrame df;
    memset(&df, 0, sizeof df);
    df.width = f->width; df.height = f->height; df.row_begin = f->row_begin; df.row_end = f->row_end;
    uint32_t rows = f->row_end - f->row_begin;
    if (f->strip_count > 1) {
        if (f->strip_index >= f->strip_count) { r->err = "strip_index >= strip_count"; return EU_ERR_INVALID_ARGUMENT; }
        rows = eu_frame_local_rows(f);
        df.strip_count = f->strip_count; df.strip_index = f->strip_index;
    }
    df.local_rows = rows;
    df.tiles_x = (f->width + 7) / 8;
    df.n_tiles = df.tiles_x * ((rows + 7) / 8);
    df.debug_crosshair = f->debug_crosshair ? 1u : 0u;
    df.time_s = (double)f->time_ms / R(1000.0);
    if (single) { df.strip_count = 0; df.local_rows = 1; df.single_pixel = 1; df.single_x = single_x; df.single_y = f->row_begin; df.tiles_x = 1; df.n_tiles = 1; }
    if (rows == 0) return EU_OK;
    if (r->prepare_only) {
        if (r->path == eu_renderer::PATH_WAVEFRONT && r->use_wavefront)
            return (r->dim == 3) ? wf_launch_frame<3>(r, stream, dc, df, rgba, hit_t, point) : wf_launch_frame<4>(r, stream, dc, df, rgba, hit_t, point);
        return EU_OK;
    }
    HIP_TRY(hipMemsetAsync(r->d_counters, 0, sizeof(EuDevCounters), stream));
    const int slot = (int)(r->launches % eu_renderer::EV_RING);
    HIP_TRY(hipEventRecord(r->ev_start[slot], stream));
    hipError_t e = hipSuccess;
    r->ts_last = r->path == eu_renderer::PATH_STREAM;
    if (r->path == eu_renderer::PATH_STREAM) {
        int rc = (r->dim == 3) ? ts_launch_frame<3>(r, stream, dc, df, rgba, hit_t, point) : ts_launch_frame<4>(r, stream, dc, df, rgba, hit_t, point);
        if (rc != EU_OK) return rc;
    } else if (r->use_wavefront) {
        int rc = (r->dim == 3) ? wf_launch_frame<3>(r, stream, dc, df, rgba, hit_t, point) : wf_launch_frame<4>(r, stream, dc, df, rgba, hit_t, point);
        if (rc != EU_OK) return rc;
    } else {
    /* hit stack in LDS when the scene's static bound is small (16 entries * 12 B * 256 lanes = 48 KB per block) */
    const bool hs_lds = r->hit_cap <= 32 && r->scene_in_lds;
    if (r->dim == 3) {
        if (hs_lds) e = launch_trace<3, 0, true>(r, stream, dc, df, rgba, hit_t, point);
        else if (r->scene_in_lds) e = launch_trace<3, 96, true>(r, stream, dc, df, rgba, hit_t, point);
        else e = launch_trace<3, 96, false>(r, stream, dc, df, rgba, hit_t, point);
    } else {
        if (hs_lds) e = launch_trace<4, 0, true>(r, stream, dc, df, rgba, hit_t, point);
        else if (r->scene_in_lds) e = launch_trace<4, 96, true>(r, stream, dc, df, rgba, hit_t, point);
        else e = launch_trace<4, 96, false>(r, stream, dc, df, rgba, hit_t, point);
    }
    }
    if (e != hipSuccess) { r->err = std::string("kernel launch: ") + hipGetErrorString(e); return EU_ERR_HIP; }
    HIP_TRY(hipEventRecord(r->ev_stop[slot], stream));
    r->launches++;
    r->last_stream = stream;
    r->have_timing = true;
    return EU_OK;
}

extern "C" uint32_t eu_frame_local_rows(const eu_frame *f) {
    if (!f || f->row_begin > f->row_end) return 0;
    const uint32_t rows = f->row_end - f->row_begin;
    if (f->strip_count <= 1) return rows;
    const uint32_t strips = (rows + 7) / 8;                       /* strips of 8 rows over [row_begin,row_end) */
    if (f->strip_index >= f->strip_count) return 0;
    const uint32_t mine = (strips + f->strip_count - 1 - f->strip_index) / f->strip_count;   /* strips s with s % count == index */
    return mine * 8;                                              /* padded: rows past row_end are left untouched */
}

extern "C" int eu_render_device(eu_renderer *r, const eu_camera *cam, const eu_frame *f, void *hip_stream, void *rgba_dev, eu_f64 *hit_t_dev) {
    return render_device_impl(r, cam, f, (hipStream_t)hip_stream, (uint32_t *)rgba_dev, hit_t_dev, nullptr);
}

extern "C" int eu_pack_rgb_device(eu_renderer *r, const void *rgba_dev, void *rgb_dev, size_t pixels, void *hip_stream) {
    if (!r || !rgba_dev || !rgb_dev) return EU_ERR_INVALID_ARGUMENT;
    if (pixels == 0) return EU_OK;
    HIP_TRY(hipSetDevice(r->device));
    size_t nq = (pixels + 3) / 4;
    hipLaunchKernelGGL(eu_pack_rgb_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, (hipStream_t)hip_stream, (const uint32_t *)rgba_dev, (uint8_t *)rgb_dev, pixels);
    HIP_TRY(hipGetLastError());
    return EU_OK;
}

extern "C" int eu_renderer_stats(eu_renderer *r, eu_stats *out) {
    if (!r || !out) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->last_stream));
    if (r->ts_last) {
        std::vector<unsigned long long> rows((size_t)r->ts_grid_last * EU_TS_ROW);
        HIP_TRY(hipMemcpy(rows.data(), r->ts.wg_counters, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        unsigned long long overflow = 0;
        ts_sum_rows(rows.data(), r->ts_grid_last, out, &overflow);
        if (overflow) { r->err = "tree-node pool exhausted (" + std::to_string(overflow) + " workgroups stopped): raise EU_TS_NODE_FACTOR or render in row tiles"; return EU_ERR_CAPACITY; }
        return EU_OK;
    }
    EuDevCounters c;
    HIP_TRY(hipMemcpy(&c, r->d_counters, sizeof c, hipMemcpyDeviceToHost));
    out->rays = c.rays; out->bg_samples = c.bg_samples; out->nan_pixels = c.nan_pixels; out->errors = c.errors;
    if (c.overflow) { r->err = "wavefront queue overflow (" + std::to_string(c.overflow) + " rays dropped): raise EU_WF_RAY_FACTOR or render in row tiles"; return EU_ERR_CAPACITY; }
    return EU_OK;
}

extern "C" int eu_renderer_debug_phases(eu_renderer *r, unsigned long long out[16]) {
    if (!r || !out) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->last_stream));
    if (r->ts_last) {      /* -DEU_TS_PROFILE builds: clock shares summed over the workgroups; [11] = workgroups, [12] = the longest workgroup's total */
        std::vector<unsigned long long> rows((size_t)r->ts_grid_last * EU_TS_ROW);
        HIP_TRY(hipMemcpy(rows.data(), r->ts.wg_counters, rows.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int i = 0; i < 16; i++) out[i] = 0;
        for (unsigned w = 0; w < r->ts_grid_last; w++) {
            unsigned long long tot = 0;
            for (int k = 0; k < 11; k++) { out[k] += rows[(size_t)w * EU_TS_ROW + 5 + k]; if (k < 7) tot += rows[(size_t)w * EU_TS_ROW + 5 + k]; }
            if (tot > out[12]) out[12] = tot;
        }
        out[11] = r->ts_grid_last;
        return EU_OK;
    }
    EuDevCounters c;
    HIP_TRY(hipMemcpy(&c, r->d_counters, sizeof c, hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; i++) out[i] = c.phase[i];
    return EU_OK;
}

extern "C" int eu_renderer_debug_generations(eu_renderer *r, unsigned long long out[17]) {
    if (!r || !out) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    for (int g = 0; g < 17; g++) out[g] = 0;
    const EuWfBuffers &B = r->wf[0];
    if (!B.seg_count || !B.n_seg) return EU_OK;
    std::vector<uint32_t> h((size_t)(EU_MAX_DEPTH + 2) * B.n_seg);
    HIP_TRY(hipMemcpy(h.data(), B.seg_count, h.size() * 4, hipMemcpyDeviceToHost));
    for (int g = 0; g < 17 && g < EU_MAX_DEPTH + 2; g++) for (uint32_t i = 0; i < B.n_seg; i++) out[g] += h[(size_t)g * B.n_seg + i];
    return EU_OK;
}

extern "C" int eu_renderer_kernel_ms(eu_renderer *r, float *ms) {
    return eu_renderer_kernel_ms_history(r, ms, 1) == 1 ? EU_OK : EU_ERR_INVALID_ARGUMENT;
}

extern "C" int eu_renderer_retraces(eu_renderer *r, uint64_t *count) {
    if (!r || !count) return EU_ERR_INVALID_ARGUMENT;
    *count = r->retraces;
    return EU_OK;
}

extern "C" int eu_renderer_kernel_ms_history(eu_renderer *r, float *ms, int max_n) {
    if (!r || !ms || max_n < 1) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    unsigned long long have = r->launches < (unsigned long long)eu_renderer::EV_RING ? r->launches : (unsigned long long)eu_renderer::EV_RING;
    int n = (int)(have < (unsigned long long)max_n ? have : (unsigned long long)max_n);
    for (int i = 0; i < n; i++) {
        const int slot = (int)((r->launches - (unsigned long long)n + (unsigned long long)i) % eu_renderer::EV_RING);
        HIP_TRY(hipEventSynchronize(r->ev_stop[slot]));
        HIP_TRY(hipEventElapsedTime(&ms[i], r->ev_start[slot], r->ev_stop[slot]));
    }
    return n;
}

static int ensure_buffers(eu_renderer *r, size_t pixels, bool want_hit) {
    if (pixels > r->rgba_pixels) {
        if (r->d_rgba) (void)hipFree(r->d_rgba);
        if (r->d_rgb) (void)hipFree(r->d_rgb);
        if (r->d_hit) { (void)hipFree(r->d_hit); r->d_hit = nullptr; }
        r->d_rgba = nullptr; r->d_rgb = nullptr; r->rgba_pixels = 0;
        HIP_TRY(hipMalloc((void **)&r->d_rgba, pixels * 4));
        HIP_TRY(hipMalloc((void **)&r->d_rgb, pixels * 3 + 16));
        r->rgba_pixels = pixels;
    }
    if (want_hit && !r->d_hit) HIP_TRY(hipMalloc((void **)&r->d_hit, r->rgba_pixels * sizeof(eu_f64)));
    return EU_OK;
}

extern "C" int eu_render(eu_renderer *r, const eu_camera *cam, const eu_frame *f, uint8_t *rgb_host, eu_f64 *hit_t_host, eu_stats *stats) {
    if (!r || !cam || !f || !rgb_host) return EU_ERR_INVALID_ARGUMENT;
    if (f->row_begin > f->row_end || f->row_end > f->height) return EU_ERR_INVALID_ARGUMENT;
    const size_t pixels = (size_t)eu_frame_local_rows(f) * f->width;
    if (pixels == 0) { if (stats) memset(stats, 0, sizeof *stats); return EU_OK; }
    HIP_TRY(hipSetDevice(r->device));
    int rc = ensure_buffers(r, pixels, hit_t_host != nullptr);
    if (rc != EU_OK) return rc;
    rc = render_device_impl(r, cam, f, nullptr, r->d_rgba, hit_t_host ? r->d_hit : nullptr, nullptr);
    if (rc != EU_OK) return rc;
    if (r->ts_last) {
        /* The stream kernel's ray queues cannot overflow; its tree-node pool can (more than ts_node_factor nodes per pixel).
         * The pool is enlarged and the frame traced again. */
        for (int attempt = 0; attempt < 6; attempt++) {
            eu_stats tmp;
            const int src = eu_renderer_stats(r, &tmp);
            if (src != EU_ERR_CAPACITY) break;
            r->ts_node_factor = r->ts_node_factor * R(2.0) > R(2.0) ? r->ts_node_factor * R(2.0) : R(2.0);
            r->retraces++;
            rc = render_device_impl(r, cam, f, nullptr, r->d_rgba, hit_t_host ? r->d_hit : nullptr, nullptr);
            if (rc != EU_OK) return rc;
        }
    } else if (r->use_wavefront) {
        /* A frame whose recursion fans out beyond the queues' capacity (more than wf_ray_factor rays per pixel in one
         * generation) cannot be finished by the wavefront pipeline; the persistent stack-based kernel needs O(depth)
         * memory per lane whatever the fan-out, so the frame is traced again with it. */
        EuDevCounters c;
        HIP_TRY(hipMemcpy(&c, r->d_counters, sizeof c, hipMemcpyDeviceToHost));
        if (c.overflow) {
            r->use_wavefront = false;
            r->retraces++;
            rc = render_device_impl(r, cam, f, nullptr, r->d_rgba, hit_t_host ? r->d_hit : nullptr, nullptr);
            r->use_wavefront = true;
            if (rc != EU_OK) return rc;
        }
    }
    rc = eu_pack_rgb_device(r, r->d_rgba, r->d_rgb, pixels, nullptr);
    if (rc != EU_OK) return rc;
    HIP_TRY(hipMemcpy(rgb_host, r->d_rgb, pixels * 3, hipMemcpyDeviceToHost));
    if (hit_t_host) HIP_TRY(hipMemcpy(hit_t_host, r->d_hit, pixels * sizeof(eu_f64), hipMemcpyDeviceToHost));
    if (stats) return eu_renderer_stats(r, stats);
    return EU_OK;
}

extern "C" int eu_trace_screen_point(eu_renderer *r, const eu_camera *cam, const eu_frame *f, int32_t x, int32_t y, eu_f64 rgb[3]) {
    if (!r || !cam || !f || !rgb) return EU_ERR_INVALID_ARGUMENT;
    if (x < 0 || y < 0 || (uint32_t)x >= f->width || (uint32_t)y >= f->height) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    int rc = ensure_buffers(r, 64, false);
    if (rc != EU_OK) return rc;
    eu_frame one = *f;
    one.row_begin = (uint32_t)y; one.row_end = (uint32_t)y + 1;
    rc = render_device_impl(r, cam, &one, nullptr, r->d_rgba, nullptr, r->d_point, true, (uint32_t)x);
    if (rc != EU_OK) return rc;
    HIP_TRY(hipMemcpy(rgb, r->d_point, 3 * sizeof(eu_f64), hipMemcpyDeviceToHost));
    return EU_OK;
}

/* ------------------------------------------------------------------ frame sequences (scope row f4)
 * The reference's loop renders a frame, uploads it as a texture and only then starts the next one
 * (simulation.rs:93-150, universe/mod.rs:300-357).  Here a sequence keeps `slots` frames in flight: frame k+1 is traced
 * on the trace stream while frame k's RGB8 image travels to pinned host memory on the copy stream. */
struct eu_sequence {
    eu_renderer *r = nullptr;
    uint32_t max_pixels = 0;
    hipStream_t copy_stream = nullptr;
    /* Every slot traces on its own stream with its own work buffers and counters (a clone of the renderer: the scene and
     * its textures once more in HBM), so that consecutive frames overlap on the GPU: small frames are bound by the
     * pipeline's ~0.5 ms of dependent launches, not by throughput. */
    std::vector<eu_renderer *> slot_renderer;      /* [0] = r */
    std::vector<hipStream_t> slot_stream;
    struct Slot {
        uint32_t *d_rgba = nullptr; uint8_t *d_rgb = nullptr; unsigned char *d_cnt = nullptr;      /* counters: EuDevCounters, or the stream kernel's per-workgroup rows */
        uint8_t *h_rgb = nullptr; unsigned char *h_cnt = nullptr; unsigned cnt_rows = 0;        /* h_rgb: the pinned image this submit copies into (one of host_rgb) */
        hipEvent_t traced = nullptr, copied = nullptr;
        uint32_t width = 0, rows = 0;
    };
    std::vector<Slot> slots;
    /* slots + 1 pinned images, used round-robin by submit number: the image handed out by eu_sequence_next stays untouched
     * until the NEXT eu_sequence_next, however many frames are submitted in between */
    std::vector<uint8_t *> host_rgb;
    unsigned long long submitted = 0, taken = 0;
};

static constexpr size_t SEQ_CNT_BYTES = 4096 * EU_TS_ROW * sizeof(unsigned long long);      /* >= sizeof(EuDevCounters); up to 4096 workgroup rows */
static_assert(sizeof(EuDevCounters) <= SEQ_CNT_BYTES, "counter buffer");

extern "C" void eu_sequence_destroy(eu_sequence *q) {
    if (!q) return;
    if (q->r) (void)hipSetDevice(q->r->device);
    for (hipStream_t st : q->slot_stream) if (st) (void)hipStreamSynchronize(st);
    if (q->copy_stream) (void)hipStreamSynchronize(q->copy_stream);
    for (auto &s : q->slots) {
        if (s.d_rgba) (void)hipFree(s.d_rgba);
        if (s.d_rgb) (void)hipFree(s.d_rgb);
        if (s.d_cnt) (void)hipFree(s.d_cnt);
        if (s.h_cnt) (void)hipHostFree(s.h_cnt);
        if (s.traced) (void)hipEventDestroy(s.traced);
        if (s.copied) (void)hipEventDestroy(s.copied);
    }
    for (uint8_t *h : q->host_rgb) if (h) (void)hipHostFree(h);
    for (hipStream_t st : q->slot_stream) if (st) (void)hipStreamDestroy(st);
    if (q->copy_stream) (void)hipStreamDestroy(q->copy_stream);
    for (size_t k = 1; k < q->slot_renderer.size(); k++) eu_renderer_destroy(q->slot_renderer[k]);
    delete q;
}

extern "C" int eu_sequence_create(eu_renderer *r, uint32_t max_width, uint32_t max_height, uint32_t slots, eu_sequence **out) {
    if (!r || !out || max_width == 0 || max_height == 0 || slots == 0 || slots > 16) return EU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    HIP_TRY(hipSetDevice(r->device));
    eu_sequence *q = new eu_sequence();
    q->r = r;
    q->max_pixels = max_width * ((max_height + 7u) & ~7u);
    q->slots.resize(slots);
    auto fail = [&](hipError_t e, const char *what) { r->err = std::string(what) + ": " + hipGetErrorString(e); eu_sequence_destroy(q); return EU_ERR_HIP; };
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&q->copy_stream, hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
    q->slot_renderer.assign(slots, nullptr);
    q->slot_stream.assign(slots, nullptr);
    q->slot_renderer[0] = r;
    for (uint32_t k = 0; k < slots; k++) {
        if ((e = hipStreamCreateWithFlags(&q->slot_stream[k], hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if (k > 0) {
            char cerr[256] = "";
            const int crc = renderer_create_impl(r->flat, r->device, &q->slot_renderer[k], cerr, sizeof cerr);
            if (crc != EU_OK) { r->err = std::string("frame sequence slot: ") + cerr; q->slot_renderer.resize(k); eu_sequence_destroy(q); return crc; }
        }
    }
    for (auto &s : q->slots) {
        if ((e = hipMalloc((void **)&s.d_rgba, (size_t)q->max_pixels * 4)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMalloc((void **)&s.d_rgb, (size_t)q->max_pixels * 3 + 16)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMalloc((void **)&s.d_cnt, SEQ_CNT_BYTES)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipHostMalloc((void **)&s.h_cnt, SEQ_CNT_BYTES, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc");
        if ((e = hipEventCreateWithFlags(&s.traced, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
        if ((e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming)) != hipSuccess) return fail(e, "hipEventCreate");
    }
    q->host_rgb.assign(slots + 1, nullptr);
    for (auto &h : q->host_rgb)
        if ((e = hipHostMalloc((void **)&h, (size_t)q->max_pixels * 3 + 16, hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc");
    *out = q;
    return EU_OK;
}

extern "C" int eu_sequence_submit(eu_sequence *q, const eu_camera *cam, const eu_frame *f) {
    if (!q || !cam || !f) return EU_ERR_INVALID_ARGUMENT;
    eu_renderer *r = q->r;
    if (q->submitted - q->taken >= q->slots.size()) { r->err = "every slot of the sequence is in flight: take a frame with eu_sequence_next first"; return EU_ERR_BUSY; }
    const uint32_t rows = eu_frame_local_rows(f);
    const size_t pixels = (size_t)rows * f->width;
    if (pixels == 0 || pixels > q->max_pixels) { r->err = "frame does not fit the sequence's buffers"; return EU_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(r->device));
    const size_t slot_no = q->submitted % q->slots.size();
    eu_sequence::Slot &s = q->slots[slot_no];
    eu_renderer *rs = q->slot_renderer[slot_no];
    hipStream_t trace_stream = q->slot_stream[slot_no];
    int rc = render_device_impl(rs, cam, f, trace_stream, s.d_rgba, nullptr, nullptr);
    if (rc != EU_OK) { if (rs != r) r->err = rs->err; return rc; }
    rc = eu_pack_rgb_device(rs, s.d_rgba, s.d_rgb, pixels, trace_stream);
    if (rc != EU_OK) return rc;
    /* this slot's next frame overwrites its renderer's counters */
    s.cnt_rows = rs->ts_last ? rs->ts_grid_last : 0u;
    if (s.cnt_rows > 4096) { r->err = "more workgroups than the sequence's counter rows"; return EU_ERR_CAPACITY; }
    const size_t cnt_bytes = s.cnt_rows ? (size_t)s.cnt_rows * EU_TS_ROW * sizeof(unsigned long long) : sizeof(EuDevCounters);
    HIP_TRY(hipMemcpyAsync(s.d_cnt, s.cnt_rows ? (const void *)rs->ts.wg_counters : (const void *)rs->d_counters, cnt_bytes, hipMemcpyDeviceToDevice, trace_stream));
    HIP_TRY(hipEventRecord(s.traced, trace_stream));
    HIP_TRY(hipStreamWaitEvent(q->copy_stream, s.traced, 0));
    s.h_rgb = q->host_rgb[q->submitted % q->host_rgb.size()];
    HIP_TRY(hipMemcpyAsync(s.h_rgb, s.d_rgb, pixels * 3, hipMemcpyDeviceToHost, q->copy_stream));
    HIP_TRY(hipMemcpyAsync(s.h_cnt, s.d_cnt, cnt_bytes, hipMemcpyDeviceToHost, q->copy_stream));
    HIP_TRY(hipEventRecord(s.copied, q->copy_stream));
    s.width = f->width; s.rows = rows;
    q->submitted++;
    return EU_OK;
}

extern "C" int eu_sequence_next(eu_sequence *q, const uint8_t **rgb_host, uint32_t *width, uint32_t *rows, eu_stats *stats) {
    if (!q || !rgb_host) return EU_ERR_INVALID_ARGUMENT;
    eu_renderer *r = q->r;
    if (q->taken == q->submitted) { r->err = "no frame in flight"; return EU_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(r->device));
    eu_sequence::Slot &s = q->slots[q->taken % q->slots.size()];
    HIP_TRY(hipEventSynchronize(s.copied));
    q->taken++;
    *rgb_host = s.h_rgb;
    if (width) *width = s.width;
    if (rows) *rows = s.rows;
    unsigned long long overflow = 0;
    if (s.cnt_rows) ts_sum_rows((const unsigned long long *)s.h_cnt, s.cnt_rows, stats, &overflow);
    else {
        const EuDevCounters *hc = (const EuDevCounters *)s.h_cnt;
        if (stats) { stats->rays = hc->rays; stats->bg_samples = hc->bg_samples; stats->nan_pixels = hc->nan_pixels; stats->errors = hc->errors; }
        overflow = hc->overflow;
    }
    if (overflow) { r->err = "tree-node pool / ray queue overflow: raise EU_TS_NODE_FACTOR (EU_WF_RAY_FACTOR) or render in row tiles"; return EU_ERR_CAPACITY; }
    return EU_OK;
}


/* ------------------------------------------------------------------ one frame across the GPUs of a node, from one process
 * (scope row e; BASELINE config 5).  The frame's rows are cut into 8-row strips dealt round-robin over the devices
 * (eu_frame.strip_*): glass objects, which cost hundreds of rays per pixel, are spread over all GPUs.  Every device traces
 * its strips and packs them to RGB8 on its own stream; the packed strips travel to the root device over xGMI
 * (hipMemcpyPeerAsync on the sending device's stream: one transfer per device per frame, landing in place in one buffer);
 * a kernel on the root restores row order = the reference's RawImage2d (universe/mod.rs:351-356).  No other exchange. */
__global__ void eu_restore_rows_kernel(const uint8_t *__restrict__ gathered, uint8_t *__restrict__ out, uint32_t width, uint32_t row_begin, uint32_t rows,
                                       uint32_t n_dev, size_t dev_stride) {
    /* output row y (of the rows [row_begin, row_begin + rows)): strip s = y / 8 belongs to device s % n at local row (s / n) * 8 + y % 8 */
    const uint32_t y = blockIdx.y;
    if (y >= rows) return;
    const uint32_t s8 = y >> 3, k = s8 % n_dev, local = (s8 / n_dev) * 8 + (y & 7);
    const size_t row_bytes = (size_t)width * 3;
    const uint8_t *src = gathered + (size_t)k * dev_stride + (size_t)local * row_bytes;
    uint8_t *dst = out + (size_t)y * row_bytes;
    for (size_t b = (size_t)blockIdx.x * blockDim.x + threadIdx.x; b < row_bytes; b += (size_t)gridDim.x * blockDim.x) dst[b] = src[b];
    (void)row_begin;
}

struct eu_multi {
    std::vector<eu_renderer *> r;            /* r[k] on devices[k]; r[0] is the root */
    std::vector<int> devices;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> sent;
    std::vector<uint32_t *> d_rgba;          /* per device: its strips, RGBA8 */
    std::vector<uint8_t *> d_rgb;            /* per device (k > 0): its strips packed to RGB8 */
    std::vector<size_t> cap_pixels;
    uint8_t *d_gathered = nullptr;           /* root: n * max_rows * width * 3 */
    uint8_t *d_out = nullptr;                /* root: rows * width * 3, frame order */
    size_t gathered_bytes = 0, out_bytes = 0;
    std::string err;
};

extern "C" void eu_multi_destroy(eu_multi *m) {
    if (!m) return;
    for (size_t k = 0; k < m->r.size(); k++) {
        (void)hipSetDevice(m->devices[k]);
        if (k < m->stream.size() && m->stream[k]) { (void)hipStreamSynchronize(m->stream[k]); (void)hipStreamDestroy(m->stream[k]); }
        if (k < m->sent.size() && m->sent[k]) (void)hipEventDestroy(m->sent[k]);
        if (k < m->d_rgba.size() && m->d_rgba[k]) (void)hipFree(m->d_rgba[k]);
        if (k < m->d_rgb.size() && m->d_rgb[k]) (void)hipFree(m->d_rgb[k]);
        if (m->r[k]) eu_renderer_destroy(m->r[k]);
    }
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    if (m->d_gathered) (void)hipFree(m->d_gathered);
    if (m->d_out) (void)hipFree(m->d_out);
    delete m;
}

extern "C" int eu_multi_create(const eu_scene *scene, const int *devices, int n_devices, eu_multi **out, char *err, size_t errlen) {
    if (!scene || !devices || !out || n_devices < 1 || n_devices > 64) return EU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    eu_multi *m = new eu_multi();
    m->devices.assign(devices, devices + n_devices);
    m->r.assign(n_devices, nullptr); m->stream.assign(n_devices, nullptr); m->sent.assign(n_devices, nullptr);
    m->d_rgba.assign(n_devices, nullptr); m->d_rgb.assign(n_devices, nullptr); m->cap_pixels.assign(n_devices, 0);
    for (int k = 0; k < n_devices; k++) {
        int rc = eu_renderer_create(scene, devices[k], &m->r[k], err, errlen);
        if (rc != EU_OK) { eu_multi_destroy(m); return rc; }
        hipError_t e = hipSetDevice(devices[k]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->sent[k], hipEventDisableTiming);
        if (e != hipSuccess) { set_err(err, errlen, std::string("eu_multi_create: ") + hipGetErrorString(e)); eu_multi_destroy(m); return EU_ERR_HIP; }
        if (k > 0 && devices[k] != devices[0]) {      /* direct xGMI transfers where the topology allows them (otherwise the runtime stages the copy) */
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, devices[k], devices[0]) == hipSuccess && can) {
                hipError_t pe = hipDeviceEnablePeerAccess(devices[0], 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
        }
    }
    *out = m;
    return EU_OK;
}

#define MULTI_TRY(expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            m->err = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            return EU_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

extern "C" int eu_render_multi(eu_multi *m, const eu_camera *cam, const eu_frame *f, uint8_t *rgb_host, void **rgb_dev_root, eu_stats *stats) {
    if (!m || !cam || !f) return EU_ERR_INVALID_ARGUMENT;
    if (f->width == 0 || f->height == 0 || f->row_begin > f->row_end || f->row_end > f->height || f->strip_count > 1) return EU_ERR_INVALID_ARGUMENT;
    const uint32_t n = (uint32_t)m->r.size(), rows = f->row_end - f->row_begin, W = f->width;
    if (stats) memset(stats, 0, sizeof *stats);
    if (rows == 0) return EU_OK;
    const size_t row_bytes = (size_t)W * 3;
    uint32_t max_rows = 0;
    std::vector<eu_frame> fr(n, *f);
    std::vector<uint32_t> lrows(n, 0);
    for (uint32_t k = 0; k < n; k++) {
        fr[k].strip_count = n; fr[k].strip_index = k;
        lrows[k] = n > 1 ? eu_frame_local_rows(&fr[k]) : rows;
        if (n == 1) { fr[k].strip_count = 0; fr[k].strip_index = 0; }
        if (lrows[k] > max_rows) max_rows = lrows[k];
    }
    const size_t dev_stride = (size_t)max_rows * row_bytes;
    /* buffers, grown on demand */
    MULTI_TRY(hipSetDevice(m->devices[0]));
    if (m->gathered_bytes < dev_stride * n + 16) {
        if (m->d_gathered) (void)hipFree(m->d_gathered);
        m->d_gathered = nullptr; m->gathered_bytes = 0;
        MULTI_TRY(hipMalloc((void **)&m->d_gathered, dev_stride * n + 16));
        m->gathered_bytes = dev_stride * n + 16;
    }
    if (m->out_bytes < (size_t)rows * row_bytes + 16) {
        if (m->d_out) (void)hipFree(m->d_out);
        m->d_out = nullptr; m->out_bytes = 0;
        MULTI_TRY(hipMalloc((void **)&m->d_out, (size_t)rows * row_bytes + 16));
        m->out_bytes = (size_t)rows * row_bytes + 16;
    }
    for (uint32_t k = 0; k < n; k++) {
        const size_t pixels = (size_t)max_rows * W;
        if (m->cap_pixels[k] < pixels) {
            MULTI_TRY(hipSetDevice(m->devices[k]));
            if (m->d_rgba[k]) (void)hipFree(m->d_rgba[k]);
            if (m->d_rgb[k]) (void)hipFree(m->d_rgb[k]);
            m->d_rgba[k] = nullptr; m->d_rgb[k] = nullptr; m->cap_pixels[k] = 0;
            MULTI_TRY(hipMalloc((void **)&m->d_rgba[k], pixels * 4));
            if (k > 0) MULTI_TRY(hipMalloc((void **)&m->d_rgb[k], pixels * 3 + 16));
            m->cap_pixels[k] = pixels;
        }
    }
    /* every renderer's work buffers, streams and events first: no allocation (which may synchronise or clear memory) happens
     * once the first device's kernels are in flight */
    for (uint32_t k = 0; k < n; k++) {
        if (lrows[k] == 0) continue;
        MULTI_TRY(hipSetDevice(m->devices[k]));
        m->r[k]->prepare_only = true;
        const int prc = render_device_impl(m->r[k], cam, &fr[k], m->stream[k], m->d_rgba[k], nullptr, nullptr);
        m->r[k]->prepare_only = false;
        if (prc != EU_OK) { m->err = m->r[k]->err; return prc; }
    }
    /* trace everywhere ... */
    for (uint32_t k = 0; k < n; k++) {
        if (lrows[k] == 0) continue;
        MULTI_TRY(hipSetDevice(m->devices[k]));
        int rc = render_device_impl(m->r[k], cam, &fr[k], m->stream[k], m->d_rgba[k], nullptr, nullptr);
        if (rc != EU_OK) { m->err = m->r[k]->err; return rc; }
    }
    /* ... then pack, and one transfer per device into its slot of the root's buffer */
    for (uint32_t k = 0; k < n; k++) {
        if (lrows[k] == 0) continue;
        MULTI_TRY(hipSetDevice(m->devices[k]));
        uint8_t *packed = k == 0 ? m->d_gathered : m->d_rgb[k];      /* the root packs straight into slot 0 */
        int rc = eu_pack_rgb_device(m->r[k], m->d_rgba[k], packed, (size_t)lrows[k] * W, m->stream[k]);
        if (rc != EU_OK) { m->err = m->r[k]->err; return rc; }
        if (k > 0) MULTI_TRY(hipMemcpyPeerAsync(m->d_gathered + (size_t)k * dev_stride, m->devices[0], packed, m->devices[k], (size_t)lrows[k] * row_bytes, m->stream[k]));
        MULTI_TRY(hipEventRecord(m->sent[k], m->stream[k]));
    }
    MULTI_TRY(hipSetDevice(m->devices[0]));
    for (uint32_t k = 1; k < n; k++) if (lrows[k]) MULTI_TRY(hipStreamWaitEvent(m->stream[0], m->sent[k], 0));
    if (n > 1) {
        unsigned gx = (unsigned)((row_bytes + 255) / 256); if (gx > 64) gx = 64;
        hipLaunchKernelGGL(eu_restore_rows_kernel, dim3(gx, rows), dim3(256), 0, m->stream[0], m->d_gathered, m->d_out, W, f->row_begin, rows, n, dev_stride);
        MULTI_TRY(hipGetLastError());
    }
    uint8_t *result = n > 1 ? m->d_out : m->d_gathered;
    if (rgb_host) MULTI_TRY(hipMemcpyAsync(rgb_host, result, (size_t)rows * row_bytes, hipMemcpyDeviceToHost, m->stream[0]));
    MULTI_TRY(hipStreamSynchronize(m->stream[0]));
    if (rgb_dev_root) *rgb_dev_root = result;
    int worst = EU_OK;
    for (uint32_t k = 0; k < n; k++) {
        if (lrows[k] == 0) continue;
        eu_stats st;
        const int rc = eu_renderer_stats(m->r[k], &st);
        if (rc != EU_OK) { m->err = m->r[k]->err; worst = rc; continue; }
        if (stats) { stats->rays += st.rays; stats->bg_samples += st.bg_samples; stats->nan_pixels += st.nan_pixels; stats->errors += st.errors; }
    }
    return worst;
}

extern "C" const char *eu_multi_error(const eu_multi *m) { return m ? m->err.c_str() : ""; }

/* Universe::trace_path_unknown (universe/mod.rs:273-286) on the resident scene: one lane, synchronous. */
extern "C" int eu_trace_path(eu_renderer *r, const eu_f64 location[4], const eu_f64 direction[4], eu_f64 distance,
                             eu_f64 out_location[4], eu_f64 out_direction[4], int32_t *found) {
    if (!r || !location || !direction || !out_location || !out_direction || !found) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    if (r->hit_cap > 96) { r->err = "scene needs a deeper hit stack than the path kernel has (96)"; return EU_ERR_CAPACITY; }
    if (!r->d_path_in) {
        HIP_TRY(hipMalloc((void **)&r->d_path_in, 9 * sizeof(double)));
        HIP_TRY(hipMalloc((void **)&r->d_path_out, sizeof(EuPathResult)));
    }
    const int D = r->dim;
    double in[9];
    for (int k = 0; k < D; k++) { in[k] = location[k]; in[D + k] = direction[k]; }
    in[2 * D] = distance;
    HIP_TRY(hipMemcpy(r->d_path_in, in, sizeof in, hipMemcpyHostToDevice));
    if (D == 3) hipLaunchKernelGGL(eu_trace_path_kernel<3>, dim3(1), dim3(64), 0, nullptr, r->d_scene, r->d_path_in, r->d_path_out);
    else hipLaunchKernelGGL(eu_trace_path_kernel<4>, dim3(1), dim3(64), 0, nullptr, r->d_scene, r->d_path_in, r->d_path_out);
    HIP_TRY(hipGetLastError());
    EuPathResult res;
    HIP_TRY(hipMemcpy(&res, r->d_path_out, sizeof res, hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; k++) { out_location[k] = k < D ? res.location[k] : R(0.0); out_direction[k] = k < D ? res.direction[k] : R(0.0); }
    if (res.found < 0) { *found = 0; r->err = "trace_path: more than 4096 surface crossings"; return EU_ERR_PATH_STEPS; }
    *found = res.found;
    return EU_OK;
}

/* Camera::update (Environment::update, universe/mod.rs:399-405): rotation on the host (camera_host.cpp),
 * translation through eu_trace_path. */
extern "C" int eu_camera_update(eu_renderer *r, eu_camera *cam, const eu_input *in) {
    if (!cam || !in) return EU_ERR_INVALID_ARGUMENT;
    if (r && cam->dim != r->dim) { r->err = "camera dimension does not match the scene"; return EU_ERR_INVALID_ARGUMENT; }
    euclider::TracePathFn fn;
    if (r) fn = [r](const eu_f64 *loc, const eu_f64 *dir, eu_f64 dist, eu_f64 *ol, eu_f64 *od) -> int {
        eu_f64 l4[4] = {0, 0, 0, 0}, d4[4] = {0, 0, 0, 0};
        for (int k = 0; k < r->dim; k++) { l4[k] = loc[k]; d4[k] = dir[k]; }
        int32_t found = 0;
        const int rc = eu_trace_path(r, l4, d4, dist, ol, od, &found);
        return rc != EU_OK ? rc : (int)found;
    };
    return euclider::camera_update(cam, in, fn);
}

extern "C" int eu_selftest_math(int device, int fn, const eu_f64 *x, const eu_f64 *y, eu_f64 *out, size_t n) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device < 0 || device >= cnt) return EU_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return EU_ERR_HIP;
    if (n == 0) return EU_OK;
    eu_f64 *dx = nullptr, *dy = nullptr, *dout = nullptr;
    int rc = EU_ERR_HIP;
    if (hipMalloc((void **)&dx, n * 8) == hipSuccess && hipMalloc((void **)&dout, n * 8) == hipSuccess && (!y || hipMalloc((void **)&dy, n * 8) == hipSuccess)) {
        if (hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice) == hipSuccess && (!y || hipMemcpy(dy, y, n * 8, hipMemcpyHostToDevice) == hipSuccess)) {
            hipLaunchKernelGGL(eu_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, fn, dx, dy, dout, n);
            if (hipGetLastError() == hipSuccess && hipMemcpy(out, dout, n * 8, hipMemcpyDeviceToHost) == hipSuccess) rc = EU_OK;
        }
    }
    if (dx) (void)hipFree(dx);
    if (dy) (void)hipFree(dy);
    if (dout) (void)hipFree(dout);
    return rc;
}
