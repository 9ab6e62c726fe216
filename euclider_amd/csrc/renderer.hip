/*
 * renderer.hip -- the trace kernels for gfx950 and the host-side renderer behind the C ABI.
 *
 * A frame (Environment::render, /root/reference/src/universe/mod.rs:300-357: one thread-pool job per pixel) is traced by the
 * wavefront pipeline of trace_wavefront.h -- per generation of the recursion one intersect and one shade launch over the rays
 * of that generation, then one resolve launch per generation bottom-up -- either with the ahead-of-time kernels that
 * interpret the flat scene, or with kernels specialised for the scene and compiled when the renderer is created (jit.cpp).
 * The persistent stack-based kernel (trace_megakernel.h) re-traces frames whose recursion overflows the ray queues.
 * Each pixel is written once as RGBA8 (one dword store); textures are read-only RGBA8 in HBM.
 * No MFMA: there is no dense contraction anywhere in this path; it is f64 VALU + divergence bound.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/euclider_amd.h"
#include "camera_host.hpp"      /* Camera::update stays f64 in every build: before eu_real.h */
#include "scene_host.hpp"       /* ends with eu_real.h: from here on `double` is the path's F */
#include "trace_device.h"
#include "trace_megakernel.h"
#include "trace_wavefront.h"
#include "trace_path.h"
#include "jit.hpp"

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

/* start of a frame: the frame's counters and the work counters (trace_wavefront.h: WfDeal) of the bands and generations it will use --
 * only the counter words themselves, each of which sits on a cache line of its own */
__global__ void eu_wf_clear_kernel(EuDevCounters *counters, uint32_t *work, uint32_t words_per_band, uint32_t n_bands, uint32_t n_gen) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    for (uint32_t i = t; i < sizeof(EuDevCounters) / 8; i += nt) ((unsigned long long *)counters)[i] = 0ull;
    const uint32_t per_gen = EU_WORK_SLOTS, per_band = n_gen * per_gen;
    for (uint32_t i = t; i < n_bands * per_band; i += nt) {
        const uint32_t band = i / per_band, r = i % per_band;
        work[(size_t)band * words_per_band + (size_t)(r / per_gen) * EU_WORK_PER_GEN + (size_t)(r % per_gen) * EU_WORK_STRIDE] = 0u;
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
    eu_renderer_opts opts = {};
    std::string cache_dir, jit_flags;
    int device = 0;
    int dim = 3;
    uint32_t hit_cap = 0, hit_cap_strict = 0;      /* hit-stack entries the wavefront kernels reserve / the strict worst case of the stack kernels (scene_host.cpp: HitUse) */
    uint32_t color_depth = 0;
    uint32_t scene_words = 0;
    uint64_t *d_scene = nullptr;
    std::vector<void *> d_textures;
    EuDevCounters *d_counters = nullptr;
    uint32_t *d_rgba = nullptr; size_t rgba_pixels = 0;      /* internal frame buffers for eu_render */
    uint8_t *d_rgb = nullptr;
    eu_f64 *d_hit = nullptr;
    eu_f64 *d_point = nullptr;
    real *d_path_in = nullptr;              /* eu_trace_path: location, direction, distance */
    EuPathResult *d_path_out = nullptr;
    static constexpr int EV_RING = 64;        /* per-launch HIP event pairs, on the launch stream */
    hipEvent_t ev_start[EV_RING] = {}, ev_stop[EV_RING] = {};
    unsigned long long launches = 0;
    hipStream_t last_stream = nullptr;
    int num_cus = 0;
    bool scene_in_lds = true;
    /* which kernels trace a frame: the generation-synchronous wavefront pipeline (trace_wavefront.h) or the persistent stack-based
     * kernel (trace_megakernel.h: O(depth) memory per lane whatever the fan-out; the re-trace path of frames that overflow the queues) */
    bool use_wavefront = true;
    uint64_t retraces = 0;                   /* frames traced a second time (ray-queue overflow): eu_renderer_retraces */
    /* scene-specialised kernels (jit.hpp), when the renderer was created with EU_SPECIALIZE_SYNC and the compilation succeeded */
    hipModule_t jit_module = nullptr;
    hipFunction_t jit_intersect = nullptr, jit_shade = nullptr, jit_intersect0 = nullptr, jit_shade0 = nullptr;      /* ...0: generation 0 */
    hipFunction_t jit_fshade = nullptr, jit_fshade0 = nullptr;      /* the fused forms: shade, then intersect the rays just queued */
    bool fuse = true;                        /* a generation is ONE launch (fused shade + intersect) where such kernels exist (eu_renderer_opts.flags: EU_RENDERER_NO_FUSE) */
    bool jit_hs_lds = true;
    bool jit_color_stack = false;      /* the specialised shade kernels evaluate some surfaces from their records: colour-operand stack in dynamic LDS */
    uint32_t jit_hs_cap = 0;
    eu_jit_info jit = {};
    std::string jit_log;
    std::shared_ptr<euclider::JitJob> jit_job;      /* EU_SPECIALIZE_ASYNC: the compilation in flight, polled when a frame is launched */
    /* wavefront pipeline buffers (HBM), sized for the largest frame seen so far */
    static constexpr int WF_MAX_STREAMS = 8;
    static constexpr size_t kCountersPad = (sizeof(EuDevCounters) + 255) & ~(size_t)255;
    static constexpr size_t kWorkWordsPerBand = (size_t)(EU_MAX_DEPTH + 1) * EU_WORK_PER_GEN;
    static constexpr size_t counters_bytes(int bands) { return kCountersPad + (size_t)bands * kWorkWordsPerBand * 4 + kCountersPad; }      /* (+ a scratch EuDevCounters behind the last band) */
    uint32_t *work_of(int band) const { return (uint32_t *)((char *)d_counters + kCountersPad) + (size_t)band * kWorkWordsPerBand; }
    EuWfBuffers wf[WF_MAX_STREAMS] = {};     /* band pipelines run concurrently on side streams */
    size_t wf_pixels = 0;
    uint32_t wf_depth = 0;                   /* deepest max_depth the node slots are sized for */
    int wf_sets = 0;                         /* buffer sets allocated */
    bool prepare_only = false;               /* render_device_impl: size the work buffers for the frame, launch nothing */
    hipStream_t wf_stream[WF_MAX_STREAMS] = {};
    hipEvent_t wf_fork = nullptr, wf_join[WF_MAX_STREAMS] = {};
    uint32_t wf_seg_per_cu = 3;              /* producer (shade) workgroups per CU = queue segments per CU: 3 for the interpreter's shade kernel (168 VGPRs),
                                              * what the specialised one's registers allow (up to 4) */
    int wf_n_streams = 2;
    bool wf_streams_auto = true;             /* no number from the caller: one band or three, by how many rays per pixel the frames turn out to have */
    unsigned long long *h_rays_sample = nullptr;      /* pinned: the ray count of a recent frame, copied behind its launches every few frames */
    uint64_t sample_pixels = 0;
    real wf_ray_factor = R(4.0);
    uint64_t wf_band_pixels = 4u << 20;      /* pixels traced per wavefront pass */
    uint64_t wf_split_pixels = 1u << 19;     /* frames of at least this many pixels are cut into wf_n_streams concurrent bands */
    uint32_t wf_permille = 0;                /* the share of a full-chip grid the buffers' queue segments were cut for */
    unsigned long long *d_prof = nullptr;    /* diagnostics: per-workgroup time stamps of kernels built with -DEU_PROFILE_WG */
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

static void fill_jit_info(const euclider::JitPlan &plan, const euclider::JitBuild &b, bool active, eu_jit_info &info) {
    info.requested = 1;
    info.active = active ? 1 : 0;
    info.from_cache = b.from_cache ? 1 : 0;
    info.hit_stack_entries = plan.hs_cap;
    info.compile_ms = b.compile_ms;
    snprintf(info.key, sizeof info.key, "%s", plan.key.c_str());
}

/* load a specialised code object and switch the renderer to its kernels */
static void renderer_load_jit(eu_renderer *r, const euclider::JitPlan &plan, const euclider::JitBuild &b) {
    fill_jit_info(plan, b, false, r->jit);
    if (hipModuleLoadData(&r->jit_module, b.code.data()) != hipSuccess) { (void)hipGetLastError(); r->jit_module = nullptr; r->jit_log += "\nhipModuleLoadData failed"; return; }
    bool ok = hipModuleGetFunction(&r->jit_intersect0, r->jit_module, "eu_jit_intersect0") == hipSuccess;
    if (plan.fused) ok = ok && hipModuleGetFunction(&r->jit_fshade, r->jit_module, "eu_jit_fshade") == hipSuccess &&
                         hipModuleGetFunction(&r->jit_fshade0, r->jit_module, "eu_jit_fshade0") == hipSuccess;
    else ok = ok && hipModuleGetFunction(&r->jit_intersect, r->jit_module, "eu_jit_intersect") == hipSuccess &&
              hipModuleGetFunction(&r->jit_shade, r->jit_module, "eu_jit_shade") == hipSuccess &&
              hipModuleGetFunction(&r->jit_shade0, r->jit_module, "eu_jit_shade0") == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        (void)hipModuleUnload(r->jit_module);
        r->jit_module = nullptr; r->jit_intersect = nullptr; r->jit_shade = nullptr; r->jit_intersect0 = nullptr; r->jit_shade0 = nullptr; r->jit_fshade = nullptr; r->jit_fshade0 = nullptr;
        r->jit_log += "\nthe code object lacks the kernels";
        return;
    }
    r->jit_hs_lds = plan.hs_lds;
    r->jit_hs_cap = plan.hs_cap;
    r->jit.active = 1;
    /* producer workgroups per CU = what the producer kernels' registers and LDS allow.  (A renderer that switches kernels between two
     * frames -- EU_SPECIALIZE_ASYNC -- keeps its buffers: wf_launch_frame uses at most the segments they were cut into.) */
    const bool fused = r->jit_fshade != nullptr;
    r->jit_color_stack = plan.color_stack;
    const size_t color_lds = plan.color_stack ? (size_t)(r->color_depth ? r->color_depth : 1u) * 4 * sizeof(real) * EU_WF_BLOCK : 0;      /* (surfaces beyond the generator's budget: jit.hpp) */
    const size_t hs_lds_bytes = fused && plan.hs_lds ? (size_t)(EU_WF_BLOCK / 64) * plan.hs_cap * 64 * (sizeof(real) + 4) : 0;
    const size_t dyn = hs_lds_bytes > color_lds ? hs_lds_bytes : color_lds;
    if (dyn > 64 * 1024) {      /* beyond the default limit of dynamic LDS per workgroup */
        (void)hipFuncSetAttribute((const void *)r->jit_fshade, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        (void)hipFuncSetAttribute((const void *)r->jit_fshade0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        (void)hipGetLastError();
    }
    int occ = 0, occ0 = 0;
    if (hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fused ? r->jit_fshade : r->jit_shade, EU_WF_BLOCK, dyn) == hipSuccess &&
        hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, fused ? r->jit_fshade0 : r->jit_shade0, EU_WF_BLOCK, dyn) == hipSuccess) {
        const int o = occ < occ0 ? occ : occ0;
        r->wf_seg_per_cu = o >= 4 ? 4u : (o >= 3 ? 3u : (o >= 2 ? 2u : 1u));
    } else (void)hipGetLastError();
}

/* EU_SPECIALIZE_SYNC: generate, compile (or fetch) and load this scene's kernels; any failure leaves the interpreter kernels in charge.
 * EU_SPECIALIZE_ASYNC: a code object already in a cache is loaded at once, otherwise the compilation goes to the library's worker thread
 * and renderer_poll_jit switches over when it is done. */
static void renderer_attach_jit(eu_renderer *r, bool async) {
    r->jit.requested = 1;
    const euclider::JitPlan plan = euclider::jit_generate(*r->flat, r->jit_flags, r->fuse);
    euclider::JitBuild b;
    const int rc = euclider::jit_build(plan, r->cache_dir, b, async);
    r->jit_log = b.log;
    fill_jit_info(plan, b, false, r->jit);
    if (rc == EU_OK) {
        renderer_load_jit(r, plan, b);
        if (!r->jit_module && b.from_cache && !async) {      /* a cached code object the runtime refused (truncated, foreign): forget it and compile once */
            euclider::jit_forget(plan, r->cache_dir);
            euclider::JitBuild b2;
            if (euclider::jit_build(plan, r->cache_dir, b2, false) == EU_OK) { r->jit_log = b2.log; renderer_load_jit(r, plan, b2); }
        }
        return;
    }
    if (async && rc == EU_ERR_BUSY) r->jit_job = euclider::jit_submit(r->flat, r->cache_dir, r->jit_flags, plan.key, r->fuse);
}

static void renderer_poll_jit(eu_renderer *r) {
    if (!r->jit_job || !r->jit_job->done.load(std::memory_order_acquire)) return;
    std::shared_ptr<euclider::JitJob> job = r->jit_job;
    r->jit_job.reset();
    r->jit_log = job->build.log;
    if (job->rc != EU_OK) { fill_jit_info(job->plan, job->build, false, r->jit); return; }
    renderer_load_jit(r, job->plan, job->build);      /* (frames in flight keep running the kernels they were launched with; the buffers stay) */
}

/* interpreter kernels: producer workgroups per CU from the occupancy of the shade kernel that will be launched (fused or not) */
template <int D> static void interp_seg_per_cu(eu_renderer *r) {
    const uint32_t hs_cap = r->hit_cap < 8 ? 8u : ((r->hit_cap + 3u) & ~3u);
    const bool hs_lds = hs_cap <= 24;
    const size_t isect_lds = hs_lds ? (size_t)(EU_WF_BLOCK / 64) * hs_cap * 64 * (sizeof(real) + 4) : 0;
    const size_t color_lds = (size_t)(r->color_depth ? r->color_depth : 1u) * 4 * sizeof(real) * EU_WF_BLOCK;
    const bool shade_lds = (size_t)r->scene_words * 8 + color_lds <= 44 * 1024 && !(r->opts.flags & EU_RENDERER_SHADE_SCENE_GLOBAL);
    const size_t shade_dyn = shade_lds ? (size_t)r->scene_words * 8 + color_lds : color_lds;
    const bool fuse = r->fuse && hs_lds && shade_lds;
    const size_t dyn = fuse ? (shade_dyn > isect_lds ? shade_dyn : isect_lds) : shade_dyn;
    int occ = 0, occ0 = 0;
    hipError_t e;
    if (fuse) {
        if (dyn > 64 * 1024) {
            (void)hipFuncSetAttribute((const void *)eu_wf_fshade_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
            (void)hipFuncSetAttribute((const void *)eu_wf_fshade0_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
        }
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, eu_wf_fshade_kernel<D, true>, EU_WF_BLOCK, dyn);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, eu_wf_fshade0_kernel<D, true>, EU_WF_BLOCK, dyn);
    } else if (shade_lds) {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, eu_wf_shade_kernel<D, true>, EU_WF_BLOCK, dyn);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, eu_wf_shade0_kernel<D, true>, EU_WF_BLOCK, dyn);
    } else {
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, eu_wf_shade_kernel<D, false>, EU_WF_BLOCK, dyn);
        if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ0, eu_wf_shade0_kernel<D, false>, EU_WF_BLOCK, dyn);
    }
    if (e != hipSuccess) { (void)hipGetLastError(); return; }
    const int o = occ < occ0 ? occ : occ0;
    r->wf_seg_per_cu = o >= 3 ? 3u : (o >= 2 ? 2u : 1u);
}

static int renderer_create_impl(std::shared_ptr<const euclider::FlatScene> flat_scene, int device, const eu_renderer_opts *opts_in, eu_renderer **out, char *err, size_t errlen) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        set_err(err, errlen, "no usable HIP device (this library has no CPU fallback)");
        return EU_ERR_NO_DEVICE;
    }
    eu_renderer *r = new eu_renderer();
    r->flat = flat_scene;
    r->device = device;
    if (opts_in) {      /* the caller's struct may be shorter (older) or longer (newer) than ours */
        size_t nbytes = opts_in->struct_size ? opts_in->struct_size : sizeof(eu_renderer_opts);
        if (nbytes > sizeof(eu_renderer_opts)) nbytes = sizeof(eu_renderer_opts);
        memcpy(&r->opts, opts_in, nbytes);
        if (nbytes > offsetof(eu_renderer_opts, cache_dir) && r->opts.cache_dir) r->cache_dir = r->opts.cache_dir;
        if (nbytes > offsetof(eu_renderer_opts, jit_flags) && r->opts.jit_flags) r->jit_flags = r->opts.jit_flags;
        r->opts.cache_dir = nullptr; r->opts.jit_flags = nullptr;      /* (the caller's strings need not outlive this call) */
    }
    const EuFlatHeader &h = flat_scene->header();
    r->dim = (int)h.dim;
    r->hit_cap = h.hit_cap & 0xffffu;
    r->hit_cap_strict = h.hit_cap >> 16;
    r->color_depth = h.color_depth;
    auto failhip = [&](int code) { set_err(err, errlen, r->err); eu_renderer_destroy(r); return code; };
    if (r->hit_cap_strict > 96) { r->err = "scene needs a per-ray hit stack of " + std::to_string(r->hit_cap_strict) + " entries (compiled maximum 96)"; return failhip(EU_ERR_CAPACITY); }
    if (h.list_depth > 8 || h.color_depth > 4 || h.rpn_depth > 8) { r->err = "scene exceeds a compiled stack depth (csg lists 8, colour 4, rpn 8)"; return failhip(EU_ERR_CAPACITY); }
    if (r->opts.kernel > EU_KERNEL_STACK || r->opts.specialize > EU_SPECIALIZE_ASYNC || r->opts.streams > (uint32_t)eu_renderer::WF_MAX_STREAMS ||
        !(r->opts.ray_factor >= 0.0)) { r->err = "bad eu_renderer_opts"; return failhip(EU_ERR_INVALID_ARGUMENT); }
    auto body = [&]() -> int {
        HIP_TRY(hipSetDevice(device));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        r->num_cus = prop.multiProcessorCount;
        std::vector<uint64_t> blob = flat_scene->words;
        EuFlatHeader *bh = reinterpret_cast<EuFlatHeader *>(blob.data());
        for (uint32_t m = 0; m < bh->n_mapped; m++) {
            const auto &tex = flat_scene->textures[m];
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
        r->use_wavefront = r->opts.kernel != EU_KERNEL_STACK;
        r->fuse = !(r->opts.flags & EU_RENDERER_NO_FUSE);
        /* band pipelines per frame: three for a scene whose recursion branches -- with the caller's stream they occupy the runtime's four
         * hardware queues, and their kernels cover each other's tails (round 4, fused kernels, 3d_room 1080p: 1 / 2 / 3 / 4 bands = 1.86 /
         * 1.36 / 1.28 / 1.87 ms; 8K: 2 / 3 = 16.4 / 15.1 ms); one for the others, whose frames are chains of small launches (3d_hallways
         * 0.56 / 0.54 / 0.54 ms, 4d_cylinders 0.45 / 0.46 / 0.51) */
        r->wf_n_streams = r->opts.streams ? (int)r->opts.streams : ((h.flags & 1u) ? 3 : 1);
        r->wf_streams_auto = r->opts.streams == 0;
        if (r->opts.ray_factor > 0.0) r->wf_ray_factor = (real)r->opts.ray_factor;
        if (r->opts.band_pixels) r->wf_band_pixels = r->opts.band_pixels;
        if (r->opts.split_pixels) r->wf_split_pixels = r->opts.split_pixels;
        uint32_t specialize = r->opts.specialize;
#ifdef EU_DIAGNOSTICS      /* profiling scripts only: the shipped library reads no environment variable here */
        if (const char *k = getenv("EU_KERNEL")) r->use_wavefront = std::string(k) != "mega" && std::string(k) != "stack";
        if (const char *k = getenv("EU_SPECIALIZE")) specialize = std::string(k) == "sync" ? EU_SPECIALIZE_SYNC : (std::string(k) == "async" ? EU_SPECIALIZE_ASYNC : EU_SPECIALIZE_OFF);
        if (const char *k = getenv("EU_WF_RAY_FACTOR")) r->wf_ray_factor = (real)atof(k);
        if (const char *k = getenv("EU_WF_BAND_PIXELS")) r->wf_band_pixels = strtoull(k, nullptr, 10);
        if (const char *k = getenv("EU_WF_STREAMS")) { int v = atoi(k); r->wf_n_streams = v < 1 ? 1 : (v > eu_renderer::WF_MAX_STREAMS ? eu_renderer::WF_MAX_STREAMS : v); }
#endif
        HIP_TRY(hipMalloc((void **)&r->d_scene, blob.size() * 8));
        HIP_TRY(hipMemcpy(r->d_scene, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
        /* the frame's counters and, behind them, the work counters of every band pipeline (trace_wavefront.h: WfDeal): one clear per frame */
        HIP_TRY(hipMalloc((void **)&r->d_counters, eu_renderer::counters_bytes(eu_renderer::WF_MAX_STREAMS)));
        HIP_TRY(hipMemset(r->d_counters, 0, eu_renderer::counters_bytes(eu_renderer::WF_MAX_STREAMS)));
        HIP_TRY(hipDeviceSynchronize());      /* (hipMemset is asynchronous to the host; see wf_ensure) */
        HIP_TRY(hipMalloc((void **)&r->d_point, 3 * sizeof(eu_f64)));
        HIP_TRY(hipHostMalloc((void **)&r->h_rays_sample, 16, hipHostMallocDefault));
        r->h_rays_sample[0] = 0ull; r->h_rays_sample[1] = 0ull;
        for (int i = 0; i < eu_renderer::EV_RING; i++) { HIP_TRY(hipEventCreate(&r->ev_start[i])); HIP_TRY(hipEventCreate(&r->ev_stop[i])); }
        if ((specialize == EU_SPECIALIZE_SYNC || specialize == EU_SPECIALIZE_ASYNC) && r->use_wavefront) renderer_attach_jit(r, specialize == EU_SPECIALIZE_ASYNC);
        if (r->use_wavefront && !r->jit_intersect0) { if (r->dim == 3) interp_seg_per_cu<3>(r); else interp_seg_per_cu<4>(r); }
        return EU_OK;
    };
    int rc = body();
    if (rc != EU_OK) return failhip(rc);
    *out = r;
    return EU_OK;
}

extern "C" int eu_renderer_create_opts(const eu_scene *scene, int device, const eu_renderer_opts *opts, eu_renderer **out, char *err, size_t errlen) {
    if (!scene || !out) return EU_ERR_INVALID_ARGUMENT;
    return renderer_create_impl(std::make_shared<const euclider::FlatScene>(scene->flat), device, opts, out, err, errlen);
}

extern "C" int eu_renderer_create(const eu_scene *scene, int device, eu_renderer **out, char *err, size_t errlen) {
    return eu_renderer_create_opts(scene, device, nullptr, out, err, errlen);
}

extern "C" const char *eu_renderer_error(const eu_renderer *r) { return r ? r->err.c_str() : ""; }
extern "C" const char *eu_renderer_jit_log(const eu_renderer *r) { return r ? r->jit_log.c_str() : ""; }

extern "C" int eu_renderer_jit_info(eu_renderer *r, eu_jit_info *out) {
    if (!r || !out) return EU_ERR_INVALID_ARGUMENT;
    *out = r->jit;
    return EU_OK;
}

extern "C" void eu_renderer_destroy(eu_renderer *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipDeviceSynchronize();
    if (r->jit_job) r->jit_job->waiters.fetch_sub(1);      /* (a job nobody waits for any more is dropped if its turn has not come; the worker keeps its own reference to the scene) */
    if (r->jit_module) (void)hipModuleUnload(r->jit_module);
    for (void *p : r->d_textures) (void)hipFree(p);
    for (void *p : r->wf_allocs) (void)hipFree(p);
    for (int k = 0; k < eu_renderer::WF_MAX_STREAMS; k++) { if (r->wf_stream[k]) (void)hipStreamDestroy(r->wf_stream[k]); if (r->wf_join[k]) (void)hipEventDestroy(r->wf_join[k]); }
    if (r->wf_fork) (void)hipEventDestroy(r->wf_fork);
    if (r->d_scene) (void)hipFree(r->d_scene);
    if (r->d_counters) (void)hipFree(r->d_counters);
    if (r->d_rgba) (void)hipFree(r->d_rgba);
    if (r->d_rgb) (void)hipFree(r->d_rgb);
    if (r->d_hit) (void)hipFree(r->d_hit);
    if (r->d_point) (void)hipFree(r->d_point);
    if (r->d_prof) (void)hipFree(r->d_prof);
    if (r->h_rays_sample) (void)hipHostFree(r->h_rays_sample);
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
        real cr[3];      /* from the pose already rounded to F: the reference's camera holds F values (with the ABI's f64 fields
                             * the low_precision build used to form these products in f64: one ulp off for a pose that is not f32-exact) */
        cr[0] = dc.forward[1] * dc.up[2] - dc.forward[2] * dc.up[1];
        cr[1] = dc.forward[2] * dc.up[0] - dc.forward[0] * dc.up[2];
        cr[2] = dc.forward[0] * dc.up[1] - dc.forward[1] * dc.up[0];
        real n = sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
        for (int i = 0; i < 3; i++) dc.right[i] = cr[i] / n;
    } else {        /* right = -left, d4/entity/camera.rs:167 */
        for (int i = 0; i < D; i++) dc.right[i] = -cam->left[i];
    }
    const real w = (real)f->width, h = (real)f->height;
    const real fov_rad = EU_PI_C * (real)cam->fov_deg / R(180.0);
    dc.dist = sqrt(w * w + h * h) / (R(2.0) * eu_tan(fov_rad / R(2.0)));
    dc.max_depth = cam->max_depth;
    return EU_OK;
}

template <int D, int HSCAP, bool LDS>
static hipError_t launch_trace(eu_renderer *r, hipStream_t stream, const EuDevCamera &dc, const EuDevFrame &df, uint32_t *rgba, eu_f64 *hit_t, eu_f64 *point) {
    auto kern = eu_trace_kernel<D, HSCAP, LDS>;
    const uint32_t hs_cap = HSCAP ? (uint32_t)HSCAP : (r->hit_cap_strict < 8 ? 8u : ((r->hit_cap_strict + 3u) & ~3u));
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
static int wf_ensure(eu_renderer *r, size_t pixels, size_t items, uint32_t max_depth, int n_sets, uint32_t permille) {
    if (pixels <= r->wf_pixels && max_depth <= r->wf_depth && n_sets <= r->wf_sets && items <= (size_t)r->wf[0].ray_cap && (permille == 0 || permille == r->wf_permille)) return EU_OK;
    if (permille == 0) permille = r->wf_permille ? r->wf_permille : 1000u;      /* (a single pixel takes the buffers as they are) */
    r->wf_permille = permille;
    if (pixels < r->wf_pixels) pixels = r->wf_pixels;
    if (max_depth > r->wf_depth) r->wf_depth = max_depth;
    if (n_sets > r->wf_sets) r->wf_sets = n_sets;      /* the second buffer set exists only once a frame is traced as two concurrent bands */
    HIP_TRY(hipDeviceSynchronize());
    for (void *p : r->wf_allocs) (void)hipFree(p);
    r->wf_allocs.clear();
    r->wf_pixels = 0;
    const int D = r->dim;
    for (int set = 0; set < r->wf_sets; set++) {
    EuWfBuffers &B = r->wf[set];
    memset(&B, 0, sizeof B);
    /* one queue segment per producer workgroup; a generation's queue holds n_seg * seg_cap ray slots */
    uint32_t n_seg = (uint32_t)r->num_cus * r->wf_seg_per_cu;
    if (n_seg > EU_WF_MAX_SEG) n_seg = EU_WF_MAX_SEG;
    n_seg = (uint32_t)(((uint64_t)n_seg * permille + 999) / 1000 + 7u) & ~7u;      /* one band pipeline's share of the chip (wf_launch_frame) */
    if (n_seg < 8) n_seg = 8;
    if (pixels / 16 < n_seg) n_seg = pixels / 16 < 16 ? 16u : (uint32_t)(pixels / 16);      /* tiny frames, single pixels: fewer producers, small buffers */
    size_t seg_cap = ((size_t)((real)pixels * r->wf_ray_factor) + n_seg - 1) / n_seg;
    if (seg_cap < 1024) seg_cap = 1024;        /* small frames: absorb uneven segments */
    if (seg_cap * n_seg < items) seg_cap = (items + n_seg - 1) / n_seg;      /* generation 0 keeps item v (8x8 pixel tiles, padded) in slot v */
    seg_cap = (seg_cap + 255) & ~(size_t)255;
    const size_t ray_cap = seg_cap * n_seg;
    const size_t node_cap = ray_cap * (size_t)(r->wf_depth ? r->wf_depth : 1u);      /* one slot per ray of every generation the deepest frame so far has */
    if (ray_cap > 0x7ffffff0ull || node_cap > 0xfffffff0ull) { r->err = "frame too large for 32-bit queue indices; render it in row tiles"; return EU_ERR_CAPACITY; }
    auto alloc = [&](void **p, size_t bytes) -> int {
        HIP_TRY(hipMalloc(p, bytes));
        r->wf_allocs.push_back(*p);
        return EU_OK;
    };
    int rc;
    for (int k = 0; k < 2; k++) {
        if ((rc = alloc((void **)&B.ray[k], ray_cap * 2 * D * sizeof(real)))) return rc;
        if ((rc = alloc((void **)&B.ray_pa[k], ray_cap * sizeof(uint2)))) return rc;
    }
    for (int k = 0; k < 2; k++) if ((rc = alloc((void **)&B.hit[k], ray_cap * sizeof(EuWfHit)))) return rc;
    /* node ids are static: generation g's queue slot q at g*ray_cap + q (only the slots that hold rays are ever touched) */
    if ((rc = alloc((void **)&B.nodes, node_cap * sizeof(EuTsNode)))) return rc;
    if ((rc = alloc((void **)&B.node_kind, node_cap))) return rc;
    if ((rc = alloc((void **)&B.seg_count, (size_t)(EU_MAX_DEPTH + 2) * n_seg * 4))) return rc;
    HIP_TRY(hipMemset(B.seg_count, 0, (size_t)(EU_MAX_DEPTH + 2) * n_seg * 4));
    /* hipMemset on device memory is asynchronous to the host and ordered on the NULL stream only, which the (non-blocking) trace
     * streams do not synchronise with: without this wait the clear can land after the first frame's kernels have published their
     * queue lengths (seen as whole strips of unwritten pixels, first frame of a fresh renderer, only with several hardware queues) */
    HIP_TRY(hipDeviceSynchronize());
    B.ray_cap = (uint32_t)ray_cap; B.node_cap = (uint32_t)node_cap;
    B.n_seg = n_seg; B.seg_cap = (uint32_t)seg_cap;
    B.prof = r->d_prof;
    B.work = r->work_of(set);
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
static int wf_grid_module(eu_renderer *r, hipFunction_t f, size_t lds_bytes, unsigned &grid) {
    int blocks_per_cu = 0;
    HIP_TRY(hipModuleOccupancyMaxActiveBlocksPerMultiprocessor(&blocks_per_cu, f, EU_WF_BLOCK, lds_bytes));
    if (blocks_per_cu < 1) blocks_per_cu = 1;
    grid = (unsigned)(r->num_cus * blocks_per_cu);
    return EU_OK;
}

template <int D>
static int wf_launch_frame(eu_renderer *r, hipStream_t caller_stream, const EuDevCamera &dc, const EuDevFrame &df_in, uint32_t *rgba, eu_f64 *hit_t, eu_f64 *point) {
    /* A frame is traced as BANDS of whole 8-row tiles.  Frames above `band_pixels` need them so that the queue and node buffers stay
     * bounded (a band of 4 Mpixel needs ~25 GB at depth 16; an 8K frame goes through in 8 passes); smaller frames are cut into
     * `wf_n_streams` bands that run as independent pipelines on streams of their own, so that one band's dependent launches -- each
     * ending in a tail during which most of the chip idles -- are covered by the other bands' kernels.  Every band launches full-chip
     * grids (`band_grid_permille` can shrink them): together the bands oversubscribe the chip, which is what fills the tails. */
    uint32_t band_rows = df_in.local_rows;
    int n_par = 1;                  /* band pipelines in flight */
    if (!df_in.single_pixel) {
        uint64_t rows_fit = r->wf_band_pixels / (df_in.width ? df_in.width : 1);
        rows_fit = rows_fit / 8 * 8;
        if (rows_fit < 8) rows_fit = 8;
        if (rows_fit < band_rows) band_rows = (uint32_t)rows_fit;
        if ((uint64_t)df_in.local_rows * df_in.width >= r->wf_split_pixels && r->wf_n_streams > 1) {
            uint32_t ns = (uint32_t)r->wf_n_streams;
            while (ns > 1 && (uint64_t)df_in.local_rows * df_in.width / ns < r->wf_split_pixels / 2) ns--;      /* bands stay large enough to be worth a launch chain */
            const uint32_t part = (((df_in.local_rows + ns - 1) / ns) + 7) / 8 * 8;
            if (part < band_rows) band_rows = part;
            n_par = (int)ns;
        }
    }
    const uint32_t n_bands = df_in.single_pixel ? 1u : (df_in.local_rows + band_rows - 1) / band_rows;
    if ((uint32_t)n_par > n_bands) n_par = (int)n_bands;
    const bool side_streams = n_par > 1;
    /* Concurrent bands of ONE round take the frame's 8-row groups in turn (band b: groups b, b + n, ...): contiguous halves of 3d_room
     * differed by 60 % in work (the glass objects sit in the lower half) and the lighter band's pipeline idled while the other finished
     * (profiles/r04_wg_profile_two_bands.txt).  Frames that need several rounds (bounded buffers) keep contiguous bands. */
    const bool interleaved = side_streams && n_bands == (uint32_t)n_par && !df_in.single_pixel;
    const uint32_t n_groups = (df_in.local_rows + 7) / 8;
    std::vector<uint32_t> band_begin(n_bands + 1);
    for (uint32_t b = 0; b <= n_bands; b++) { const uint64_t r0 = (uint64_t)b * band_rows; band_begin[b] = r0 < df_in.local_rows ? (uint32_t)r0 : df_in.local_rows; }
    if (interleaved) band_rows = ((n_groups + (uint32_t)n_par - 1) / (uint32_t)n_par) * 8;      /* (the largest band: what the buffers are sized for) */
    const size_t band_pixels = (size_t)band_rows * df_in.width;
    const size_t band_items = df_in.single_pixel ? 64 : (size_t)df_in.tiles_x * ((band_rows + 7) / 8) * 64;      /* generation 0: one slot per item of the band's 8x8 tiles */
    /* the share of a full-chip grid one band pipeline launches */
    uint32_t permille = 1000;
    if (side_streams && r->opts.band_grid_permille) permille = r->opts.band_grid_permille;      /* (default: every band launches full-chip grids -- measured, 3 bands of 3d_room: 1000 / 667 = 1.37 / 1.39 ms) */
    if (permille > 1000) permille = 1000;
    int rc = wf_ensure(r, df_in.single_pixel ? 64 : band_pixels, band_items, dc.max_depth, n_par, df_in.single_pixel ? 0u : permille);
    if (rc != EU_OK) return rc;
    if (r->prepare_only) return EU_OK;      /* buffers, streams and events exist now: nothing is allocated while the frame is in flight */
    const bool jit = r->jit_intersect0 != nullptr;
    uint32_t hs_cap = r->hit_cap < 8 ? 8u : ((r->hit_cap + 3u) & ~3u);
    if (jit && r->jit_hs_cap) hs_cap = r->jit_hs_cap;      /* (the same unless the plan was tuned: jit.cpp) */
    const bool hs_lds = jit ? r->jit_hs_lds : hs_cap <= 24;      /* two workgroups per CU at least; else: private (scratch) hit stack */
    const bool hs_small = !hs_lds && r->hit_cap <= 16;
    const size_t isect_lds = hs_lds ? (size_t)(EU_WF_BLOCK / 64) * hs_cap * 64 * (sizeof(real) + 4) : 0;     /* (t, code) per entry; the intersect kernel reads the scene through scalar loads */
    /* the interpreter's shade kernel: dynamic LDS = the colour-operand stack (color_depth RGBA reals per lane) and, when three
     * workgroups per CU still fit (160 KB / 3, minus ~8 KB static), a copy of the flat scene; the specialised one needs neither */
    const size_t color_lds = (size_t)(r->color_depth ? r->color_depth : 1u) * 4 * sizeof(real) * EU_WF_BLOCK;
    const bool shade_lds = (size_t)r->scene_words * 8 + color_lds <= 44 * 1024 && !(r->opts.flags & EU_RENDERER_SHADE_SCENE_GLOBAL);
    const size_t shade_dyn = jit ? (r->jit_color_stack ? color_lds : 0) : (shade_lds ? (size_t)r->scene_words * 8 + color_lds : color_lds);
    /* one launch per generation where a fused kernel exists: the specialised module's, or -- interpreter -- the ones with the hit stack
     * and the scene copy in LDS (deeper stacks and larger scenes keep the two-kernel pipeline) */
    const bool fuse = jit ? r->jit_fshade != nullptr : (r->fuse && hs_lds && shade_lds);
    const size_t fshade_dyn = fuse ? (shade_dyn > isect_lds ? shade_dyn : isect_lds) : shade_dyn;      /* (the hit stack lies over what the shading part no longer needs) */
    unsigned g_isect, g_isect0, g_res;      /* (the generation-0 kernels need more registers; their grids are sized on their own) */
    if (jit) { if ((rc = wf_grid_module(r, r->jit_intersect0, isect_lds, g_isect0)) || (rc = r->jit_intersect ? wf_grid_module(r, r->jit_intersect, isect_lds, g_isect) : ((g_isect = g_isect0), EU_OK))) return rc; }
    else if (hs_small) { if ((rc = wf_grid(r, eu_wf_intersect_kernel<D, 16>, 0, g_isect)) || (rc = wf_grid(r, eu_wf_intersect0_kernel<D, 16>, 0, g_isect0))) return rc; }
    else if (!hs_lds) { if ((rc = wf_grid(r, eu_wf_intersect_kernel<D, 96>, 0, g_isect)) || (rc = wf_grid(r, eu_wf_intersect0_kernel<D, 96>, 0, g_isect0))) return rc; }
    else if ((rc = wf_grid(r, eu_wf_intersect_kernel<D, 0>, isect_lds, g_isect)) || (rc = wf_grid(r, eu_wf_intersect0_kernel<D, 0>, isect_lds, g_isect0))) return rc;
    if ((rc = wf_grid(r, eu_wf_resolve_kernel, 0, g_res))) return rc;
    auto share = [&](unsigned g) { unsigned v = (unsigned)(((uint64_t)g * permille + 999) / 1000); v = (v + 7u) & ~7u; return v < 8u ? 8u : v; };      /* multiples of 8: every XCD gets the same number */
    g_isect = share(g_isect); g_isect0 = share(g_isect0); g_res = share(g_res);
    if (side_streams) {      /* fork: the side streams wait for everything queued on the caller's stream so far */
        HIP_TRY(hipEventRecord(r->wf_fork, caller_stream));
        for (int k = 0; k < n_par; k++) HIP_TRY(hipStreamWaitEvent(r->wf_stream[k], r->wf_fork, 0));
    }
    const uint64_t *scene = r->d_scene;
    EuDevCounters *counters = r->d_counters;
    EuDevCamera cam = dc;
    const uint32_t n_gen = dc.max_depth ? dc.max_depth : 1u;      /* depth 0: generation 0 only marks its pixels and samples the background */
    uint32_t max_depth = dc.max_depth;
    /* rounds of n_par bands; inside a round the launches are issued generation by generation across the bands, so that every band's
     * pipeline starts at once (band after band, the last band's first kernel would wait for the host to issue everything before it) */
    for (uint32_t band0 = 0; band0 < n_bands; band0 += (uint32_t)n_par) {
        const uint32_t nb = n_bands - band0 < (uint32_t)n_par ? n_bands - band0 : (uint32_t)n_par;
        EuDevFrame dfs[eu_renderer::WF_MAX_STREAMS];
        EuWfBuffers Bs[eu_renderer::WF_MAX_STREAMS];
        hipStream_t sts[eu_renderer::WF_MAX_STREAMS];
        uint32_t total0s[eu_renderer::WF_MAX_STREAMS];
        for (uint32_t k = 0; k < nb; k++) {
            EuDevFrame &df = dfs[k];
            df = df_in;
            const uint32_t row0 = band_begin[band0 + k], row1 = band_begin[band0 + k + 1];
            sts[k] = side_streams ? r->wf_stream[k] : caller_stream;
            Bs[k] = r->wf[side_streams ? k : 0];
            Bs[k].pad = band0 + k;
            {   /* producers: at most what the producer kernel's occupancy gives (the buffers may have been cut for more) */
                const unsigned want = share((unsigned)(r->num_cus * r->wf_seg_per_cu));
                if (want < Bs[k].n_seg) Bs[k].n_seg = want;
            }
            if (df.single_pixel) { df.band_row0 = 0; df.band_rows = 1; df.root_base = 0; Bs[k].npix = 1u; }
            else if (interleaved) {
                const uint32_t groups = (n_groups - k + (uint32_t)n_par - 1) / (uint32_t)n_par;      /* groups k, k + n, ... < n_groups */
                df.band_row0 = k; df.band_stride = (uint32_t)n_par;
                df.band_rows = groups * 8;
                df.root_base = 0;
                df.n_tiles = df.tiles_x * groups;
                Bs[k].npix = df.band_rows * df.width;
            } else {
                df.band_row0 = row0;
                df.band_rows = row1 - row0;
                df.root_base = row0 * df.width;
                df.n_tiles = df.tiles_x * ((df.band_rows + 7) / 8);
                Bs[k].npix = df.band_rows * df.width;
            }
            if ((size_t)df.n_tiles * 64 > (size_t)Bs[k].ray_cap) { r->err = "internal: generation 0 does not fit its queue"; return EU_ERR_CAPACITY; }
            total0s[k] = df.n_tiles * 64u;
            if (band0 > 0)      /* (the frame's first round was cleared with the counters; a cleared EuDevCounters would lose the frame's sums: the scratch block behind the last band takes that part) */
                hipLaunchKernelGGL(eu_wf_clear_kernel, dim3(1), dim3(256), 0, sts[k], (EuDevCounters *)r->work_of(eu_renderer::WF_MAX_STREAMS), Bs[k].work, (uint32_t)eu_renderer::kWorkWordsPerBand, 1u, n_gen + 1u);
        }
        for (uint32_t g = 0; g < n_gen; g++) {
            uint32_t gen = g;
            for (uint32_t k = 0; k < nb; k++) {
                EuDevFrame &df = dfs[k];
                EuWfBuffers &B = Bs[k];
                hipStream_t stream = sts[k];
                const unsigned g_prod = B.n_seg;      /* producers: one output segment per workgroup */
                real time_s = df.time_s;
                const bool tail = fuse;                        /* this launch also intersects the rays it queues (the last generation queues none) */
                const bool isect_now = !fuse || g == 0;        /* ... so only generation 0 has an intersect launch of its own */
                if (jit) {
                    if (isect_now && g == 0) {
                        void *ia[] = {(void *)&scene, (void *)&hs_cap, (void *)&cam, (void *)&df, (void *)&B, (void *)&counters, (void *)&hit_t};
                        HIP_TRY(hipModuleLaunchKernel(r->jit_intersect0, g_isect0, 1, 1, EU_WF_BLOCK, 1, 1, (unsigned)isect_lds, stream, ia, nullptr));
                    } else if (isect_now) {
                        void *ia[] = {(void *)&scene, (void *)&hs_cap, (void *)&gen, (void *)&B, (void *)&counters};
                        HIP_TRY(hipModuleLaunchKernel(r->jit_intersect, g_isect, 1, 1, EU_WF_BLOCK, 1, 1, (unsigned)isect_lds, stream, ia, nullptr));
                    }
                    if (g == 0 && tail) {
                        void *sa[] = {(void *)&scene, (void *)&r->scene_words, (void *)&hs_cap, (void *)&cam, (void *)&df, (void *)&B, (void *)&counters, (void *)&rgba, (void *)&hit_t, (void *)&point};
                        HIP_TRY(hipModuleLaunchKernel(r->jit_fshade0, g_prod, 1, 1, EU_WF_BLOCK, 1, 1, (unsigned)fshade_dyn, stream, sa, nullptr));
                    } else if (g == 0) {
                        void *sa[] = {(void *)&scene, (void *)&r->scene_words, (void *)&cam, (void *)&df, (void *)&B, (void *)&counters, (void *)&rgba, (void *)&hit_t, (void *)&point};
                        HIP_TRY(hipModuleLaunchKernel(r->jit_shade0, g_prod, 1, 1, EU_WF_BLOCK, 1, 1, (unsigned)shade_dyn, stream, sa, nullptr));
                    } else if (tail) {
                        void *sa[] = {(void *)&scene, (void *)&r->scene_words, (void *)&hs_cap, (void *)&gen, (void *)&max_depth, (void *)&time_s, (void *)&B, (void *)&counters, (void *)&rgba, (void *)&point};
                        HIP_TRY(hipModuleLaunchKernel(r->jit_fshade, g_prod, 1, 1, EU_WF_BLOCK, 1, 1, (unsigned)fshade_dyn, stream, sa, nullptr));
                    } else {
                        void *sa[] = {(void *)&scene, (void *)&r->scene_words, (void *)&gen, (void *)&max_depth, (void *)&time_s, (void *)&B, (void *)&counters, (void *)&rgba, (void *)&point};
                        HIP_TRY(hipModuleLaunchKernel(r->jit_shade, g_prod, 1, 1, EU_WF_BLOCK, 1, 1, (unsigned)shade_dyn, stream, sa, nullptr));
                    }
                } else {
                    if (isect_now && g == 0) {
                        if (hs_lds) hipLaunchKernelGGL((eu_wf_intersect0_kernel<D, 0>), dim3(g_isect0), dim3(EU_WF_BLOCK), isect_lds, stream, scene, hs_cap, cam, df, B, counters, hit_t);
                        else if (hs_small) hipLaunchKernelGGL((eu_wf_intersect0_kernel<D, 16>), dim3(g_isect0), dim3(EU_WF_BLOCK), 0, stream, scene, 16u, cam, df, B, counters, hit_t);
                        else hipLaunchKernelGGL((eu_wf_intersect0_kernel<D, 96>), dim3(g_isect0), dim3(EU_WF_BLOCK), 0, stream, scene, 96u, cam, df, B, counters, hit_t);
                    } else if (isect_now) {
                        if (hs_lds) hipLaunchKernelGGL((eu_wf_intersect_kernel<D, 0>), dim3(g_isect), dim3(EU_WF_BLOCK), isect_lds, stream, scene, hs_cap, gen, B, counters);
                        else if (hs_small) hipLaunchKernelGGL((eu_wf_intersect_kernel<D, 16>), dim3(g_isect), dim3(EU_WF_BLOCK), 0, stream, scene, 16u, gen, B, counters);
                        else hipLaunchKernelGGL((eu_wf_intersect_kernel<D, 96>), dim3(g_isect), dim3(EU_WF_BLOCK), 0, stream, scene, 96u, gen, B, counters);
                    }
                    if (g == 0 && tail) hipLaunchKernelGGL((eu_wf_fshade0_kernel<D, true>), dim3(g_prod), dim3(EU_WF_BLOCK), fshade_dyn, stream, scene, r->scene_words, hs_cap, cam, df, B, counters, rgba, hit_t, point);
                    else if (g == 0) {
                        if (shade_lds) hipLaunchKernelGGL((eu_wf_shade0_kernel<D, true>), dim3(g_prod), dim3(EU_WF_BLOCK), shade_dyn, stream, scene, r->scene_words, cam, df, B, counters, rgba, hit_t, point);
                        else hipLaunchKernelGGL((eu_wf_shade0_kernel<D, false>), dim3(g_prod), dim3(EU_WF_BLOCK), shade_dyn, stream, scene, r->scene_words, cam, df, B, counters, rgba, hit_t, point);
                    } else if (tail) hipLaunchKernelGGL((eu_wf_fshade_kernel<D, true>), dim3(g_prod), dim3(EU_WF_BLOCK), fshade_dyn, stream, scene, r->scene_words, hs_cap, gen, max_depth, time_s, B, counters, rgba, point);
                    else {
                        if (shade_lds) hipLaunchKernelGGL((eu_wf_shade_kernel<D, true>), dim3(g_prod), dim3(EU_WF_BLOCK), shade_dyn, stream, scene, r->scene_words, gen, max_depth, time_s, B, counters, rgba, point);
                        else hipLaunchKernelGGL((eu_wf_shade_kernel<D, false>), dim3(g_prod), dim3(EU_WF_BLOCK), shade_dyn, stream, scene, r->scene_words, gen, max_depth, time_s, B, counters, rgba, point);
                    }
                }
            }
        }
        /* bottom-up: the nodes that wait for two children (trace_nodes.h); a scene none of whose surfaces reflects has none */
        if (!(r->flat->header().flags & 4u))
        for (uint32_t g = n_gen; g-- > 0;)
            for (uint32_t k = 0; k < nb; k++)
                hipLaunchKernelGGL(eu_wf_resolve_kernel, dim3(g_res), dim3(EU_WF_BLOCK), 0, sts[k], g, total0s[k], Bs[k], counters, rgba, point);
        if (df_in.single_pixel) break;
    }
    if (side_streams) {      /* join */
        for (int k = 0; k < n_par; k++) { HIP_TRY(hipEventRecord(r->wf_join[k], r->wf_stream[k])); HIP_TRY(hipStreamWaitEvent(caller_stream, r->wf_join[k], 0)); }
    }
    HIP_TRY(hipGetLastError());
    return EU_OK;
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
    df.time_s = (real)f->time_ms / R(1000.0);
    if (single) { df.strip_count = 0; df.local_rows = 1; df.single_pixel = 1; df.single_x = single_x; df.single_y = f->row_begin; df.tiles_x = 1; df.n_tiles = 1; }
    if (rows == 0) return EU_OK;
    renderer_poll_jit(r);
    if (r->prepare_only) {
        if (r->use_wavefront) return (r->dim == 3) ? wf_launch_frame<3>(r, stream, dc, df, rgba, hit_t, point) : wf_launch_frame<4>(r, stream, dc, df, rgba, hit_t, point);
        return EU_OK;
    }
    /* The scene's flag says whether its recursion CAN branch; whether it does shows in the frames: 4d_cylinders' surfaces reflect (ratio 0.2)
     * but are opaque, its frames hold one ray per pixel, and three band pipelines cost it 12 % (0.51 against 0.45 ms).  Every few frames the
     * ray count is copied to pinned memory behind the frame's launches; a renderer whose caller named no number of bands reads what has
     * arrived so far -- never waiting -- and takes three bands above 2.5 rays per pixel, one below 1.6. */
    if (r->use_wavefront && r->wf_streams_auto && !single && r->sample_pixels) {
        const double rpp = (double)r->h_rays_sample[0] / (double)r->sample_pixels;
        if (r->h_rays_sample[0] != 0ull) {
            if (rpp > 2.5) r->wf_n_streams = 3;
            else if (rpp < 1.6) r->wf_n_streams = 1;
        }
    }
    {
        const uint32_t nb = r->use_wavefront ? (uint32_t)(r->wf_n_streams > 1 ? r->wf_n_streams : 1) : 0u;
        const uint32_t ng = (cam->max_depth ? cam->max_depth : 1u) + 1u;
        hipLaunchKernelGGL(eu_wf_clear_kernel, dim3(nb ? 4 : 1), dim3(256), 0, stream, r->d_counters, r->work_of(0), (uint32_t)eu_renderer::kWorkWordsPerBand, nb, ng);
    }
    const int slot = (int)(r->launches % eu_renderer::EV_RING);
    HIP_TRY(hipEventRecord(r->ev_start[slot], stream));
    hipError_t e = hipSuccess;
    if (r->use_wavefront) {
        int rc = (r->dim == 3) ? wf_launch_frame<3>(r, stream, dc, df, rgba, hit_t, point) : wf_launch_frame<4>(r, stream, dc, df, rgba, hit_t, point);
        if (rc != EU_OK) return rc;
    } else {
        /* hit stack in LDS when the scene's static bound is small (16 entries * 12 B * 256 lanes = 48 KB per block) */
        const bool hs_lds = r->hit_cap_strict <= 32 && r->scene_in_lds;
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
    if (r->use_wavefront && r->wf_streams_auto && !single && (r->launches < 2 || r->launches % 8 == 0)) {
        HIP_TRY(hipMemcpyAsync(r->h_rays_sample, &r->d_counters->rays, 8, hipMemcpyDeviceToHost, stream));
        r->sample_pixels = (uint64_t)rows * f->width;
    }
    r->launches++;
    r->last_stream = stream;
    return EU_OK;
}

/* why the wavefront pipeline could not finish a frame ("" = it could) */
static std::string unfinished_reason(const EuDevCounters &c) {
    if (c.overflow) return "ray queue overflow (" + std::to_string(c.overflow) + " rays dropped): raise eu_renderer_opts.ray_factor or render in row tiles";
    if (c.hs_full) return "hit stack full for " + std::to_string(c.hs_full) + " rays (the wavefront kernels reserve two hits per convex chain; rounding let a third through): "
                          "only the stack kernel finishes this frame -- eu_render, eu_trace_screen_point, eu_sequence_next and eu_render_multi fall back to it by themselves, "
                          "eu_render_device callers use a renderer created with EU_KERNEL_STACK";
    return std::string();
}

/* A frame whose recursion fans out beyond the queues' capacity (more than ray_factor rays per pixel in one generation) cannot be
 * finished by the wavefront pipeline; the persistent stack-based kernel needs O(depth) memory per lane whatever the fan-out, so
 * the synchronous entry points trace such a frame again with it.  Waits for the frame; *retraced says whether that happened. */
static int retrace_if_overflowed(eu_renderer *r, const eu_camera *cam, const eu_frame *f, hipStream_t stream, uint32_t *rgba, eu_f64 *hit_t, bool *retraced) {
    if (retraced) *retraced = false;
    if (!r->use_wavefront) return EU_OK;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(stream));
    EuDevCounters c;
    HIP_TRY(hipMemcpy(&c, r->d_counters, sizeof c, hipMemcpyDeviceToHost));
    if (!c.overflow && !c.hs_full) return EU_OK;
    r->use_wavefront = false;
    r->retraces++;
    const int rc = render_device_impl(r, cam, f, stream, rgba, hit_t, nullptr);
    r->use_wavefront = true;
    if (retraced) *retraced = true;
    return rc;
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
    EuDevCounters c;
    HIP_TRY(hipMemcpy(&c, r->d_counters, sizeof c, hipMemcpyDeviceToHost));
    out->rays = c.rays; out->bg_samples = c.bg_samples; out->nan_pixels = c.nan_pixels; out->errors = c.errors;
    if (c.overflow || c.hs_full) { r->err = unfinished_reason(c); return EU_ERR_CAPACITY; }
    return EU_OK;
}

extern "C" int eu_renderer_debug_phases(eu_renderer *r, unsigned long long out[16]) {
    if (!r || !out) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipStreamSynchronize(r->last_stream));
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
    for (int g = 0; g < 17 && g < EU_MAX_DEPTH + 2; g++) for (uint32_t i = 0; i < B.n_seg; i++) out[g] += h[(size_t)g * B.n_seg + i];      /* (generation 0 has no queue: out[0] stays 0) */
    return EU_OK;
}

/* Diagnostics (kernels built with -DEU_PROFILE_WG): the first call (out == NULL) switches the recording on; later calls wait for the
 * device, copy up to max_records records of four words each (trace_wavefront.h: WF_PROF_END) and start a new recording. */
extern "C" int eu_renderer_debug_wg_profile(eu_renderer *r, unsigned long long *out, size_t max_records, size_t *n_records) {
    if (!r) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    HIP_TRY(hipDeviceSynchronize());
    const size_t cap = (size_t)1 << 20;      /* EU_PROF_CAP */
    if (!r->d_prof) {
        HIP_TRY(hipMalloc((void **)&r->d_prof, (4 + 4 * cap) * 8));
        HIP_TRY(hipMemset(r->d_prof, 0, 32));
        HIP_TRY(hipDeviceSynchronize());
        for (int k = 0; k < eu_renderer::WF_MAX_STREAMS; k++) r->wf[k].prof = r->d_prof;
        if (n_records) *n_records = 0;
        return EU_OK;
    }
    unsigned long long n = 0;
    HIP_TRY(hipMemcpy(&n, r->d_prof, 8, hipMemcpyDeviceToHost));
    if (n > cap) n = cap;
    if (n > max_records) n = max_records;
    if (out && n) HIP_TRY(hipMemcpy(out, r->d_prof + 4, (size_t)n * 32, hipMemcpyDeviceToHost));
    if (n_records) *n_records = (size_t)n;
    HIP_TRY(hipMemset(r->d_prof, 0, 32));
    HIP_TRY(hipDeviceSynchronize());
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
    rc = retrace_if_overflowed(r, cam, f, nullptr, r->d_rgba, hit_t_host ? r->d_hit : nullptr, nullptr);
    if (rc != EU_OK) return rc;
    rc = eu_pack_rgb_device(r, r->d_rgba, r->d_rgb, pixels, nullptr);
    if (rc != EU_OK) return rc;
    HIP_TRY(hipMemcpy(rgb_host, r->d_rgb, pixels * 3, hipMemcpyDeviceToHost));
    if (hit_t_host) HIP_TRY(hipMemcpy(hit_t_host, r->d_hit, pixels * sizeof(eu_f64), hipMemcpyDeviceToHost));
    eu_stats tmp;
    rc = eu_renderer_stats(r, &tmp);      /* (always: a failure must not depend on whether the caller asked for the counters) */
    if (stats) *stats = tmp;
    return rc;
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
    if (r->use_wavefront) {      /* a ray dropped from a full queue or hit stack would leave a colour made of what was left: trace the pixel again with the stack kernel */
        EuDevCounters c;
        HIP_TRY(hipMemcpy(&c, r->d_counters, sizeof c, hipMemcpyDeviceToHost));
        if (c.overflow || c.hs_full) {
            r->use_wavefront = false;
            r->retraces++;
            rc = render_device_impl(r, cam, &one, nullptr, r->d_rgba, nullptr, r->d_point, true, (uint32_t)x);
            r->use_wavefront = true;
            if (rc != EU_OK) return rc;
        }
    }
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
        uint32_t *d_rgba = nullptr; uint8_t *d_rgb = nullptr; EuDevCounters *d_cnt = nullptr;
        uint8_t *h_rgb = nullptr; EuDevCounters *h_cnt = nullptr;        /* h_rgb: the pinned image this submit copies into (one of host_rgb) */
        hipEvent_t traced = nullptr, copied = nullptr;
        uint32_t width = 0, rows = 0;
        eu_camera cam; eu_frame f;      /* the frame in this slot: one the wavefront pipeline cannot finish is traced again when it is collected */
    };
    std::vector<Slot> slots;
    /* slots + 1 pinned images, used round-robin by submit number: the image handed out by eu_sequence_next stays untouched
     * until the NEXT eu_sequence_next, however many frames are submitted in between */
    std::vector<uint8_t *> host_rgb;
    unsigned long long submitted = 0, taken = 0;
    int saved_streams = 0;                   /* the renderer's own band pipelines per frame, put back when the sequence goes */
};

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
    if (q->r && q->saved_streams) { q->r->wf_n_streams = q->saved_streams; q->r->wf_streams_auto = q->r->opts.streams == 0; }
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
    /* with several frames in flight it is the FRAMES that cover each other's kernel tails: every slot traces its frame as one band
     * (unless the caller chose a number), the launches stay large */
    const bool one_band = slots > 1 && r->opts.streams == 0;
    if (one_band) { q->saved_streams = r->wf_n_streams; r->wf_n_streams = 1; r->wf_streams_auto = false; }
    for (uint32_t k = 0; k < slots; k++) {
        if ((e = hipStreamCreateWithFlags(&q->slot_stream[k], hipStreamNonBlocking)) != hipSuccess) return fail(e, "hipStreamCreate");
        if (k > 0) {
            char cerr[256] = "";
            eu_renderer_opts o = r->opts;
            o.struct_size = sizeof o;
            o.cache_dir = r->cache_dir.empty() ? nullptr : r->cache_dir.c_str();
            o.jit_flags = r->jit_flags.empty() ? nullptr : r->jit_flags.c_str();
            if (one_band) o.streams = 1;
            const int crc = renderer_create_impl(r->flat, r->device, &o, &q->slot_renderer[k], cerr, sizeof cerr);      /* (a specialised renderer's clones find its code object in the in-process cache) */
            if (crc != EU_OK) { r->err = std::string("frame sequence slot: ") + cerr; q->slot_renderer.resize(k); eu_sequence_destroy(q); return crc; }
        }
    }
    for (auto &s : q->slots) {
        if ((e = hipMalloc((void **)&s.d_rgba, (size_t)q->max_pixels * 4)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMalloc((void **)&s.d_rgb, (size_t)q->max_pixels * 3 + 16)) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipMalloc((void **)&s.d_cnt, sizeof(EuDevCounters))) != hipSuccess) return fail(e, "hipMalloc");
        if ((e = hipHostMalloc((void **)&s.h_cnt, sizeof(EuDevCounters), hipHostMallocDefault)) != hipSuccess) return fail(e, "hipHostMalloc");
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
    HIP_TRY(hipMemcpyAsync(s.d_cnt, rs->d_counters, sizeof(EuDevCounters), hipMemcpyDeviceToDevice, trace_stream));
    HIP_TRY(hipEventRecord(s.traced, trace_stream));
    HIP_TRY(hipStreamWaitEvent(q->copy_stream, s.traced, 0));
    s.h_rgb = q->host_rgb[q->submitted % q->host_rgb.size()];
    HIP_TRY(hipMemcpyAsync(s.h_rgb, s.d_rgb, pixels * 3, hipMemcpyDeviceToHost, q->copy_stream));
    HIP_TRY(hipMemcpyAsync(s.h_cnt, s.d_cnt, sizeof(EuDevCounters), hipMemcpyDeviceToHost, q->copy_stream));
    HIP_TRY(hipEventRecord(s.copied, q->copy_stream));
    s.width = f->width; s.rows = rows;
    s.cam = *cam; s.f = *f;
    q->submitted++;
    return EU_OK;
}

extern "C" int eu_sequence_next(eu_sequence *q, const uint8_t **rgb_host, uint32_t *width, uint32_t *rows, eu_stats *stats) {
    if (!q || !rgb_host) return EU_ERR_INVALID_ARGUMENT;
    eu_renderer *r = q->r;
    if (q->taken == q->submitted) { r->err = "no frame in flight"; return EU_ERR_INVALID_ARGUMENT; }
    HIP_TRY(hipSetDevice(r->device));
    const size_t slot_no = q->taken % q->slots.size();
    eu_sequence::Slot &s = q->slots[slot_no];
    HIP_TRY(hipEventSynchronize(s.copied));
    q->taken++;
    *rgb_host = s.h_rgb;
    if (width) *width = s.width;
    if (rows) *rows = s.rows;
    const EuDevCounters *hc = s.h_cnt;
    eu_renderer *rs = q->slot_renderer[slot_no];
    if ((hc->overflow || hc->hs_full) && rs->use_wavefront) {
        /* the wavefront pipeline dropped rays (full queue or hit stack): this frame is traced again by the stack kernel, as eu_render
         * does -- behind whatever this slot's stream holds, synchronously: the slow path, visible as eu_renderer_retraces */
        hipStream_t st = q->slot_stream[slot_no];
        const size_t pixels = (size_t)s.rows * s.width;
        rs->use_wavefront = false;
        r->retraces++;
        int rc = render_device_impl(rs, &s.cam, &s.f, st, s.d_rgba, nullptr, nullptr);
        rs->use_wavefront = true;
        if (rc != EU_OK) { if (rs != r) r->err = rs->err; return rc; }
        rc = eu_pack_rgb_device(rs, s.d_rgba, s.d_rgb, pixels, st);
        if (rc != EU_OK) return rc;
        HIP_TRY(hipMemcpyAsync(s.h_rgb, s.d_rgb, pixels * 3, hipMemcpyDeviceToHost, st));
        HIP_TRY(hipMemcpyAsync(s.h_cnt, rs->d_counters, sizeof(EuDevCounters), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
    }
    if (stats) { stats->rays = hc->rays; stats->bg_samples = hc->bg_samples; stats->nan_pixels = hc->nan_pixels; stats->errors = hc->errors; }
    if (hc->overflow || hc->hs_full) { r->err = unfinished_reason(*hc); return EU_ERR_CAPACITY; }
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

/* One frame of eu_multi in flight: its per-device strip buffers, the root's gather / output buffers, a snapshot of every renderer's
 * counters taken in stream order right behind its trace (the next frame's trace zeroes the live ones), and the frame itself (an
 * overflowing device traces its strips again when the frame is collected).  Two slots: frame k's pack + peer transfer + row restore
 * run on the devices' copy streams while frame k + 1 is traced (eu_render_multi_begin / _end). */
struct MultiSlot {
    std::vector<uint32_t *> d_rgba;          /* per device: its strips, RGBA8 */
    std::vector<uint8_t *> d_rgb;            /* per device (k > 0): its strips packed to RGB8 */
    std::vector<size_t> cap_pixels;
    std::vector<EuDevCounters *> d_cnt;      /* per device: counters of this frame's trace */
    std::vector<hipEvent_t> traced, sent;    /* per device: strips traced + packed + counters saved / arrived at the root */
    hipEvent_t restored = nullptr;           /* root: d_out holds the frame */
    uint8_t *d_gathered = nullptr;           /* root: n * max_rows * width * 3 */
    uint8_t *d_out = nullptr;                /* root: rows * width * 3, frame order */
    size_t gathered_bytes = 0, out_bytes = 0;
    bool busy = false;
    eu_camera cam; eu_frame f;
    std::vector<eu_frame> fr;
    std::vector<uint32_t> lrows;
    size_t dev_stride = 0;
};

struct eu_multi {
    std::vector<eu_renderer *> r;            /* r[k] on devices[k]; r[0] is the root */
    std::vector<int> devices;
    std::vector<hipStream_t> stream;         /* per device: trace + pack */
    std::vector<hipStream_t> copy;           /* per device: peer transfer; the root's also restores the row order and reads back */
    MultiSlot slot[2];
    uint64_t begun = 0, ended = 0;
    std::string err;
};

extern "C" void eu_multi_destroy(eu_multi *m) {
    if (!m) return;
    for (size_t k = 0; k < m->r.size(); k++) {
        (void)hipSetDevice(m->devices[k]);
        if (k < m->stream.size() && m->stream[k]) { (void)hipStreamSynchronize(m->stream[k]); (void)hipStreamDestroy(m->stream[k]); }
        if (k < m->copy.size() && m->copy[k]) { (void)hipStreamSynchronize(m->copy[k]); (void)hipStreamDestroy(m->copy[k]); }
        for (MultiSlot &s : m->slot) {
            if (k < s.traced.size() && s.traced[k]) (void)hipEventDestroy(s.traced[k]);
            if (k < s.sent.size() && s.sent[k]) (void)hipEventDestroy(s.sent[k]);
            if (k < s.d_rgba.size() && s.d_rgba[k]) (void)hipFree(s.d_rgba[k]);
            if (k < s.d_rgb.size() && s.d_rgb[k]) (void)hipFree(s.d_rgb[k]);
            if (k < s.d_cnt.size() && s.d_cnt[k]) (void)hipFree(s.d_cnt[k]);
        }
        if (m->r[k]) eu_renderer_destroy(m->r[k]);
    }
    if (!m->devices.empty()) (void)hipSetDevice(m->devices[0]);
    for (MultiSlot &s : m->slot) {
        if (s.restored) (void)hipEventDestroy(s.restored);
        if (s.d_gathered) (void)hipFree(s.d_gathered);
        if (s.d_out) (void)hipFree(s.d_out);
    }
    delete m;
}

extern "C" int eu_multi_create_opts(const eu_scene *scene, const int *devices, int n_devices, const eu_renderer_opts *opts, eu_multi **out, char *err, size_t errlen) {
    if (!scene || !devices || !out || n_devices < 1 || n_devices > 64) return EU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    eu_multi *m = new eu_multi();
    m->devices.assign(devices, devices + n_devices);
    m->r.assign(n_devices, nullptr); m->stream.assign(n_devices, nullptr); m->copy.assign(n_devices, nullptr);
    for (MultiSlot &s : m->slot) {
        s.traced.assign(n_devices, nullptr); s.sent.assign(n_devices, nullptr); s.d_cnt.assign(n_devices, nullptr);
        s.d_rgba.assign(n_devices, nullptr); s.d_rgb.assign(n_devices, nullptr); s.cap_pixels.assign(n_devices, 0);
    }
    for (int k = 0; k < n_devices; k++) {
        int rc = eu_renderer_create_opts(scene, devices[k], opts, &m->r[k], err, errlen);
        if (rc != EU_OK) { eu_multi_destroy(m); return rc; }
        hipError_t e = hipSetDevice(devices[k]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->stream[k], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->copy[k], hipStreamNonBlocking);
        for (MultiSlot &s : m->slot) {
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.traced[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s.sent[k], hipEventDisableTiming);
            if (e == hipSuccess) e = hipMalloc((void **)&s.d_cnt[k], sizeof(EuDevCounters));
            if (e == hipSuccess && k == 0) e = hipEventCreateWithFlags(&s.restored, hipEventDisableTiming);
        }
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

extern "C" int eu_multi_create(const eu_scene *scene, const int *devices, int n_devices, eu_multi **out, char *err, size_t errlen) {
    return eu_multi_create_opts(scene, devices, n_devices, nullptr, out, err, errlen);
}

/* before an error return: nothing may still be in flight on any device when the caller sees the failure (frames begun are lost) */
static void multi_drain(eu_multi *m) {
    for (size_t k = 0; k < m->r.size(); k++) {
        if (hipSetDevice(m->devices[k]) != hipSuccess) continue;
        if (m->stream[k]) (void)hipStreamSynchronize(m->stream[k]);
        if (m->copy[k]) (void)hipStreamSynchronize(m->copy[k]);
    }
    (void)hipGetLastError();
    for (MultiSlot &s : m->slot) s.busy = false;
    m->ended = m->begun;
}

#define MULTI_TRY(expr)                                                                   \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                           \
            m->err = std::string(#expr) + ": " + hipGetErrorString(e_);                   \
            multi_drain(m);                                                               \
            return EU_ERR_HIP;                                                            \
        }                                                                                 \
    } while (0)

/* device k's strips of slot s: counters saved and strips packed behind the trace on its trace stream, then -- on its copy stream, so
 * that the trace stream is free for the next frame -- one transfer into its slot of the root's buffer */
static int multi_pack_and_send(eu_multi *m, MultiSlot &s, uint32_t k) {
    const uint32_t W = s.f.width;
    const size_t row_bytes = (size_t)W * 3;
    MULTI_TRY(hipSetDevice(m->devices[k]));
    MULTI_TRY(hipMemcpyAsync(s.d_cnt[k], m->r[k]->d_counters, sizeof(EuDevCounters), hipMemcpyDeviceToDevice, m->stream[k]));
    uint8_t *packed = k == 0 ? s.d_gathered : s.d_rgb[k];      /* the root packs straight into slot 0 */
    int rc = eu_pack_rgb_device(m->r[k], s.d_rgba[k], packed, (size_t)s.lrows[k] * W, m->stream[k]);
    if (rc != EU_OK) { m->err = m->r[k]->err; multi_drain(m); return rc; }
    MULTI_TRY(hipEventRecord(s.traced[k], m->stream[k]));
    MULTI_TRY(hipStreamWaitEvent(m->copy[k], s.traced[k], 0));
    if (k > 0) MULTI_TRY(hipMemcpyPeerAsync(s.d_gathered + (size_t)k * s.dev_stride, m->devices[0], packed, m->devices[k], (size_t)s.lrows[k] * row_bytes, m->copy[k]));
    MULTI_TRY(hipEventRecord(s.sent[k], m->copy[k]));
    return EU_OK;
}

static int multi_restore(eu_multi *m, MultiSlot &s) {
    const uint32_t n = (uint32_t)m->r.size(), rows = s.f.row_end - s.f.row_begin, W = s.f.width;
    const size_t row_bytes = (size_t)W * 3;
    MULTI_TRY(hipSetDevice(m->devices[0]));
    for (uint32_t k = 0; k < n; k++) if (s.lrows[k]) MULTI_TRY(hipStreamWaitEvent(m->copy[0], s.sent[k], 0));
    if (n > 1) {
        unsigned gx = (unsigned)((row_bytes + 255) / 256); if (gx > 64) gx = 64;
        hipLaunchKernelGGL(eu_restore_rows_kernel, dim3(gx, rows), dim3(256), 0, m->copy[0], s.d_gathered, s.d_out, W, s.f.row_begin, rows, n, s.dev_stride);
        MULTI_TRY(hipGetLastError());
    }
    MULTI_TRY(hipEventRecord(s.restored, m->copy[0]));
    return EU_OK;
}

extern "C" int eu_render_multi_begin(eu_multi *m, const eu_camera *cam, const eu_frame *f) {
    if (!m || !cam || !f) return EU_ERR_INVALID_ARGUMENT;
    if (f->width == 0 || f->height == 0 || f->row_begin > f->row_end || f->row_end > f->height || f->strip_count > 1) return EU_ERR_INVALID_ARGUMENT;
    if (m->begun - m->ended >= 2) { m->err = "eu_render_multi_begin: two frames are in flight already (collect one with eu_render_multi_end)"; return EU_ERR_INVALID_ARGUMENT; }
    MultiSlot &s = m->slot[m->begun % 2];
    const uint32_t n = (uint32_t)m->r.size(), rows = f->row_end - f->row_begin, W = f->width;
    s.cam = *cam; s.f = *f;
    s.fr.assign(n, *f); s.lrows.assign(n, 0);
    s.busy = true;
    m->begun++;
    if (rows == 0) return EU_OK;
    const size_t row_bytes = (size_t)W * 3;
    uint32_t max_rows = 0;
    for (uint32_t k = 0; k < n; k++) {
        s.fr[k].strip_count = n; s.fr[k].strip_index = k;
        s.lrows[k] = n > 1 ? eu_frame_local_rows(&s.fr[k]) : rows;
        if (n == 1) { s.fr[k].strip_count = 0; s.fr[k].strip_index = 0; }
        if (s.lrows[k] > max_rows) max_rows = s.lrows[k];
    }
    s.dev_stride = (size_t)max_rows * row_bytes;
    /* buffers, grown on demand */
    MULTI_TRY(hipSetDevice(m->devices[0]));
    if (s.gathered_bytes < s.dev_stride * n + 16) {
        if (s.d_gathered) (void)hipFree(s.d_gathered);
        s.d_gathered = nullptr; s.gathered_bytes = 0;
        MULTI_TRY(hipMalloc((void **)&s.d_gathered, s.dev_stride * n + 16));
        s.gathered_bytes = s.dev_stride * n + 16;
    }
    if (s.out_bytes < (size_t)rows * row_bytes + 16) {
        if (s.d_out) (void)hipFree(s.d_out);
        s.d_out = nullptr; s.out_bytes = 0;
        MULTI_TRY(hipMalloc((void **)&s.d_out, (size_t)rows * row_bytes + 16));
        s.out_bytes = (size_t)rows * row_bytes + 16;
    }
    for (uint32_t k = 0; k < n; k++) {
        const size_t pixels = (size_t)max_rows * W;
        if (s.cap_pixels[k] < pixels) {
            MULTI_TRY(hipSetDevice(m->devices[k]));
            if (s.d_rgba[k]) (void)hipFree(s.d_rgba[k]);
            if (s.d_rgb[k]) (void)hipFree(s.d_rgb[k]);
            s.d_rgba[k] = nullptr; s.d_rgb[k] = nullptr; s.cap_pixels[k] = 0;
            MULTI_TRY(hipMalloc((void **)&s.d_rgba[k], pixels * 4));
            if (k > 0) MULTI_TRY(hipMalloc((void **)&s.d_rgb[k], pixels * 3 + 16));
            s.cap_pixels[k] = pixels;
        }
    }
    /* every renderer's work buffers, streams and events first: no allocation (which may synchronise or clear memory) happens
     * once the first device's kernels are in flight */
    for (uint32_t k = 0; k < n; k++) {
        if (s.lrows[k] == 0) continue;
        MULTI_TRY(hipSetDevice(m->devices[k]));
        m->r[k]->prepare_only = true;
        const int prc = render_device_impl(m->r[k], cam, &s.fr[k], m->stream[k], s.d_rgba[k], nullptr, nullptr);
        m->r[k]->prepare_only = false;
        if (prc != EU_OK) { m->err = m->r[k]->err; multi_drain(m); return prc; }
    }
    /* trace everywhere ... */
    for (uint32_t k = 0; k < n; k++) {
        if (s.lrows[k] == 0) continue;
        MULTI_TRY(hipSetDevice(m->devices[k]));
        int rc = render_device_impl(m->r[k], cam, &s.fr[k], m->stream[k], s.d_rgba[k], nullptr, nullptr);
        if (rc != EU_OK) { m->err = m->r[k]->err; multi_drain(m); return rc; }
    }
    /* ... then pack, one transfer per device into its slot of the root's buffer, and the row order restored there */
    for (uint32_t k = 0; k < n; k++) if (s.lrows[k]) { const int rc = multi_pack_and_send(m, s, k); if (rc != EU_OK) return rc; }
    return multi_restore(m, s);
}

extern "C" int eu_render_multi_end(eu_multi *m, uint8_t *rgb_host, void **rgb_dev_root, eu_stats *stats) {
    if (!m) return EU_ERR_INVALID_ARGUMENT;
    if (stats) memset(stats, 0, sizeof *stats);
    if (m->begun == m->ended) { m->err = "eu_render_multi_end: no frame in flight"; return EU_ERR_INVALID_ARGUMENT; }
    MultiSlot &s = m->slot[m->ended % 2];
    const uint32_t n = (uint32_t)m->r.size(), rows = s.f.row_end - s.f.row_begin, W = s.f.width;
    const size_t row_bytes = (size_t)W * 3;
    if (rows == 0) { s.busy = false; m->ended++; if (rgb_dev_root) *rgb_dev_root = nullptr; return EU_OK; }
    /* a device whose ray queues overflowed (the glass-heavy strips are exactly the fan-out case) traces its strips again with the
     * stack-based kernel, as eu_render does -- behind whatever its trace stream holds of the next frame; its strips are packed and
     * sent again and the row order restored once more */
    std::vector<EuDevCounters> c(n);
    auto counters_of = [&](uint32_t k) -> int {
        MULTI_TRY(hipSetDevice(m->devices[k]));
        MULTI_TRY(hipEventSynchronize(s.traced[k]));
        MULTI_TRY(hipMemcpy(&c[k], s.d_cnt[k], sizeof(EuDevCounters), hipMemcpyDeviceToHost));
        return EU_OK;
    };
    bool any_retraced = false;
    for (uint32_t k = 0; k < n; k++) {
        if (s.lrows[k] == 0) continue;
        int rc = counters_of(k);
        if (rc != EU_OK) return rc;
        if ((!c[k].overflow && !c[k].hs_full) || !m->r[k]->use_wavefront) continue;
        eu_renderer *r = m->r[k];
        r->use_wavefront = false;
        r->retraces++;
        rc = render_device_impl(r, &s.cam, &s.fr[k], m->stream[k], s.d_rgba[k], nullptr, nullptr);
        r->use_wavefront = true;
        if (rc != EU_OK) { m->err = r->err; multi_drain(m); return rc; }
        any_retraced = true;
        if ((rc = multi_pack_and_send(m, s, k)) != EU_OK) return rc;
        if ((rc = counters_of(k)) != EU_OK) return rc;
    }
    if (any_retraced) { const int rc = multi_restore(m, s); if (rc != EU_OK) return rc; }
    MULTI_TRY(hipSetDevice(m->devices[0]));
    uint8_t *result = n > 1 ? s.d_out : s.d_gathered;
    MULTI_TRY(hipEventSynchronize(s.restored));
    if (rgb_host) {
        MULTI_TRY(hipMemcpyAsync(rgb_host, result, (size_t)rows * row_bytes, hipMemcpyDeviceToHost, m->copy[0]));
        MULTI_TRY(hipStreamSynchronize(m->copy[0]));
    }
    if (rgb_dev_root) *rgb_dev_root = result;      /* valid until the second eu_render_multi_begin from here */
    s.busy = false;
    m->ended++;
    int worst = EU_OK;
    eu_stats sum = {0, 0, 0, 0};
    for (uint32_t k = 0; k < n; k++) {
        if (s.lrows[k] == 0) continue;
        if (c[k].overflow || c[k].hs_full) { m->err = "device " + std::to_string(m->devices[k]) + ": " + unfinished_reason(c[k]) + " (the stack kernel did not take over)"; worst = EU_ERR_CAPACITY; continue; }
        sum.rays += c[k].rays; sum.bg_samples += c[k].bg_samples; sum.nan_pixels += c[k].nan_pixels; sum.errors += c[k].errors;
    }
    if (stats && worst == EU_OK) *stats = sum;      /* (never partially summed) */
    return worst;
}

extern "C" int eu_render_multi(eu_multi *m, const eu_camera *cam, const eu_frame *f, uint8_t *rgb_host, void **rgb_dev_root, eu_stats *stats) {
    if (!m || !cam || !f) return EU_ERR_INVALID_ARGUMENT;
    if (stats) memset(stats, 0, sizeof *stats);
    if (m->begun != m->ended) { m->err = "eu_render_multi: frames begun with eu_render_multi_begin are still in flight"; return EU_ERR_INVALID_ARGUMENT; }
    const int rc = eu_render_multi_begin(m, cam, f);
    if (rc != EU_OK) return rc;
    return eu_render_multi_end(m, rgb_host, rgb_dev_root, stats);
}

extern "C" const char *eu_multi_error(const eu_multi *m) { return m ? m->err.c_str() : ""; }

/* Universe::trace_path_unknown (universe/mod.rs:273-286) on the resident scene: one lane, synchronous. */
extern "C" int eu_trace_path(eu_renderer *r, const eu_f64 location[4], const eu_f64 direction[4], eu_f64 distance,
                             eu_f64 out_location[4], eu_f64 out_direction[4], int32_t *found) {
    if (!r || !location || !direction || !out_location || !out_direction || !found) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    if (r->hit_cap_strict > 96) { r->err = "scene needs a deeper hit stack than the path kernel has (96)"; return EU_ERR_CAPACITY; }
    if (!r->d_path_in) {
        HIP_TRY(hipMalloc((void **)&r->d_path_in, 9 * sizeof(real)));
        HIP_TRY(hipMalloc((void **)&r->d_path_out, sizeof(EuPathResult)));
    }
    const int D = r->dim;
    real in[9];
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
