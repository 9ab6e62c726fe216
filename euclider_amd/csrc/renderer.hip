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
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/euclider_amd.h"
#include "scene_host.hpp"
#include "trace_device.h"

#define EU_BLOCK 256

/* ------------------------------------------------------------------ the lane state machine */
template <int D, int HSCAP /* 0: hit stack in LDS (capacity = hs_cap), else private array of HSCAP */, bool SCENE_IN_LDS>
__global__ __launch_bounds__(EU_BLOCK) void eu_trace_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t hs_cap,
                                                            EuDevCamera cam, EuDevFrame fr, EuDevCounters *counters,
                                                            uint32_t *__restrict__ rgba, double *__restrict__ hit_t,
                                                            double *__restrict__ point_rgb /* single-pixel mode: un-quantised Rgb<F> */) {
    extern __shared__ uint64_t lds_dyn[];
    const uint64_t *base = scene_g;
    uint32_t lds_words = 0;
    if (SCENE_IN_LDS) {
        for (uint32_t i = threadIdx.x; i < scene_words; i += blockDim.x) lds_dyn[i] = scene_g[i];
        __syncthreads();
        base = lds_dyn;
        lds_words = scene_words;
    }
    EuScene S;
    S.init(base);

    typename std::conditional<HSCAP == 0, HitStackLds, HitStackPriv<(HSCAP ? HSCAP : 1)>>::type HS;
    if constexpr (HSCAP == 0) {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        double *hs_t = (double *)(lds_dyn + lds_words);
        uint32_t *hs_c = (uint32_t *)(hs_t + (EU_BLOCK / 64) * hs_cap * 64);
        HS.t = hs_t + wave * hs_cap * 64 + lane;
        HS.c = hs_c + wave * hs_cap * 64 + lane;
        HS.cap = hs_cap;
    }
    FrameStack<D> FS;
    LaneCounters cnt = {0, 0, 0, 0};

    const unsigned long long total_items = (unsigned long long)fr.n_tiles * 64ull;
    const uint32_t rows = fr.local_rows;

    /* lane state */
    bool active = false;
    uint32_t px_x = 0, px_y = 0, out_idx = 0;
    double o[D], d[D];
    int ent = 0;
    uint32_t depth = 0, fsp = 0;
    bool primary = false;
    double first_hit = -1.0;

    for (;;) {
        /* ---- refill: idle lanes pull the next pixel (wave-aggregated atomic) ---- */
        bool exhausted = false;
        while (!active && !exhausted) {
            const unsigned long long mask = __ballot(1);
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            unsigned long long basei = 0;
            if (rank == 0) basei = atomicAdd(&counters->next_item, (unsigned long long)__popcll(mask));
            basei = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(basei >> 32)) << 32) |
                    (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)basei);
            const unsigned long long item = basei + rank;
            if (item >= total_items) { exhausted = true; break; }
            uint32_t ry;
            if (fr.single_pixel) {   /* Environment::trace_screen_point: exactly one item */
                if (item != 0) continue;
                px_x = fr.single_x; px_y = fr.single_y; ry = 0; out_idx = 0;
            } else {
                const uint32_t tile = (uint32_t)(item >> 6), within = (uint32_t)(item & 63);
                px_x = (tile % fr.tiles_x) * 8 + (within & 7);
                ry = (tile / fr.tiles_x) * 8 + (within >> 3);
                if (px_x >= fr.width || ry >= rows) continue;
                if (fr.strip_count > 1) {   /* interleaved 8-row strips: this rank owns strips s with s % count == index */
                    const uint32_t gstrip = (ry >> 3) * fr.strip_count + fr.strip_index;
                    px_y = fr.row_begin + gstrip * 8 + (ry & 7);
                    if (px_y >= fr.row_end) {   /* padding rows of the last strip: defined contents */
                        rgba[ry * fr.width + px_x] = 0u;
                        if (hit_t) hit_t[ry * fr.width + px_x] = -1.0;
                        continue;
                    }
                } else px_y = fr.row_begin + ry;
                out_idx = ry * fr.width + px_x;
            }

            /* Environment::render's cross-hair (universe/mod.rs:321-333) */
            const uint32_t hw = fr.width / 2, hh = fr.height / 2;
            if (fr.debug_crosshair && ((px_x == hw && (px_y == hh - 1 || px_y == hh + 1)) || (px_y == hh && (px_x == hw - 1 || px_x == hw + 1)))) {
                rgba[out_idx] = 0xff0000ffu;
                if (hit_t) hit_t[out_idx] = -1.0;
                if (point_rgb) { point_rgb[0] = 1.0; point_rgb[1] = 0.0; point_rgb[2] = 0.0; }
                continue;
            }
            /* camera ray (d3/entity/camera.rs:164-185, d4/entity/camera.rs:155-176) */
            const int sw = (int)fr.width, sh = (int)fr.height;
            const double rel_x = (double)((int)px_x - sw / 2) + (double)(1 - sw % 2) / 2.0;
            const double rel_y = (double)((int)px_y - sh / 2) + (double)(1 - sh % 2) / 2.0;
            double dl[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
                const double center = cam.location[i] + cam.forward[i] * cam.dist;
                const double p = center + (cam.up[i] * rel_y) + (cam.right[i] * rel_x);
                dl[i] = p - cam.location[i];
                o[i] = cam.location[i];
            }
            vnormalize<D>(dl, d);
            /* trace_unknown (universe/mod.rs:253-271) */
            ent = material_at<D>(S, o);
            if (ent < 0) {   /* trace_screen_point's checkerboard (universe/mod.rs:387-395) */
                const bool black = (((int)px_x / 8 + (int)px_y / 8) % 2) == 0;
                rgba[out_idx] = black ? 0xff000000u : 0xffff00ffu;
                if (hit_t) hit_t[out_idx] = -1.0;
                if (point_rgb) { point_rgb[0] = black ? 0.0 : 1.0; point_rgb[1] = 0.0; point_rgb[2] = black ? 0.0 : 1.0; }
                continue;
            }
            material_apply<D>(S, S.entity((uint32_t)ent)->material, d, false);
            depth = cam.max_depth;
            fsp = 0;
            primary = true;
            first_hit = -1.0;
            active = true;
        }
        if (!active) break;   /* no work left for this lane */

        /* ---- TRACE one segment: Universe::trace (universe/mod.rs:149-184) ---- */
        Rgba ret = {0.0, 0.0, 0.0, 0.0};
        bool returning = false;
        {
            bool have = false;
            double best_t = 0.0;
            uint32_t best_code = 0, best_ent = 0;
            if (depth > 0) {
                cnt.rays++;
                /* trace_closest (universe/mod.rs:85-147): first hit of every surfaced entity, strict minimum */
                for (uint32_t e = 0; e < S.n_entities; e++) {
                    const EuFlatEntity *E = S.entity(e);
                    if (E->surface < 0) continue;
                    double t = 0.0; uint32_t code = 0;
                    const uint32_t n = eval_shape<D>(S, E->shape_first, E->shape_root, o, d, HS, cnt, t, code);
                    if (n == 0) continue;
                    if (!have || best_t > t) { have = true; best_t = t; best_code = code; best_ent = e; }
                }
            }
            if (primary) { first_hit = have ? best_t : -1.0; primary = false; }
            if (have) {
                /* ComposableSurface::get_color (surface.rs:62-162) */
                HitCtx<D> c;
#pragma unroll
                for (int i = 0; i < D; i++) { c.loc[i] = o[i] + d[i] * best_t; c.dir[i] = d[i]; }
                hit_normal<D>(S, best_code, o, d, c.loc, c.normal);
                c.exiting = angle_between<D>(c.dir, c.normal) < EU_FRAC_PI_2_C;      /* universe/mod.rs:118-125 */
#pragma unroll
                for (int i = 0; i < D; i++) c.nc[i] = c.exiting ? -c.normal[i] : c.normal[i];
                const EuFlatEntity *HE = S.entity(best_ent);
                const EuFlatSurface *F = S.surface((uint32_t)HE->surface);
                double ratio = reflection_ratio<D>(F, c);
                ratio = rust_max(rust_min(ratio, 1.0), 0.0);

                bool have_inter = false, need_trans = false;
                Rgba inter = {0.0, 0.0, 0.0, 0.0};
                uint32_t spx = 0;
                double t_o[D], t_d[D];
                int dest = -1;
                if (!(ratio >= 1.0)) {                                              /* get_intersection_color */
                    const Rgba sc = surface_color<D>(S, F, c, fr.time_s, cnt);
                    spx = to_pixel4(sc, cnt);
                    if ((spx >> 24) == 255u) { inter = sc; have_inter = true; }
                    else {
                        threshold_direction<D>(F, c, t_d);
#pragma unroll
                        for (int i = 0; i < D; i++) t_o[i] = c.loc[i] + -c.nc[i] * EU_EPS * 128.0;
                        dest = c.exiting ? material_at<D>(S, t_o) : (int)best_ent;
                        if (dest >= 0) {
                            material_apply<D>(S, S.entity((uint32_t)ent)->material, t_d, true);
                            material_apply<D>(S, S.entity((uint32_t)dest)->material, t_d, false);
                            need_trans = true;
                        }
                    }
                }
                const bool need_refl = !(ratio <= 0.0);                              /* get_reflection_color */
                double r_o[D], r_d[D];
                if (need_refl) {
                    const double dn = vdot<D>(c.dir, c.nc);
#pragma unroll
                    for (int i = 0; i < D; i++) {
                        r_d[i] = c.nc[i] * -2.0 * dn + c.dir[i];                     /* surface.rs:246-256 */
                        r_o[i] = c.loc[i] + c.nc[i] * EU_EPS * 128.0;
                    }
                }
                const uint32_t child_depth = depth - 1;
                if (need_trans) {
                    if (need_refl) {
                        FS.meta[fsp] = FR_TRANS_THEN_REFL | (child_depth << 8) | ((uint32_t)ent << 16);
                        FS.ratio[fsp] = ratio;
#pragma unroll
                        for (int i = 0; i < D; i++) { FS.data[fsp][i] = r_o[i]; FS.data[fsp][D + i] = r_d[i]; }
                    } else {
                        FS.meta[fsp] = FR_OVER;
                    }
                    FS.px[fsp] = spx;
                    fsp++;
#pragma unroll
                    for (int i = 0; i < D; i++) { o[i] = t_o[i]; d[i] = t_d[i]; }
                    ent = dest;
                    depth = child_depth;
                } else if (need_refl) {
                    if (have_inter) {
                        FS.meta[fsp] = FR_COMBINE;
                        FS.ratio[fsp] = ratio;
                        FS.data[fsp][0] = inter.r; FS.data[fsp][1] = inter.g; FS.data[fsp][2] = inter.b; FS.data[fsp][3] = inter.a;
                        fsp++;
                    }   /* else: the reflection colour is the result (surface.rs:153-154): tail call */
#pragma unroll
                    for (int i = 0; i < D; i++) { o[i] = r_o[i]; d[i] = r_d[i]; }
                    depth = child_depth;
                } else {
                    if (!have_inter) cnt.errors++;            /* the reference panics here (surface.rs:154) */
                    ret = inter;
                    returning = true;
                }
            } else {
                /* background().get_color(&direction.to_point()) (universe/mod.rs:183) */
                cnt.bg++;
                double pt[D];
#pragma unroll
                for (int i = 0; i < D; i++) pt[i] = 0.0 + d[i];
                ret = mapped_get_color(S, S.background, pt, cnt);
                returning = true;
            }
        }

        /* ---- RETURN through pending frames ---- */
        while (returning) {
            if (fsp == 0) {
                /* trace_unknown: fg.over(white) un-premultiplied, then Rgb::to_pixel (universe/mod.rs:263-269,342) */
                const Rgba white = {1.0, 1.0, 1.0, 1.0};
                const Rgba out = from_premultiplied(blend_pre(EU_BL_OVER, into_premultiplied(ret), into_premultiplied(white)));
                const uint32_t idx = out_idx;
                rgba[idx] = to_u8(out.r, cnt) | (to_u8(out.g, cnt) << 8) | (to_u8(out.b, cnt) << 16) | 0xff000000u;
                if (hit_t) hit_t[idx] = first_hit;
                if (point_rgb) { point_rgb[0] = out.r; point_rgb[1] = out.g; point_rgb[2] = out.b; }
                active = false;
                break;
            }
            fsp--;
            const uint32_t meta = FS.meta[fsp];
            const uint32_t kind = meta & 0xff;
            if (kind == FR_COMBINE) {                                               /* surface.rs:159-161 */
                const Rgba inter = {FS.data[fsp][0], FS.data[fsp][1], FS.data[fsp][2], FS.data[fsp][3]};
                ret = combine_palette_color(ret, inter, FS.ratio[fsp]);
            } else {
                /* surface_palette.over(transition_palette), both re-quantised to u8 (surface.rs:104-114) */
                const uint32_t tpx = to_pixel4(ret, cnt);
                const Rgba inter = blend_rgba(EU_BL_OVER, new_u8(FS.px[fsp]), new_u8(tpx));
                if (kind == FR_OVER) ret = inter;
                else {                                                              /* now the reflection child */
#pragma unroll
                    for (int i = 0; i < D; i++) { o[i] = FS.data[fsp][i]; d[i] = FS.data[fsp][D + i]; }
                    ent = (int)(meta >> 16);
                    depth = (meta >> 8) & 0xff;
                    FS.meta[fsp] = FR_COMBINE;
                    FS.data[fsp][0] = inter.r; FS.data[fsp][1] = inter.g; FS.data[fsp][2] = inter.b; FS.data[fsp][3] = inter.a;
                    fsp++;
                    returning = false;
                }
            }
        }
    }

    /* ---- counters: wave reduction, one atomic per wave and counter ---- */
    unsigned long long v0 = cnt.rays, v1 = cnt.bg, v2 = cnt.nan_px, v3 = cnt.errors;
    for (int off = 32; off > 0; off >>= 1) {
        v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); v3 += __shfl_down(v3, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (v0) atomicAdd(&counters->rays, v0);
        if (v1) atomicAdd(&counters->bg_samples, v1);
        if (v2) atomicAdd(&counters->nan_pixels, v2);
        if (v3) atomicAdd(&counters->errors, v3);
    }
}

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

__global__ void eu_math_kernel(int fn, const double *x, const double *y, double *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = x[i], b = y ? y[i] : 0.0, r;
    switch (fn) {
    case 0: r = eu_acos(a); break;
    case 1: r = eu_asin(a); break;
    case 2: r = eu_sin(a); break;
    case 3: r = eu_cos(a); break;
    case 4: r = eu_tan(a); break;
    case 5: r = eu_atan2(a, b); break;
    case 6: r = sqrt(a); break;
    case 7: r = a / b; break;
    case 8: r = fmod(a, b); break;
    default: r = 0.0;
    }
    out[i] = r;
}

/* ------------------------------------------------------------------ host side */
struct eu_renderer {
    int device = 0;
    int dim = 3;
    uint32_t hit_cap = 0;
    uint32_t scene_words = 0;
    uint64_t *d_scene = nullptr;
    std::vector<void *> d_textures;
    EuDevCounters *d_counters = nullptr;
    uint32_t *d_rgba = nullptr; size_t rgba_pixels = 0;      /* internal frame buffers for eu_render */
    uint8_t *d_rgb = nullptr;
    double *d_hit = nullptr;
    double *d_point = nullptr;
    static constexpr int EV_RING = 64;        /* per-launch HIP event pairs, on the launch stream */
    hipEvent_t ev_start[EV_RING] = {}, ev_stop[EV_RING] = {};
    unsigned long long launches = 0;
    hipStream_t last_stream = nullptr;
    bool have_timing = false;
    int num_cus = 0;
    bool scene_in_lds = true;
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

extern "C" int eu_renderer_create(const eu_scene *scene, int device, eu_renderer **out, char *err, size_t errlen) {
    if (!scene || !out) return EU_ERR_INVALID_ARGUMENT;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        set_err(err, errlen, "no usable HIP device (this library has no CPU fallback)");
        return EU_ERR_NO_DEVICE;
    }
    eu_renderer *r = new eu_renderer();
    r->device = device;
    const EuFlatHeader &h = scene->flat.header();
    r->dim = (int)h.dim;
    r->hit_cap = h.hit_cap;
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
        HIP_TRY(hipMalloc((void **)&r->d_scene, blob.size() * 8));
        HIP_TRY(hipMemcpy(r->d_scene, blob.data(), blob.size() * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&r->d_counters, sizeof(EuDevCounters)));
        HIP_TRY(hipMemset(r->d_counters, 0, sizeof(EuDevCounters)));
        HIP_TRY(hipMalloc((void **)&r->d_point, 3 * sizeof(double)));
        for (int i = 0; i < eu_renderer::EV_RING; i++) { HIP_TRY(hipEventCreate(&r->ev_start[i])); HIP_TRY(hipEventCreate(&r->ev_stop[i])); }
        return EU_OK;
    };
    int rc = body();
    if (rc != EU_OK) return failhip(rc);
    *out = r;
    return EU_OK;
}

extern "C" void eu_renderer_destroy(eu_renderer *r) {
    if (!r) return;
    (void)hipSetDevice(r->device);
    for (void *p : r->d_textures) (void)hipFree(p);
    if (r->d_scene) (void)hipFree(r->d_scene);
    if (r->d_counters) (void)hipFree(r->d_counters);
    if (r->d_rgba) (void)hipFree(r->d_rgba);
    if (r->d_rgb) (void)hipFree(r->d_rgb);
    if (r->d_hit) (void)hipFree(r->d_hit);
    if (r->d_point) (void)hipFree(r->d_point);
    for (int i = 0; i < eu_renderer::EV_RING; i++) { if (r->ev_start[i]) (void)hipEventDestroy(r->ev_start[i]); if (r->ev_stop[i]) (void)hipEventDestroy(r->ev_stop[i]); }
    delete r;
}

static int make_dev_camera(const eu_camera *cam, const eu_frame *f, EuDevCamera &dc) {
    const int D = cam->dim;
    memset(&dc, 0, sizeof dc);
    for (int i = 0; i < D; i++) { dc.location[i] = cam->location[i]; dc.forward[i] = cam->forward[i]; dc.up[i] = cam->up[i]; }
    if (D == 3) {   /* get_right = cross(forward, up).normalize(), d3/entity/camera.rs:62-64 */
        double cr[3];
        cr[0] = cam->forward[1] * cam->up[2] - cam->forward[2] * cam->up[1];
        cr[1] = cam->forward[2] * cam->up[0] - cam->forward[0] * cam->up[2];
        cr[2] = cam->forward[0] * cam->up[1] - cam->forward[1] * cam->up[0];
        double n = sqrt((cr[0] * cr[0] + cr[1] * cr[1]) + cr[2] * cr[2]);
        for (int i = 0; i < 3; i++) dc.right[i] = cr[i] / n;
    } else {        /* right = -left, d4/entity/camera.rs:167 */
        for (int i = 0; i < D; i++) dc.right[i] = -cam->left[i];
    }
    const double w = (double)f->width, h = (double)f->height;
    const double fov_rad = EU_PI_C * (double)cam->fov_deg / 180.0;
    dc.dist = sqrt(w * w + h * h) / (2.0 * eu_tan(fov_rad / 2.0));
    dc.max_depth = cam->max_depth;
    return EU_OK;
}

template <int D, int HSCAP, bool LDS>
static hipError_t launch_trace(eu_renderer *r, hipStream_t stream, const EuDevCamera &dc, const EuDevFrame &df, uint32_t *rgba, double *hit_t, double *point) {
    auto kern = eu_trace_kernel<D, HSCAP, LDS>;
    const uint32_t hs_cap = HSCAP ? (uint32_t)HSCAP : (r->hit_cap < 4 ? 4u : ((r->hit_cap + 3u) & ~3u));
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

static int render_device_impl(eu_renderer *r, const eu_camera *cam, const eu_frame *f, hipStream_t stream, uint32_t *rgba, double *hit_t, double *point,
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
    df.time_s = (double)f->time_ms / 1000.0;
    if (single) { df.strip_count = 0; df.local_rows = 1; df.single_pixel = 1; df.single_x = single_x; df.single_y = f->row_begin; df.tiles_x = 1; df.n_tiles = 1; }
    if (rows == 0) return EU_OK;
    HIP_TRY(hipMemsetAsync(r->d_counters, 0, sizeof(EuDevCounters), stream));
    const int slot = (int)(r->launches % eu_renderer::EV_RING);
    HIP_TRY(hipEventRecord(r->ev_start[slot], stream));
    hipError_t e;
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

extern "C" int eu_render_device(eu_renderer *r, const eu_camera *cam, const eu_frame *f, void *hip_stream, void *rgba_dev, double *hit_t_dev) {
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
    return EU_OK;
}

extern "C" int eu_renderer_kernel_ms(eu_renderer *r, float *ms) {
    return eu_renderer_kernel_ms_history(r, ms, 1) == 1 ? EU_OK : EU_ERR_INVALID_ARGUMENT;
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
    if (want_hit && !r->d_hit) HIP_TRY(hipMalloc((void **)&r->d_hit, r->rgba_pixels * sizeof(double)));
    return EU_OK;
}

extern "C" int eu_render(eu_renderer *r, const eu_camera *cam, const eu_frame *f, uint8_t *rgb_host, double *hit_t_host, eu_stats *stats) {
    if (!r || !cam || !f || !rgb_host) return EU_ERR_INVALID_ARGUMENT;
    if (f->row_begin > f->row_end || f->row_end > f->height) return EU_ERR_INVALID_ARGUMENT;
    const size_t pixels = (size_t)eu_frame_local_rows(f) * f->width;
    if (pixels == 0) { if (stats) memset(stats, 0, sizeof *stats); return EU_OK; }
    HIP_TRY(hipSetDevice(r->device));
    int rc = ensure_buffers(r, pixels, hit_t_host != nullptr);
    if (rc != EU_OK) return rc;
    rc = render_device_impl(r, cam, f, nullptr, r->d_rgba, hit_t_host ? r->d_hit : nullptr, nullptr);
    if (rc != EU_OK) return rc;
    rc = eu_pack_rgb_device(r, r->d_rgba, r->d_rgb, pixels, nullptr);
    if (rc != EU_OK) return rc;
    HIP_TRY(hipMemcpy(rgb_host, r->d_rgb, pixels * 3, hipMemcpyDeviceToHost));
    if (hit_t_host) HIP_TRY(hipMemcpy(hit_t_host, r->d_hit, pixels * sizeof(double), hipMemcpyDeviceToHost));
    if (stats) return eu_renderer_stats(r, stats);
    return EU_OK;
}

extern "C" int eu_trace_screen_point(eu_renderer *r, const eu_camera *cam, const eu_frame *f, int32_t x, int32_t y, double rgb[3]) {
    if (!r || !cam || !f || !rgb) return EU_ERR_INVALID_ARGUMENT;
    if (x < 0 || y < 0 || (uint32_t)x >= f->width || (uint32_t)y >= f->height) return EU_ERR_INVALID_ARGUMENT;
    HIP_TRY(hipSetDevice(r->device));
    int rc = ensure_buffers(r, 64, false);
    if (rc != EU_OK) return rc;
    eu_frame one = *f;
    one.row_begin = (uint32_t)y; one.row_end = (uint32_t)y + 1;
    rc = render_device_impl(r, cam, &one, nullptr, r->d_rgba, nullptr, r->d_point, true, (uint32_t)x);
    if (rc != EU_OK) return rc;
    HIP_TRY(hipMemcpy(rgb, r->d_point, 3 * sizeof(double), hipMemcpyDeviceToHost));
    return EU_OK;
}

extern "C" int eu_selftest_math(int device, int fn, const double *x, const double *y, double *out, size_t n) {
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device < 0 || device >= cnt) return EU_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return EU_ERR_HIP;
    if (n == 0) return EU_OK;
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
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
