/*
 * trace_wavefront.h -- the trace loop as a wavefront pipeline for gfx950.
 *
 * The reference recurses per pixel: trace -> trace_closest -> get_color -> trace ...
 * (/root/reference/src/universe/mod.rs:149-184, universe/entity/surface.rs:62-162).  Here the recursion
 * tree of the whole frame is processed one GENERATION (= recursion depth) at a time:
 *
 *   gen kernel            one thread per pixel: camera ray, material_at, Material::enter -> ray queue 0
 *   for g = 0 .. max_depth-1
 *     intersect kernel    one thread per queued ray: Universe::trace_closest -> (t, hit code, entity)
 *     shade kernel        one thread per ray: ComposableSurface::get_color up to the recursive calls:
 *                         a finished colour is DELIVERED to the parent's child slot; otherwise a tree
 *                         node {kind, quantised surface pixel, ratio} is created and 1-2 child rays are
 *                         appended to queue g+1 (wave-aggregated atomics)
 *   for g = max_depth-1 .. 0
 *     resolve kernel      one thread per node of generation g: over / combine of the delivered child
 *                         colours (surface.rs:104-114,159-161) -> delivered to its own parent
 *   final kernel          per pixel: fg.over(white), to_pixel (universe/mod.rs:263-269,342) -> RGBA8
 *
 * Everything a ray needs between kernels lives in HBM as struct-of-arrays queues (coalesced), sized
 * for the 288 GB part; every kernel is small enough for the register allocator to reach several
 * waves per SIMD, and queues keep all 64 lanes busy whatever the per-pixel ray count is.  The order in
 * which rays are processed does not matter: every step is a pure function, so the result is
 * bit-identical to the depth-first recursion.
 */
#ifndef EU_TRACE_WAVEFRONT_H
#define EU_TRACE_WAVEFRONT_H

#include "trace_device.h"

#define EU_WF_BLOCK 256

enum { WF_NONE = 0, WF_ROOT = 1, WF_OVER = 2, WF_COMBINE_TRANS = 3, WF_COMBINE_INTER = 4 };

struct EuWfBuffers {
    double *ray_od[2];          /* [2*D][ray_cap] origin then direction, component-major; ping-pong by generation */
    uint32_t *ray_parent[2];    /* node id that receives this ray's colour */
    uint32_t *ray_aux[2];       /* entity the ray travels in (bits 0..15) | child slot in the parent (bit 16) */
    double *hit_t;              /* per ray of the current generation */
    uint32_t *hit_code;
    uint32_t *hit_ent;          /* 0xffffffff: nothing hit */
    /* tree nodes: id < npix are the per-pixel roots, then one id per traced ray in queue order */
    double *node_child;         /* [node_cap][2][4] colours delivered by the children */
    double *node_ratio;
    uint32_t *node_px;
    uint32_t *node_meta;        /* kind | slot-in-parent << 8 */
    uint32_t *node_parent;
    uint32_t ray_cap, node_cap, npix, pad;
};

/* generation bookkeeping lives in EuDevCounters::gen_count[] (device memory, zeroed per frame) */

EU_DEV void wf_flush_counters(EuDevCounters *counters, const LaneCounters &cnt) {
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

/* wave-aggregated append: returns this lane's slot in the queue (call from divergent code is fine) */
EU_DEV unsigned long long wf_append(unsigned long long *counter) {
    const unsigned long long mask = __ballot(1);
    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    unsigned long long base = 0;
    if (rank == 0) base = atomicAdd(counter, (unsigned long long)__popcll(mask));
    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
           (unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)base);
    return base + rank;
}

EU_DEV const uint64_t *wf_stage_scene(const uint64_t *scene_g, uint32_t scene_words, uint64_t *lds) {
    for (uint32_t i = threadIdx.x; i < scene_words; i += blockDim.x) lds[i] = scene_g[i];
    __syncthreads();
    return lds;
}

EU_DEV void wf_deliver(const EuWfBuffers &B, uint32_t parent, uint32_t slot, const Rgba &c) {
    double *p = B.node_child + ((size_t)parent * 2 + slot) * 4;
    p[0] = c.r; p[1] = c.g; p[2] = c.b; p[3] = c.a;
}

template <int D> EU_DEV Rgba wf_background(const EuScene &S, const double *d, LaneCounters &cnt) {
    /* background().get_color(&direction.to_point()) (universe/mod.rs:183) */
    cnt.bg++;
    double pt[D];
#pragma unroll
    for (int i = 0; i < D; i++) pt[i] = 0.0 + d[i];
    return mapped_get_color(S, S.background, pt, cnt);
}

/* ------------------------------------------------------------------ primary rays */
template <int D>
__global__ __launch_bounds__(EU_WF_BLOCK) void eu_wf_gen_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, EuDevCamera cam, EuDevFrame fr,
                                                                EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba,
                                                                double *__restrict__ hit_t, double *__restrict__ point_rgb) {
    extern __shared__ uint64_t lds_dyn[];
    EuScene S;
    S.init(wf_stage_scene(scene_g, scene_words, lds_dyn));
    const unsigned long long total_items = (unsigned long long)fr.n_tiles * 64ull;
    const uint32_t rows = fr.local_rows;
    for (unsigned long long item = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; item < total_items;
         item += (unsigned long long)gridDim.x * blockDim.x) {
        uint32_t px_x, px_y, ry, out_idx;
        if (fr.single_pixel) {      /* Environment::trace_screen_point: exactly one item */
            if (item != 0) continue;
            px_x = fr.single_x; px_y = fr.single_y; ry = 0; out_idx = 0;
        } else {
            const uint32_t tile = (uint32_t)(item >> 6), within = (uint32_t)(item & 63);   /* 8x8 pixel tiles: coherent waves */
            px_x = (tile % fr.tiles_x) * 8 + (within & 7);
            ry = (tile / fr.tiles_x) * 8 + (within >> 3);
            if (px_x >= fr.width || ry >= rows) continue;
            out_idx = ry * fr.width + px_x;
            if (fr.strip_count > 1) {   /* interleaved 8-row strips: this rank owns strips s with s % count == index */
                const uint32_t gstrip = (ry >> 3) * fr.strip_count + fr.strip_index;
                px_y = fr.row_begin + gstrip * 8 + (ry & 7);
                if (px_y >= fr.row_end) {   /* padding rows of the last strip: defined contents */
                    rgba[out_idx] = 0u;
                    if (hit_t) hit_t[out_idx] = -1.0;
                    B.node_meta[out_idx] = WF_NONE;
                    continue;
                }
            } else px_y = fr.row_begin + ry;
        }
        B.node_meta[out_idx] = WF_NONE;
        if (hit_t) hit_t[out_idx] = -1.0;
        /* Environment::render's cross-hair (universe/mod.rs:321-333) */
        const uint32_t hw = fr.width / 2, hh = fr.height / 2;
        if (fr.debug_crosshair && ((px_x == hw && (px_y == hh - 1 || px_y == hh + 1)) || (px_y == hh && (px_x == hw - 1 || px_x == hw + 1)))) {
            rgba[out_idx] = 0xff0000ffu;
            if (point_rgb) { point_rgb[0] = 1.0; point_rgb[1] = 0.0; point_rgb[2] = 0.0; }
            continue;
        }
        /* camera ray (d3/entity/camera.rs:164-185, d4/entity/camera.rs:155-176) */
        const int sw = (int)fr.width, sh = (int)fr.height;
        const double rel_x = (double)((int)px_x - sw / 2) + (double)(1 - sw % 2) / 2.0;
        const double rel_y = (double)((int)px_y - sh / 2) + (double)(1 - sh % 2) / 2.0;
        double o[D], d[D], dl[D];
#pragma unroll
        for (int i = 0; i < D; i++) {
            const double center = cam.location[i] + cam.forward[i] * cam.dist;
            const double p = center + (cam.up[i] * rel_y) + (cam.right[i] * rel_x);
            dl[i] = p - cam.location[i];
            o[i] = cam.location[i];
        }
        vnormalize<D>(dl, d);
        /* trace_unknown (universe/mod.rs:253-271) */
        const int ent = material_at<D>(S, o);
        if (ent < 0) {   /* trace_screen_point's checkerboard (universe/mod.rs:387-395) */
            const bool black = (((int)px_x / 8 + (int)px_y / 8) % 2) == 0;
            rgba[out_idx] = black ? 0xff000000u : 0xffff00ffu;
            if (point_rgb) { point_rgb[0] = black ? 0.0 : 1.0; point_rgb[1] = 0.0; point_rgb[2] = black ? 0.0 : 1.0; }
            continue;
        }
        material_apply<D>(S, S.entity((uint32_t)ent)->material, d, false);
        B.node_meta[out_idx] = WF_ROOT;
        if (cam.max_depth == 0) {   /* trace() with depth 0 goes straight to the background */
            LaneCounters cnt = {0, 0, 0, 0};
            const Rgba c = wf_background<D>(S, d, cnt);
            wf_deliver(B, out_idx, 0, c);
            atomicAdd(&counters->bg_samples, 1ull);
            if (cnt.errors) atomicAdd(&counters->errors, (unsigned long long)cnt.errors);
            continue;
        }
        const unsigned long long pos = wf_append(&counters->gen_count[0]);
        if (pos >= B.ray_cap) { atomicAdd(&counters->overflow, 1ull); continue; }
#pragma unroll
        for (int i = 0; i < D; i++) { B.ray_od[0][(size_t)i * B.ray_cap + pos] = o[i]; B.ray_od[0][(size_t)(D + i) * B.ray_cap + pos] = d[i]; }
        B.ray_parent[0][pos] = out_idx;
        B.ray_aux[0][pos] = (uint32_t)ent;
    }
}

/* ------------------------------------------------------------------ trace_closest */
template <int D>
__global__ __launch_bounds__(EU_WF_BLOCK) void eu_wf_intersect_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t hs_cap, uint32_t gen,
                                                                      EuWfBuffers B, EuDevCounters *counters, double *__restrict__ hit_t_aov) {
    extern __shared__ uint64_t lds_dyn[];
    EuScene S;
    S.init(wf_stage_scene(scene_g, scene_words, lds_dyn));
    HitStackLds HS;
    {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        double *hs_t = (double *)(lds_dyn + scene_words);
        uint32_t *hs_c = (uint32_t *)(hs_t + (EU_WF_BLOCK / 64) * hs_cap * 64);
        HS.t = hs_t + wave * hs_cap * 64 + lane;
        HS.c = hs_c + wave * hs_cap * 64 + lane;
        HS.cap = hs_cap;
    }
    LaneCounters cnt = {0, 0, 0, 0};
    const unsigned long long count = counters->gen_count[gen] < B.ray_cap ? counters->gen_count[gen] : B.ray_cap;
    const uint32_t in = gen & 1u;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
        double o[D], d[D];
#pragma unroll
        for (int k = 0; k < D; k++) { o[k] = B.ray_od[in][(size_t)k * B.ray_cap + i]; d[k] = B.ray_od[in][(size_t)(D + k) * B.ray_cap + i]; }
        cnt.rays++;
        /* trace_closest (universe/mod.rs:85-147): first hit of every surfaced entity, strict minimum */
        bool have = false;
        double best_t = 0.0;
        uint32_t best_code = 0, best_ent = 0xffffffffu;
        for (uint32_t e = 0; e < S.n_entities; e++) {
            const EuFlatEntity *E = S.entity(e);
            if (E->surface < 0) continue;
            double t = 0.0; uint32_t code = 0;
            const uint32_t n = eval_shape<D>(S, E->shape_first, E->shape_root, o, d, HS, cnt, t, code);
            if (n == 0) continue;
            if (!have || best_t > t) { have = true; best_t = t; best_code = code; best_ent = e; }
        }
        B.hit_t[i] = best_t;
        B.hit_code[i] = best_code;
        B.hit_ent[i] = best_ent;
        if (gen == 0 && hit_t_aov) hit_t_aov[B.ray_parent[0][i]] = have ? best_t : -1.0;
    }
    wf_flush_counters(counters, cnt);
}

/* ------------------------------------------------------------------ ComposableSurface::get_color up to the recursive calls */
template <int D>
EU_DEV void wf_spawn(const EuScene &S, const EuWfBuffers &B, EuDevCounters *counters, uint32_t gen, uint32_t child_depth,
                     const double *o, const double *d, uint32_t ent, uint32_t parent, uint32_t slot, LaneCounters &cnt) {
    if (child_depth == 0) {     /* trace() with max_depth 0: background only (universe/mod.rs:157,183) */
        wf_deliver(B, parent, slot, wf_background<D>(S, d, cnt));
        return;
    }
    const unsigned long long pos = wf_append(&counters->gen_count[gen + 1]);
    if (pos >= B.ray_cap) { atomicAdd(&counters->overflow, 1ull); return; }
    const uint32_t out = (gen + 1) & 1u;
#pragma unroll
    for (int k = 0; k < D; k++) { B.ray_od[out][(size_t)k * B.ray_cap + pos] = o[k]; B.ray_od[out][(size_t)(D + k) * B.ray_cap + pos] = d[k]; }
    B.ray_parent[out][pos] = parent;
    B.ray_aux[out][pos] = ent | (slot << 16);
}

template <int D>
__global__ __launch_bounds__(EU_WF_BLOCK) void eu_wf_shade_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t gen, uint32_t max_depth, double time_s,
                                                                  EuWfBuffers B, EuDevCounters *counters) {
    extern __shared__ uint64_t lds_dyn[];
    EuScene S;
    S.init(wf_stage_scene(scene_g, scene_words, lds_dyn));
    LaneCounters cnt = {0, 0, 0, 0};
    const unsigned long long count = counters->gen_count[gen] < B.ray_cap ? counters->gen_count[gen] : B.ray_cap;
    unsigned long long gen_base = B.npix;
    for (uint32_t h = 0; h < gen; h++) gen_base += counters->gen_count[h] < B.ray_cap ? counters->gen_count[h] : B.ray_cap;
    const uint32_t in = gen & 1u;
    const uint32_t child_depth = max_depth - gen - 1;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long nid64 = gen_base + i;
        const uint32_t parent = B.ray_parent[in][i];
        const uint32_t aux = B.ray_aux[in][i];
        const uint32_t ent = aux & 0xffffu, slot = (aux >> 16) & 1u;
        double o[D], d[D];
#pragma unroll
        for (int k = 0; k < D; k++) { o[k] = B.ray_od[in][(size_t)k * B.ray_cap + i]; d[k] = B.ray_od[in][(size_t)(D + k) * B.ray_cap + i]; }
        const uint32_t hit_ent = B.hit_ent[i];
        if (nid64 >= B.node_cap) { atomicAdd(&counters->overflow, 1ull); continue; }
        const uint32_t nid = (uint32_t)nid64;
        uint32_t node_kind = WF_NONE;
        if (hit_ent == 0xffffffffu) {
            wf_deliver(B, parent, slot, wf_background<D>(S, d, cnt));
        } else {
            const double best_t = B.hit_t[i];
            const uint32_t best_code = B.hit_code[i];
            HitCtx<D> c;
#pragma unroll
            for (int k = 0; k < D; k++) { c.loc[k] = o[k] + d[k] * best_t; c.dir[k] = d[k]; }
            hit_normal<D>(S, best_code, o, d, c.loc, c.normal);
            c.exiting = angle_between<D>(c.dir, c.normal) < EU_FRAC_PI_2_C;      /* universe/mod.rs:118-125 */
#pragma unroll
            for (int k = 0; k < D; k++) c.nc[k] = c.exiting ? -c.normal[k] : c.normal[k];
            const EuFlatEntity *HE = S.entity(hit_ent);
            const EuFlatSurface *F = S.surface((uint32_t)HE->surface);
            double ratio = reflection_ratio<D>(F, c);
            ratio = rust_max(rust_min(ratio, 1.0), 0.0);                          /* surface.rs:145-147 */

            bool have_inter = false, need_trans = false;
            Rgba inter = {0.0, 0.0, 0.0, 0.0};
            uint32_t spx = 0;
            double t_o[D], t_d[D];
            int dest = -1;
            if (!(ratio >= 1.0)) {                                                /* get_intersection_color, surface.rs:62-117 */
                const Rgba sc = surface_color<D>(S, F, c, time_s, cnt);
                spx = to_pixel4(sc, cnt);
                if ((spx >> 24) == 255u) { inter = sc; have_inter = true; }
                else {
                    threshold_direction<D>(F, c, t_d);
#pragma unroll
                    for (int k = 0; k < D; k++) t_o[k] = c.loc[k] + -c.nc[k] * EU_EPS * 128.0;
                    dest = c.exiting ? material_at<D>(S, t_o) : (int)hit_ent;
                    if (dest >= 0) {
                        material_apply<D>(S, S.entity(ent)->material, t_d, true);
                        material_apply<D>(S, S.entity((uint32_t)dest)->material, t_d, false);
                        need_trans = true;
                    }
                }
            }
            const bool need_refl = !(ratio <= 0.0);                                /* get_reflection_color, surface.rs:119-139 */
            double r_o[D], r_d[D];
            if (need_refl) {
                const double dn = vdot<D>(c.dir, c.nc);
#pragma unroll
                for (int k = 0; k < D; k++) {
                    r_d[k] = c.nc[k] * -2.0 * dn + c.dir[k];                       /* surface.rs:246-256 */
                    r_o[k] = c.loc[k] + c.nc[k] * EU_EPS * 128.0;
                }
            }
            if (need_trans) {
                node_kind = need_refl ? WF_COMBINE_TRANS : WF_OVER;
                B.node_px[nid] = spx;
                if (need_refl) B.node_ratio[nid] = ratio;
                B.node_parent[nid] = parent;
                wf_spawn<D>(S, B, counters, gen, child_depth, t_o, t_d, (uint32_t)dest, nid, 0u, cnt);
                if (need_refl) wf_spawn<D>(S, B, counters, gen, child_depth, r_o, r_d, ent, nid, 1u, cnt);
            } else if (need_refl) {
                if (have_inter) {
                    node_kind = WF_COMBINE_INTER;
                    B.node_ratio[nid] = ratio;
                    B.node_parent[nid] = parent;
                    wf_deliver(B, nid, 0u, inter);
                    wf_spawn<D>(S, B, counters, gen, child_depth, r_o, r_d, ent, nid, 1u, cnt);
                } else {   /* the reflection colour is the result (surface.rs:153-154): the child reports to our parent */
                    wf_spawn<D>(S, B, counters, gen, child_depth, r_o, r_d, ent, parent, slot, cnt);
                }
            } else {
                if (!have_inter) cnt.errors++;            /* the reference panics here (surface.rs:154) */
                wf_deliver(B, parent, slot, inter);
            }
        }
        B.node_meta[nid] = node_kind | (slot << 8);
    }
    wf_flush_counters(counters, cnt);
}

/* ------------------------------------------------------------------ bottom-up resolve of one generation's nodes */
template <int D>
__global__ __launch_bounds__(EU_WF_BLOCK) void eu_wf_resolve_kernel(uint32_t gen, EuWfBuffers B, EuDevCounters *counters) {
    LaneCounters cnt = {0, 0, 0, 0};
    const unsigned long long count = counters->gen_count[gen] < B.ray_cap ? counters->gen_count[gen] : B.ray_cap;
    unsigned long long gen_base = B.npix;
    for (uint32_t h = 0; h < gen; h++) gen_base += counters->gen_count[h] < B.ray_cap ? counters->gen_count[h] : B.ray_cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long nid = gen_base + i;
        if (nid >= B.node_cap) continue;
        const uint32_t meta = B.node_meta[nid];
        const uint32_t kind = meta & 0xffu;
        if (kind == WF_NONE) continue;
        const double *ch = B.node_child + (size_t)nid * 8;
        Rgba c0 = {ch[0], ch[1], ch[2], ch[3]};
        Rgba res;
        if (kind == WF_COMBINE_INTER) {                                            /* surface.rs:159-161 */
            const Rgba c1 = {ch[4], ch[5], ch[6], ch[7]};
            res = combine_palette_color(c1, c0, B.node_ratio[nid]);
        } else {
            /* surface_palette.over(transition_palette), both re-quantised to u8 (surface.rs:104-114) */
            const uint32_t tpx = to_pixel4(c0, cnt);
            const Rgba inter = blend_rgba(EU_BL_OVER, new_u8(B.node_px[nid]), new_u8(tpx));
            if (kind == WF_OVER) res = inter;
            else {
                const Rgba c1 = {ch[4], ch[5], ch[6], ch[7]};
                res = combine_palette_color(c1, inter, B.node_ratio[nid]);
            }
        }
        wf_deliver(B, B.node_parent[nid], (meta >> 8) & 1u, res);
    }
    wf_flush_counters(counters, cnt);
}

/* trace_unknown: fg.over(white) un-premultiplied, then Rgb::to_pixel (universe/mod.rs:263-269,342) */
__global__ __launch_bounds__(EU_WF_BLOCK) void eu_wf_final_kernel(EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, double *__restrict__ point_rgb) {
    LaneCounters cnt = {0, 0, 0, 0};
    for (uint32_t p = blockIdx.x * blockDim.x + threadIdx.x; p < B.npix; p += gridDim.x * blockDim.x) {
        if ((B.node_meta[p] & 0xffu) != WF_ROOT) continue;
        const double *ch = B.node_child + (size_t)p * 8;
        const Rgba ret = {ch[0], ch[1], ch[2], ch[3]};
        const Rgba white = {1.0, 1.0, 1.0, 1.0};
        const Rgba out = from_premultiplied(blend_pre(EU_BL_OVER, into_premultiplied(ret), into_premultiplied(white)));
        rgba[p] = to_u8(out.r, cnt) | (to_u8(out.g, cnt) << 8) | (to_u8(out.b, cnt) << 16) | 0xff000000u;
        if (point_rgb) { point_rgb[0] = out.r; point_rgb[1] = out.g; point_rgb[2] = out.b; }
    }
    wf_flush_counters(counters, cnt);
}

#endif
