/*
 * trace_megakernel.h -- persistent-wavefront megakernel variant of the trace loop (one lane = one
 * primary ray at a time, lane refill, explicit per-lane frame stack).  Kept as the A/B reference
 * for the wavefront pipeline (trace_wavefront.h); selected with EU_KERNEL=mega.
 */
#ifndef EU_TRACE_MEGAKERNEL_H
#define EU_TRACE_MEGAKERNEL_H

#include "trace_device.h"

#define EU_BLOCK 256

/* Diagnostic build only (-DEU_PROFILE_PHASES): per-wave s_memtime shares of the kernel's phases, written
 * to EuDevCounters::phase[] (never to an output).  The shipped library is built without it. */
#ifdef EU_PROFILE_PHASES
#define EU_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#define EU_ACC(slot, a, b) ph[slot] += (b) - (a)
#else
#define EU_STAMP(var)
#define EU_ACC(slot, a, b)
#endif

/* ------------------------------------------------------------------ the lane state machine */
template <int D, int HSCAP /* 0: hit stack in LDS (capacity = hs_cap), else private array of HSCAP */, bool SCENE_IN_LDS>
__global__ __launch_bounds__(EU_BLOCK) void eu_trace_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t hs_cap,
                                                            EuDevCamera cam, EuDevFrame fr, EuDevCounters *counters,
                                                            uint32_t *__restrict__ rgba, eu_f64 *__restrict__ hit_t,
                                                            eu_f64 *__restrict__ point_rgb /* single-pixel mode: un-quantised Rgb<F> */) {
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

    typename eu_conditional<HSCAP == 0, HitStackLds, HitStackPriv<(HSCAP ? HSCAP : 1)>>::type HS;
    if constexpr (HSCAP == 0) {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        real *hs_t = (real *)(lds_dyn + lds_words);
        uint32_t *hs_c = (uint32_t *)(hs_t + (EU_BLOCK / 64) * hs_cap * 64);
        HS.t = hs_t + wave * hs_cap * 64 + lane;
        HS.c = hs_c + wave * hs_cap * 64 + lane;
        HS.cap = hs_cap;
    }
    FrameStack<D> FS;
    LaneCounters cnt = {0, 0, 0, 0};
#ifdef EU_PROFILE_PHASES
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif

    const unsigned long long total_items = (unsigned long long)fr.n_tiles * 64ull;
    const uint32_t rows = fr.local_rows;

    /* lane state */
    bool active = false;
    uint32_t px_x = 0, px_y = 0, out_idx = 0;
    real o[D], d[D];
    int ent = 0;
    uint32_t depth = 0, fsp = 0;
    bool primary = false;
    real first_hit = -R(1.0);

    for (;;) {
        /* ---- refill: idle lanes pull the next pixel (wave-aggregated atomic) ---- */
        EU_STAMP(s0);
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
                        if (hit_t) hit_t[ry * fr.width + px_x] = -R(1.0);
                        continue;
                    }
                } else px_y = fr.row_begin + ry;
                out_idx = ry * fr.width + px_x;
            }

            /* Environment::render's cross-hair (universe/mod.rs:321-333) */
            const uint32_t hw = fr.width / 2, hh = fr.height / 2;
            if (fr.debug_crosshair && ((px_x == hw && (px_y == hh - 1 || px_y == hh + 1)) || (px_y == hh && (px_x == hw - 1 || px_x == hw + 1)))) {
                rgba[out_idx] = 0xff0000ffu;
                if (hit_t) hit_t[out_idx] = -R(1.0);
                if (point_rgb) { point_rgb[0] = R(1.0); point_rgb[1] = R(0.0); point_rgb[2] = R(0.0); }
                continue;
            }
            /* camera ray (d3/entity/camera.rs:164-185, d4/entity/camera.rs:155-176) */
            const int sw = (int)fr.width, sh = (int)fr.height;
            const real rel_x = (real)((int)px_x - sw / 2) + (real)(1 - sw % 2) / R(2.0);
            const real rel_y = (real)((int)px_y - sh / 2) + (real)(1 - sh % 2) / R(2.0);
            real dl[D];
#pragma unroll
            for (int i = 0; i < D; i++) {
                const real center = cam.location[i] + cam.forward[i] * cam.dist;
                const real p = center + (cam.up[i] * rel_y) + (cam.right[i] * rel_x);
                dl[i] = p - cam.location[i];
                o[i] = cam.location[i];
            }
            vnormalize<D>(dl, d);
            /* trace_unknown (universe/mod.rs:253-271) */
            ent = material_at<D>(S, o);
            if (ent < 0) {   /* trace_screen_point's checkerboard (universe/mod.rs:387-395) */
                const bool black = (((int)px_x / 8 + (int)px_y / 8) % 2) == 0;
                rgba[out_idx] = black ? 0xff000000u : 0xffff00ffu;
                if (hit_t) hit_t[out_idx] = -R(1.0);
                if (point_rgb) { point_rgb[0] = black ? R(0.0) : R(1.0); point_rgb[1] = R(0.0); point_rgb[2] = black ? R(0.0) : R(1.0); }
                continue;
            }
            material_apply<D>(S, S.entity((uint32_t)ent).material, d, false);
            depth = cam.max_depth;
            fsp = 0;
            primary = true;
            first_hit = -R(1.0);
            active = true;
        }
        if (!active) break;   /* no work left for this lane */
        EU_STAMP(s1); EU_ACC(0, s0, s1);

        /* ---- TRACE one segment: Universe::trace (universe/mod.rs:149-184) ---- */
        Rgba ret = {R(0.0), R(0.0), R(0.0), R(0.0)};
        bool returning = false;
#ifdef EU_PROFILE_PHASES
        unsigned long long s2 = 0;
#endif
        {
            bool have = false;
            real best_t = R(0.0);
            uint32_t best_code = 0, best_ent = 0;
            if (depth > 0) {
                cnt.rays++;
                /* trace_closest (universe/mod.rs:85-147): first hit of every surfaced entity, strict minimum */
                for (uint32_t e = 0; e < S.n_entities; e++) {
                    const EuScene::EntityView E = S.entity(e);
                    if (E.surface < 0) continue;
                    real t = R(0.0); uint32_t code = 0;
                    const uint32_t n = eval_shape<D>(S, E.shape_first, E.shape_root, o, d, HS, cnt, t, code);
                    if (n == 0) continue;
                    if (!have || best_t > t) { have = true; best_t = t; best_code = code; best_ent = e; }
                }
            }
            if (primary) { first_hit = have ? best_t : -R(1.0); primary = false; }
#ifdef EU_PROFILE_PHASES
            s2 = __builtin_amdgcn_s_memtime(); ph[1] += s2 - s1;
#endif
            if (have) {
                /* ComposableSurface::get_color (surface.rs:62-162) */
                HitCtx<D> c;
                c.finish(best_t, o, d);
                hit_normal<D>(S, best_code, o, d, c.loc, c.normal);
                c.classify();
                const EuScene::EntityView HE = S.entity(best_ent);
                const EuFlatSurface *F = S.surface((uint32_t)HE.surface);
                real ratio = reflection_ratio<D>(F, c);
                ratio = rust_max(rust_min(ratio, R(1.0)), R(0.0));

                bool have_inter = false, need_trans = false;
                Rgba inter = {R(0.0), R(0.0), R(0.0), R(0.0)};
                uint32_t spx = 0;
                real t_o[D], t_d[D];
                int dest = -1;
                if (!(ratio >= R(1.0))) {                                              /* get_intersection_color */
                    real cst_priv[16];
                    const Rgba sc = surface_color<D>(S, F, c, fr.time_s, cnt, cst_priv, 1u);
                    spx = to_pixel4(sc, cnt);
                    if ((spx >> 24) == 255u) { inter = sc; have_inter = true; }
                    else {
                        threshold_direction<D>(F, c, t_d);
#pragma unroll
                        for (int i = 0; i < D; i++) t_o[i] = c.loc[i] + -c.nc[i] * EU_EPS * R(128.0);
                        dest = c.exiting ? material_at<D>(S, t_o) : (int)best_ent;
                        if (dest >= 0) {
                            material_apply<D>(S, S.entity((uint32_t)ent).material, t_d, true);
                            material_apply<D>(S, S.entity((uint32_t)dest).material, t_d, false);
                            need_trans = true;
                        }
                    }
                }
                const bool need_refl = !(ratio <= R(0.0));                              /* get_reflection_color */
                real r_o[D], r_d[D];
                if (need_refl) {
                    const real dn = vdot<D>(c.dir, c.nc);
#pragma unroll
                    for (int i = 0; i < D; i++) {
                        r_d[i] = c.nc[i] * -R(2.0) * dn + c.dir[i];                     /* surface.rs:246-256 */
                        r_o[i] = c.loc[i] + c.nc[i] * EU_EPS * R(128.0);
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
                real pt[D];
#pragma unroll
                for (int i = 0; i < D; i++) pt[i] = R(0.0) + d[i];
                ret = mapped_get_color(S, S.background, pt, cnt);
                returning = true;
            }
        }

        EU_STAMP(s3); EU_ACC(2, s2, s3);
        /* ---- RETURN through pending frames ---- */
        while (returning) {
            if (fsp == 0) {
                /* trace_unknown: fg.over(white) un-premultiplied, then Rgb::to_pixel (universe/mod.rs:263-269,342) */
                const Rgba white = {R(1.0), R(1.0), R(1.0), R(1.0)};
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
        EU_STAMP(s4); EU_ACC(3, s3, s4);
    }

#ifdef EU_PROFILE_PHASES
    for (int i = 0; i < 8; i++) {      /* stamps are wave-level: the lane that stayed longest holds the full sums */
        unsigned long long v = ph[i];
        for (int off = 32; off > 0; off >>= 1) { unsigned long long w2 = __shfl_down(v, off); v = w2 > v ? w2 : v; }
        if ((threadIdx.x & 63) == 0) atomicAdd(&counters->phase[i], v);
    }
#endif
    /* ---- counters: wave reduction, one atomic per wave and counter ---- */
    unsigned long long v0 = cnt.rays, v1 = cnt.bg, v2 = cnt.nan_px, v3 = cnt.errors;
    for (int off = 32; off > 0; off >>= 1) {
        v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); v3 += __shfl_down(v3, off);
    }
    if ((threadIdx.x & 63) == 0) {
        v3 = (v3 & 0x3fffffffull) + (v3 >> 30);      /* (EU_CNT_HS_FULL cannot happen on this kernel's strict-size stack; were it to, it is reported) */
        if (v0) atomicAdd(&counters->rays, v0);
        if (v1) atomicAdd(&counters->bg_samples, v1);
        if (v2) atomicAdd(&counters->nan_pixels, v2);
        if (v3) atomicAdd(&counters->errors, v3);
    }
}


#endif
