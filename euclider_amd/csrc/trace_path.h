/*
 * trace_path.h -- Universe::trace_path_unknown on the device: how far and in which direction a
 * moving camera gets when it is pushed `distance` along `direction` through surfaces and materials
 * (portals stretch or squeeze the step, universe/mod.rs:186-227,273-286; surface.rs:164-197;
 * material.rs:54-56,144-146).  One call = one ray = one lane: it reuses the trace loop's
 * intersectors, CSG evaluation and material programs so that a camera walks through exactly the
 * geometry the frames show.  The reference recurses; every level returns its callee's result
 * unchanged, so the device runs it as a loop.
 */
#ifndef EUCLIDER_AMD_TRACE_PATH_H
#define EUCLIDER_AMD_TRACE_PATH_H

#include "trace_device.h"

#define EU_PATH_MAX_STEPS 4096u      /* surface crossings per call; beyond it the call fails (the reference would overflow its stack) */

struct EuPathResult {
    real location[4], direction[4];
    int32_t found;                   /* 1 = Some, 0 = None (no material at the start), -1 = step cap */
    uint32_t steps;                  /* surfaces crossed */
};

template <int D>
__global__ __launch_bounds__(64) void eu_trace_path_kernel(const uint64_t *__restrict__ scene_g, const real *__restrict__ in /* location[D], direction[D], distance */,
                                                           EuPathResult *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    EuScene S;
    S.init(scene_g);
    HitStackPriv<96> HS;
    LaneCounters cnt = {0, 0, 0, 0};
    real o[D], d[D];
#pragma unroll
    for (int k = 0; k < D; k++) { o[k] = in[k]; d[k] = in[D + k]; }
    real distance = in[2 * D];
    EuPathResult res;
    for (int k = 0; k < 4; k++) { res.location[k] = R(0.0); res.direction[k] = R(0.0); }
    res.found = 0; res.steps = 0;
    int ent = material_at<D>(S, o);                                         /* universe/mod.rs:280 */
    if (ent >= 0) {
        material_apply<D>(S, S.entity((uint32_t)ent).material, d, false);  /* enter, :283 */
        res.found = -1;
        for (uint32_t step = 0; step <= EU_PATH_MAX_STEPS; step++) {
            /* trace_closest over surfaced entities (universe/mod.rs:194-196) */
            bool have = false;
            real best_t = R(0.0);
            uint32_t best_code = 0, best_ent = 0;
            for (uint32_t e = 0; e < S.n_entities; e++) {
                const EuScene::EntityView E = S.entity(e);
                if (E.surface < 0) continue;
                if (E.bound != 0xffffffffu && ray_misses_bound<D>(S.bounds(E.bound, D), o, d)) continue;
                real t = R(0.0); uint32_t code = 0;
                const uint32_t n = eval_shape<D>(S, E.shape_first, E.shape_root, o, d, HS, cnt, t, code);
                if (n == 0) continue;
                if (!have || best_t > t) { have = true; best_t = t; best_code = code; best_ent = e; }
            }
            bool moved_on = false;
            if (have && !(distance - best_t <= R(0.0))) {                      /* Surface::get_path, surface.rs:165-167 */
                HitCtx<D> c;
                c.finish(best_t, o, d);
                hit_normal<D>(S, best_code, o, d, c.loc, c.normal);
                c.classify();
                real no[D];
#pragma unroll
                for (int k = 0; k < D; k++) no[k] = c.loc[k] + -c.nc[k] * EU_EPS * R(128.0);
                const int dest = c.exiting ? material_at<D>(S, no) : (int)best_ent;   /* surface.rs:177-185 */
                if (dest >= 0) {
                    material_apply<D>(S, S.entity((uint32_t)ent).material, d, true);     /* exit the origin's material, :188 */
                    material_apply<D>(S, S.entity((uint32_t)dest).material, d, false);   /* enter the destination's, :189 */
                    distance = distance - best_t;
#pragma unroll
                    for (int k = 0; k < D; k++) o[k] = no[k];
                    ent = dest;
                    res.steps = step + 1;
                    moved_on = true;
                }
            }
            if (!moved_on) {                                                 /* Material::trace_path + exit, mod.rs:221-226 */
#pragma unroll
                for (int k = 0; k < D; k++) { res.location[k] = o[k] + d[k] * distance; res.direction[k] = d[k]; }
                material_apply<D>(S, S.entity((uint32_t)ent).material, res.direction, true);
                res.found = 1;
                break;
            }
        }
    }
    *out = res;
}

#endif
