/*
 * trace_wavefront.h -- the trace loop as a wavefront pipeline for gfx950.
 *
 * The reference recurses per pixel: trace -> trace_closest -> get_color -> trace ...
 * (/root/reference/src/universe/mod.rs:149-184, universe/entity/surface.rs:62-162).  Here the recursion
 * tree of the whole frame is processed one GENERATION (= recursion depth) at a time:
 *
 *   for g = 0 .. max_depth-1
 *     intersect kernel    one thread per ray of generation g: Universe::trace_closest -> (t, hit code, entity).
 *                         Generation 0 has no queue: its rays are computed from the pixel index (camera ray,
 *                         material_at, Material::enter) here and again in the shade kernel
 *     shade kernel        one thread per ray: ComposableSurface::get_color up to the recursive calls:
 *                         a finished colour is DELIVERED to the parent's child slot (a primary ray's colour:
 *                         fg.over(white), to_pixel -> the RGBA8 pixel, universe/mod.rs:263-269,342); otherwise a tree
 *                         node {kind, quantised surface pixel, ratio} is created and 1-2 child rays are
 *                         appended to queue g+1 (wave-aggregated LDS counter)
 *   for g = max_depth-1 .. 0
 *     resolve kernel      one thread per node of generation g: over / combine of the delivered child
 *                         colours (surface.rs:104-114,159-161) -> delivered to its own parent
 *
 * The kernels are templates over a scene policy (trace_device.h, EuInterp<D>): the ahead-of-time library instantiates them
 * with the interpreter of the flat scene; jit.cpp instantiates the same bodies with straight-line code generated for one scene.
 *
 * Everything a ray needs between kernels lives in HBM as queues of records (coalesced), sized
 * for the 288 GB part.  A queue is cut into one SEGMENT per producer workgroup: a workgroup appends
 * its children to its own segment through an LDS counter (no global atomics: a single hot counter
 * saturates at ~88 appends/us on this chip and was the first bottleneck), publishes the segment
 * length when it ends, and the next kernel's workgroups walk whole segments.  Queue slot ids are
 * static (segment * segment_capacity + offset), so tree-node ids need no prefix sums either; every kernel is small enough for the register allocator to reach several
 * waves per SIMD, and queues keep all 64 lanes busy whatever the per-pixel ray count is.  The order in
 * which rays are processed does not matter: every step is a pure function, so the result is
 * bit-identical to the depth-first recursion.
 */
#ifndef EU_TRACE_WAVEFRONT_H
#define EU_TRACE_WAVEFRONT_H

#include "trace_device.h"
#include "trace_nodes.h"

#define EU_WF_BLOCK 256
#ifndef EU_WF_WIN
#define EU_WF_WIN 1024        /* rays sorted together in the shade kernel */
#endif
#define EU_WF_KEYS 32
#ifndef EU_ISECT_WAVES
#define EU_ISECT_WAVES 3     /* waves per SIMD the intersect kernel is compiled for */
#endif
#ifndef EU_ISECT_PREFETCH
#define EU_ISECT_PREFETCH 1  /* the next ray's loads are issued before the current ray is intersected */
#endif
#ifndef EU_SHADE_WAVES
#define EU_SHADE_WAVES 3      /* waves per SIMD the shade kernel is compiled for (168 VGPRs) */
#endif
/* Work is DEALT, not divided: every launch gives each wave (intersect) or workgroup (shade) a first, static share and hands out the
 * rest through counters in device memory, in units of one 64-ray batch / one shade window.  With static shares alone the slowest
 * workgroup of a 1080p launch ran 1.5-1.9 times the mean (a few batches per wave, glass rays costing five times a wall ray's
 * instructions) while the rest of the chip idled: profiles/r04_wg_profile_*.txt. */
#ifndef EU_WF_DEAL_ISECT
#define EU_WF_DEAL_ISECT 0      /* intersect kernel: batches beyond the static share are dealt through counters */
#endif
#ifndef EU_WF_DEAL_SHADE
#define EU_WF_DEAL_SHADE 0      /* shade kernel: windows beyond each workgroup's first are dealt through a counter */
#endif
#ifndef EU_WF_STATIC_PCT
#define EU_WF_STATIC_PCT 50   /* intersect: the share of a generation's batches dealt round-robin without asking a counter */
#endif
#ifndef EU_WF_WIN_MIN
#define EU_WF_WIN_MIN 512u
#endif
#ifndef EU_WF_EQUAL_WIN
#define EU_WF_EQUAL_WIN 1
#endif
#ifndef EU_SHADE_TAKE_CHUNKS
#define EU_SHADE_TAKE_CHUNKS 1
#endif
#ifndef EU_WF_SPREAD
#define EU_WF_SPREAD 0        /* 1: a shade window is made of 256-ray pieces from equally spaced places of the generation's queue (each workgroup a sample of the whole) */
#endif
#ifndef EU_WF_DEAL_FACTOR
#define EU_WF_DEAL_FACTOR 4u
#endif
#define EU_WORK_STRIDE 32u                       /* words between two work counters: each on a 128-byte line of its own (same-address atomics drain at ~88 per microsecond) */
#ifndef EU_WORK_SHARDS
#define EU_WORK_SHARDS 64u                       /* intersect: the dealt batches form this many interleaved classes with a counter each (a power of two).  512 waves on one
                                                  * counter, a claim per ~6 us batch each, is the ~88 claims per microsecond at which one word saturates: with 8 classes the
                                                  * claims' latency grew beyond the batch that was to hide it */
#endif
#define EU_WORK_WINDOWS (EU_WORK_SHARDS * EU_WORK_STRIDE)             /* the shade kernel's window counter */
#define EU_WORK_TOTAL ((EU_WORK_SHARDS + 1u) * EU_WORK_STRIDE)      /* rays queued for this generation (summed by the producers: one atomic per workgroup that queued any) */
#define EU_WORK_SLOTS (EU_WORK_SHARDS + 2u)
#define EU_WORK_PER_GEN (EU_WORK_SLOTS * EU_WORK_STRIDE)

#if defined(EU_PROFILE_SHADE_WAVE)      /* diagnostic build: wave-level shares of the shade kernel (LDS rows, first active lane: trace_device.h SHP): STAMP(k) closes
                                         * section k-1 (STAMP(0): what lies between two batches -> 15), SUB(k) closes sub-section k; sections do not overlap */
#define WF_STAMP(k) SHP(cnt, (k) > 0 ? (k) - 1 : 15)
#define WF_SUB(k) SHP(cnt, (k))
#else
#define WF_STAMP(k) do { } while (0)
#define WF_SUB(k) do { } while (0)
#endif


/* Queue records are arrays of structures: a ray is ONE 2D-value record, its hit ONE 16-byte record.  As structure of arrays (one
 * stream per component: 6 + 2 + 3 streams for D = 3) every vector memory instruction of a ray batch went to a different page, 1.5 MB
 * further on than in the batch before, and ISSUING the six component loads took a wave 16 K cycles per batch (`-DEU_PROFILE_SHAPE`: 41 %
 * of the generation-0 intersect kernel; with the same bytes from consecutive words: 11 %). */
template <int D> struct alignas(2 * sizeof(real)) EuWfRay { real o[D], d[D]; };
struct alignas(16) EuWfHit {
    real t; uint32_t code, ent;      /* ent 0xffffffff: nothing hit */
#if EU_REAL_BITS == 32
    uint32_t pad;
#endif
};
static_assert(sizeof(EuWfHit) == 16, "one 16-byte store per ray");
struct EuWfBuffers {
    real *ray[2];             /* [ray_cap] EuWfRay<D>: origin then direction; ping-pong by generation */
    uint2 *ray_pa[2];           /* x: node id that receives this ray's colour; y: entity the ray travels in (bits 0..15) | delivery slot / mode (bits 16..18) */
    EuWfHit *hit[2];            /* per ray of a generation, ping-pong by generation like the rays (the fused kernel writes generation g + 1's while other
                                 * workgroups still read generation g's) */
    /* tree nodes (trace_nodes.h): one id per traced ray in queue order, only the slots of rays that need one are touched */
    EuTsNode *nodes;            /* [node_cap] */
    uint8_t *node_kind;         /* [node_cap] TS_NONE / TS_OVER / ...: what resolve has to do for the ray in this slot */
    uint32_t *seg_count;        /* [EU_MAX_DEPTH + 1][n_seg] rays in each segment of each generation's queue */
    uint32_t ray_cap, node_cap, npix, pad;      /* pad: the band pipeline's number (diagnostics) */
    uint32_t n_seg, seg_cap;    /* ray_cap = n_seg * seg_cap; node id of queue slot q of generation g = g * ray_cap + q */
    uint32_t *work;             /* [EU_MAX_DEPTH + 1][EU_WORK_PER_GEN] work counters of this band pipeline, zeroed before every frame */
    unsigned long long *prof;   /* diagnostics (-DEU_PROFILE_WG): [0] = records written, then 4 words per workgroup and launch; else NULL */
};

/* -DEU_PROFILE_WG: every workgroup of every launch leaves {kind | gen << 8 | band << 16 | block << 32, start, end, XCC | CU ids} (100 MHz clock):
 * tools/wg_profile.py turns them into the share of the chip a launch keeps busy (tails, imbalance). */
#ifdef EU_PROFILE_WG
#define EU_PROF_CAP (1u << 20)
#define WF_PROF_BEGIN() const unsigned long long wf_prof_t0 = wall_clock64()
#define WF_PROF_END(B_, kind_, gen_) do { if (threadIdx.x == 0 && (B_).prof) { const unsigned long long i_ = atomicAdd((B_).prof, 1ull); if (i_ < EU_PROF_CAP) { \
    unsigned long long *r_ = (B_).prof + 4 + 4 * i_; r_[0] = (unsigned long long)(kind_) | ((unsigned long long)(gen_) << 8) | ((unsigned long long)((B_).pad & 0xffu) << 16) | ((unsigned long long)blockIdx.x << 32); \
    r_[1] = wf_prof_t0; r_[2] = wall_clock64(); r_[3] = (unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)gridDim.x << 32); } } } while (0)
#else
#define WF_PROF_BEGIN() do { } while (0)
#define WF_PROF_END(B_, kind_, gen_) do { } while (0)
#endif

EU_DEV void wf_flush_counters(EuDevCounters *counters, const LaneCounters &cnt) {
    /* one atomic per counter and WORKGROUP: same-address atomics drain at ~90 per microsecond, a per-wave flush of a
     * 768-workgroup launch kept the kernel alive for tens of microseconds after its last ray.  Called by every thread. */
    __shared__ unsigned long long wg_cnt[4];
    if (threadIdx.x < 4) wg_cnt[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long v0 = cnt.rays, v1 = cnt.bg, v2 = cnt.nan_px, v3 = cnt.errors;
    for (int off = 32; off > 0; off >>= 1) {
        v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); v3 += __shfl_down(v3, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (v3 >> 30) atomicAdd(&counters->hs_full, v3 >> 30);      /* lanes whose hit stack was full (EU_CNT_HS_FULL): the frame is traced again */
        v3 &= 0x3fffffffull;
        if (v0) atomicAdd(&wg_cnt[0], v0);
        if (v1) atomicAdd(&wg_cnt[1], v1);
        if (v2) atomicAdd(&wg_cnt[2], v2);
        if (v3) atomicAdd(&wg_cnt[3], v3);
    }
    __syncthreads();
#if defined(EU_PROFILE_SHAPE) || defined(EU_PROFILE_SHADE_WAVE)
    if ((threadIdx.x & 63) == 0 && cnt.prof) for (int q = 0; q < 16; q++) if (cnt.prof[q]) atomicAdd(&counters->phase[q], cnt.prof[q]);
#endif
    if (threadIdx.x == 0) {
        if (wg_cnt[0]) atomicAdd(&counters->rays, wg_cnt[0]);
        if (wg_cnt[1]) atomicAdd(&counters->bg_samples, wg_cnt[1]);
        if (wg_cnt[2]) atomicAdd(&counters->nan_pixels, wg_cnt[2]);
        if (wg_cnt[3]) atomicAdd(&counters->errors, wg_cnt[3]);
    }
}

/* append to this workgroup's output segment: wave-aggregated add on an LDS counter */
EU_DEV uint32_t wf_append_local(uint32_t *lds_counter, uint32_t n /* 0..2 slots wanted by this lane */, uint32_t &second) {
    const unsigned long long m1 = __ballot(n >= 1), m2 = __ballot(n >= 2);
    const uint32_t lo = (uint32_t)m1, hi = (uint32_t)(m1 >> 32), lo2 = (uint32_t)m2, hi2 = (uint32_t)(m2 >> 32);
    const uint32_t rank1 = __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u));
    const uint32_t rank2 = __builtin_amdgcn_mbcnt_hi(hi2, __builtin_amdgcn_mbcnt_lo(lo2, 0u));
    const uint32_t total = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0) base = atomicAdd(lds_counter, total);
    base = __builtin_amdgcn_readfirstlane(base);
    /* first slots of all lanes, then second slots */
    second = base + (uint32_t)__popcll(m1) + rank2;
    return base + rank1;
}

/* Balanced consumption of a segmented queue: every workgroup scans the (<= 1024) segment lengths of
 * the generation into LDS; the rays then form one virtual index space that is dealt out grid-stride,
 * and a lane maps its virtual index back to (segment, offset) with a binary search in LDS. */
#ifndef EU_WF_MAX_SEG
#define EU_WF_MAX_SEG 1024      /* a power of two (wf_map_index halves its step from here); the table is padded with sentinels up to it */
#endif
EU_DEV uint32_t wf_build_prefix(const uint32_t *seg_count, uint32_t n_seg, uint32_t *pref /* LDS, EU_WF_MAX_SEG + 1 words */, uint32_t *wave_tot /* LDS, 4 words */) {
    const uint32_t t = threadIdx.x, lane = t & 63, wave = t >> 6;
    uint32_t v[4], s = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { const uint32_t idx = t * 4 + k; v[k] = idx < n_seg ? seg_count[idx] : 0u; s += v[k]; }
    uint32_t inc = s;                                  /* inclusive scan inside the wave */
    for (int off = 1; off < 64; off <<= 1) { const uint32_t y = __shfl_up(inc, off); if ((int)lane >= off) inc += y; }
    if (lane == 63) wave_tot[wave] = inc;
    __syncthreads();
    uint32_t base = 0;
    for (uint32_t w = 0; w < wave; w++) base += wave_tot[w];
    uint32_t run = base + inc - s;                     /* exclusive prefix of this thread's first element */
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) { const uint32_t idx = t * 4 + k; if (idx < n_seg) pref[idx] = run; else if (idx > n_seg) pref[idx] = 0xffffffffu; run += v[k]; }
    const uint32_t total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    if (t == 0) pref[n_seg] = total;      /* (entries behind n_seg: sentinels for wf_map_index, which reads indices < EU_WF_MAX_SEG only) */
    __syncthreads();
    return total;
}
/* queue slot of ray number v < total: the largest seg with pref[seg] <= v, found with a fixed number of steps and no branch (the table
 * is padded with 0xffffffff up to EU_WF_MAX_SEG: half the instructions of a bisection loop with its exit test, once per ray) */
EU_DEV uint32_t wf_map_index(const uint32_t *pref, uint32_t n_seg, uint32_t seg_cap, uint32_t v) {
    static_assert((EU_WF_MAX_SEG & (EU_WF_MAX_SEG - 1)) == 0, "power of two");
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = EU_WF_MAX_SEG / 2; step != 0; step >>= 1) lo += pref[lo + step] <= v ? step : 0u;
    return lo * seg_cap + (v - pref[lo]);
}

/* ------------------------------------------------------------------ dealing 64-ray batches to waves (intersect kernel)
 * Batch c covers rays [64 c, 64 c + 64) of the generation.  The first `s_end` batches are dealt round-robin (wave w takes w,
 * w + n_waves, ...: no counter involved); a wave that has finished its static share claims the others one at a time (one claim is
 * always in flight while a batch is traced, so its latency is hidden).  Any partition of the rays gives the same frame. */
#define EU_WF_NO_BATCH 0xffffffffu
struct WfDeal {
    uint32_t n_waves, s_end, c_static, n_dyn, shard, len, pending;
    uint32_t *ctr;
    bool has_pending;
    EU_DEV void init(uint32_t n_batches, uint32_t *work_gen) {
        const uint32_t wave = blockIdx.x * (EU_WF_BLOCK / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
        n_waves = gridDim.x * (EU_WF_BLOCK / 64);
        s_end = n_batches;
#if EU_WF_DEAL_ISECT
        if (n_batches > n_waves) {
            uint32_t per = (n_batches / n_waves) * EU_WF_STATIC_PCT / 100u;
            if (per < 1u) per = 1u;
            s_end = per * n_waves;
        }
#endif
        c_static = wave;
        n_dyn = n_batches - s_end;
        /* the dealt batches are cut into EU_WORK_SHARDS interleaved classes (batch s_end + 8 j + x belongs to class x), each behind a
         * counter on a cache line of its own, and a workgroup draws from class blockIdx % 8 only: neighbouring workgroups (dealt
         * round-robin over the XCDs) use different counters, every class samples the whole generation, and a wave whose class is
         * used up is done -- no wave ever goes looking for another counter (when that was allowed, all 4096 waves of a launch ended
         * it by queueing on the same last counter: 40 us of same-address atomics per launch). */
        shard = blockIdx.x & (EU_WORK_SHARDS - 1u);
        len = n_dyn > shard ? (n_dyn - shard + EU_WORK_SHARDS - 1u) / EU_WORK_SHARDS : 0u;
        ctr = work_gen + shard * EU_WORK_STRIDE;
        has_pending = false; pending = 0;
    }
    EU_DEV uint32_t claim() const {      /* (every lane returns lane 0's ticket) */
        uint32_t j = 0;
        if ((threadIdx.x & 63u) == 0) j = atomicAdd(ctr, 1u);
        return j;
    }
    /* the next batch of this wave, or EU_WF_NO_BATCH; wave-uniform */
    EU_DEV uint32_t next() {
        if (c_static < s_end) { const uint32_t c = c_static; c_static += n_waves; return c; }
#if EU_WF_DEAL_ISECT
        if (len == 0) return EU_WF_NO_BATCH;
        const uint32_t j = __builtin_amdgcn_readfirstlane(has_pending ? pending : claim());
        if (j < len) {
            pending = claim();      /* for the call after this one: answered while this batch is traced */
            has_pending = true;
            return s_end + j * EU_WORK_SHARDS + shard;
        }
        len = 0;
#endif
        return EU_WF_NO_BATCH;
    }
};

/* ------------------------------------------------------------------ queue helpers */
struct WfRay { uint32_t q; };

template <int D> EU_DEV void wf_store_ray(const EuWfBuffers &B, uint32_t buf, uint32_t q, const real *o, const real *d, uint32_t parent, uint32_t aux) {
    EuWfRay<D> r;
#pragma unroll
    for (int k = 0; k < D; k++) { r.o[k] = o[k]; r.d[k] = d[k]; }
    ((EuWfRay<D> *)B.ray[buf])[q] = r;
    B.ray_pa[buf][q] = make_uint2(parent, aux);
}
template <int D> EU_DEV void wf_load_ray(const EuWfBuffers &B, uint32_t buf, uint32_t q, real *o, real *d) {
    const EuWfRay<D> r = ((const EuWfRay<D> *)B.ray[buf])[q];
#pragma unroll
    for (int k = 0; k < D; k++) { o[k] = r.o[k]; d[k] = r.d[k]; }
}

/* ------------------------------------------------------------------ primary rays
 * Generation 0 has no queue: its "rays" are the band's pixel items (8x8 tiles, 64 items each: coherent waves), item v sits in
 * slot v, and both the intersect and the shade kernel compute the camera ray from the item number (rounds 1-2 had a kernel of
 * its own write 64 bytes per pixel that the next two kernels read back). */
enum { EU_PR_NONE = 0, EU_PR_PAD = 1, EU_PR_CROSS = 2, EU_PR_CHECKER = 3, EU_PR_RAY = 4 };
#define EU_WF_ENT_MISS 0xffffffffu       /* EuWfHit::ent: the ray hit nothing */
#define EU_WF_ENT_SPECIAL 0xfffffffeu    /* generation 0: a pixel without a ray (padding row, cross-hair, checkerboard): the shade kernel writes it */
#define EU_WF_ENT_DEAD 0xfffffffdu       /* generation 0: an item outside the band */
struct EuPrimary { uint32_t status, out_idx, px_x, px_y; };

template <int D, class P>
EU_DEV EuPrimary wf_primary_ray(const EuScene &S, const EuDevCamera &cam, const EuDevFrame &fr, int cam_ent, uint32_t item, real *o, real *d) {
    EuPrimary pr = {EU_PR_NONE, 0u, 0u, 0u};
    uint32_t ry;
    if (fr.single_pixel) {      /* Environment::trace_screen_point: exactly one item */
        if (item != 0) return pr;
        pr.px_x = fr.single_x; pr.px_y = fr.single_y; ry = 0; pr.out_idx = 0;
    } else {
        const uint32_t tile = item >> 6, within = item & 63u;   /* 8x8 pixel tiles: coherent waves */
        pr.px_x = (tile % fr.tiles_x) * 8 + (within & 7);
        if (fr.band_stride > 1) {      /* concurrent bands of one frame take 8-row groups in turn: equal shares of whatever the picture shows */
            ry = ((tile / fr.tiles_x) * fr.band_stride + fr.band_row0) * 8 + (within >> 3);
            if (pr.px_x >= fr.width || ry >= fr.local_rows) return pr;
        } else {
            ry = fr.band_row0 + (tile / fr.tiles_x) * 8 + (within >> 3);
            if (pr.px_x >= fr.width || ry >= fr.local_rows || ry >= fr.band_row0 + fr.band_rows) return pr;
        }
        pr.out_idx = ry * fr.width + pr.px_x;
        if (fr.strip_count > 1) {   /* interleaved 8-row strips: this rank owns strips s with s % count == index */
            const uint32_t gstrip = (ry >> 3) * fr.strip_count + fr.strip_index;
            pr.px_y = fr.row_begin + gstrip * 8 + (ry & 7);
            if (pr.px_y >= fr.row_end) { pr.status = EU_PR_PAD; return pr; }   /* padding rows of the last strip: defined contents */
        } else pr.px_y = fr.row_begin + ry;
    }
    /* Environment::render's cross-hair (universe/mod.rs:321-333) */
    const uint32_t hw = fr.width / 2, hh = fr.height / 2;
    if (fr.debug_crosshair && ((pr.px_x == hw && (pr.px_y == hh - 1 || pr.px_y == hh + 1)) || (pr.px_y == hh && (pr.px_x == hw - 1 || pr.px_x == hw + 1)))) {
        pr.status = EU_PR_CROSS;
        return pr;
    }
    /* trace_unknown: no material at the camera -> trace_screen_point's checkerboard (universe/mod.rs:253-271,387-395) */
    if (cam_ent < 0) { pr.status = EU_PR_CHECKER; return pr; }
    /* camera ray (d3/entity/camera.rs:164-185, d4/entity/camera.rs:155-176) */
    const int sw = (int)fr.width, sh = (int)fr.height;
    const real rel_x = (real)((int)pr.px_x - sw / 2) + (real)(1 - sw % 2) / R(2.0);
    const real rel_y = (real)((int)pr.px_y - sh / 2) + (real)(1 - sh % 2) / R(2.0);
    real dl[D];
#pragma unroll
    for (int i = 0; i < D; i++) {
        const real center = cam.location[i] + cam.forward[i] * cam.dist;
        const real p = center + (cam.up[i] * rel_y) + (cam.right[i] * rel_x);
        dl[i] = p - cam.location[i];
        o[i] = cam.location[i];
    }
    vnormalize<D>(dl, d);
    P::material_apply(S, (uint32_t)cam_ent, d, false);
    pr.status = EU_PR_RAY;
    return pr;
}

/* ------------------------------------------------------------------ trace_closest */
/* G0: the kernel of generation 0 (rays from the pixel index; `gen` is 0, camera and frame are used) is a separate instantiation:
 * the 58 SGPRs of camera + frame and the primary-ray code (Material::enter: the interpreter's RPN machine) would otherwise weigh on
 * the kernels of every generation, and the intersect loop is out of SGPRs as it is. */
template <int D, int HSCAP /* 0: per-lane hit stack in LDS (capacity hs_cap); else a private array of HSCAP entries */, class P, bool G0>
EU_DEV void wf_intersect_body(const uint64_t *__restrict__ scene_g, uint32_t hs_cap, uint32_t gen, const EuDevCamera &cam, const EuDevFrame &fr,
                              const EuWfBuffers &B, EuDevCounters *counters, eu_f64 *__restrict__ hit_t_aov, uint64_t *lds_dyn) {
    constexpr bool g0 = G0;
    WF_PROF_BEGIN();
    /* an empty generation (3d_hallways at depth 12: the last five) costs a launch, not a prologue: one scalar load says so */
    if (!g0 && B.work[gen * EU_WORK_PER_GEN + EU_WORK_TOTAL] == 0u) { WF_PROF_END(B, 0, gen); return; }
    EuScene S;
    S.init(scene_g);      /* wave-uniform addresses: the scene arrives through scalar loads */
    typename eu_conditional<HSCAP == 0, HitStackLds, HitStackPriv<(HSCAP ? HSCAP : 1)>>::type HS;
    if constexpr (HSCAP == 0) {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        real *hs_t = (real *)(lds_dyn);
        uint32_t *hs_c = (uint32_t *)(hs_t + (EU_WF_BLOCK / 64) * hs_cap * 64);
        HS.t = hs_t + wave * hs_cap * 64 + lane;
        HS.c = hs_c + wave * hs_cap * 64 + lane;
        HS.cap = hs_cap;
    }
    LaneCounters cnt = {0, 0, 0, 0};
#ifdef EU_PROFILE_SHAPE
    __shared__ unsigned long long prof_rows[EU_WF_BLOCK / 64][17];
    cnt.prof = prof_rows[threadIdx.x >> 6];
    if ((threadIdx.x & 63) < 16) cnt.prof[threadIdx.x & 63] = 0;
    if ((threadIdx.x & 63) == 16) cnt.prof[16] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_wave_barrier();
#endif
    const uint32_t in = gen & 1u;
    __shared__ uint32_t pref[EU_WF_MAX_SEG + 1];
    __shared__ uint32_t wave_tot[4];
    int cam_ent = -1;
    uint32_t total;
    if (g0) {      /* every camera ray starts at the camera (get_ray_point, d3/entity/camera.rs:147-153): material_at(origin) is one value per frame */
        cam_ent = P::material_at(S, cam.location);
        total = fr.n_tiles * 64u;
    } else total = wf_build_prefix(B.seg_count + gen * B.n_seg, B.n_seg, pref, wave_tot);
    SHP(cnt, 8);      /* kernel prologue: scene header, queue prefix */
    const bool trace_any = !g0 || cam.max_depth != 0;      /* trace() with depth 0 goes straight to the background: generation 0 only marks the pixels */
    {
        /* software pipeline: the next batch of this wave is asked for, its rays are located and their loads issued before the current
         * batch is intersected (the kernel keeps 3 waves per SIMD: too few to hide an HBM round trip behind other waves) */
        const uint32_t lane = threadIdx.x & 63u;
        WfDeal deal;
        deal.init((total + 63u) >> 6, B.work + gen * EU_WORK_PER_GEN);
        uint32_t c_next = deal.next();
        uint32_t i_next = 0;
        real o_next[D], d_next[D];
#pragma unroll
        for (int k = 0; k < D; k++) { o_next[k] = R(0.0); d_next[k] = R(0.0); }
        if (!g0 && c_next != EU_WF_NO_BATCH && c_next * 64u + lane < total) {
            i_next = wf_map_index(pref, B.n_seg, B.seg_cap, c_next * 64u + lane);
            wf_load_ray<D>(B, in, i_next, o_next, d_next);
        }
        while (c_next != EU_WF_NO_BATCH) {
            const uint32_t v = c_next * 64u + lane;
            uint32_t i = i_next;
            real o[D], d[D];
            bool live = v < total;      /* (the generation's last batch may be ragged) */
            uint32_t out_idx = 0;
#if EU_ISECT_PREFETCH
            c_next = deal.next();
#endif
            if (g0) {
                i = v;
                if (live) {
                const EuPrimary pr = wf_primary_ray<D, P>(S, cam, fr, cam_ent, v, o, d);
                out_idx = pr.out_idx;
                if (pr.status != EU_PR_RAY) {
                    EuWfHit h;
                    h.t = R(0.0); h.code = 0; h.ent = pr.status == EU_PR_NONE ? EU_WF_ENT_DEAD : EU_WF_ENT_SPECIAL;
#if EU_REAL_BITS == 32
                    h.pad = 0;
#endif
                    B.hit[in][i] = h;
                    live = false;
                }
                }
            } else {
#pragma unroll
                for (int k = 0; k < D; k++) { o[k] = o_next[k]; d[k] = d_next[k]; }
#if EU_ISECT_PREFETCH
                if (c_next != EU_WF_NO_BATCH && c_next * 64u + lane < total) {
                    i_next = wf_map_index(pref, B.n_seg, B.seg_cap, c_next * 64u + lane);
                    wf_load_ray<D>(B, in, i_next, o_next, d_next);
                }
#endif
            }
            SHP(cnt, 9);      /* per batch: current ray out of the prefetch registers, next ray located and requested */
            if (live) {
            /* trace_closest (universe/mod.rs:85-147): first hit of every surfaced entity, strict minimum.  Box chains are answered
             * in closed form (chain_slab), and where that refuses a ray (a near-tie of two plane hits, an origin in a face plane, a
             * NaN ray) by the generic chain routine at once: eval_chain. */
            bool have = false;
            real best_t = R(0.0);
            uint32_t best_code = 0, best_ent = EU_WF_ENT_MISS;
            if (trace_any) {
                cnt.rays++;
#ifdef EU_NO_SLAB     /* A/B diagnostic (bench.py --jit-flags=-DEU_NO_SLAB), round 2's route: a wave whose rays are all regular (finite, no zero
                       * direction component) evaluates box chains with one product per dot product (chain_matrices_box); should a lane then
                       * report a non-finite hit point, the wave's rays are traced once more the generic way */
                int use_box = __ballot(!ray_is_regular<D>(o, d)) == 0ull ? 1 : 0;
                for (;;) {
                    bool fail = false;
                    LaneCounters c1 = cnt;
                    have = false; best_t = R(0.0); best_code = 0; best_ent = EU_WF_ENT_MISS;
                    P::trace_closest(S, o, d, HS, c1, use_box, fail, have, best_t, best_code, best_ent);
                    if (__ballot(fail) == 0ull) { cnt = c1; break; }
                    use_box = 0;
                }
#else
                bool fail = false;      /* (only set on the route above) */
                P::trace_closest(S, o, d, HS, cnt, 2, fail, have, best_t, best_code, best_ent);
#endif
            }
            {
                EuWfHit h;
                h.t = best_t; h.code = best_code; h.ent = best_ent;
#if EU_REAL_BITS == 32
                h.pad = 0;
#endif
                B.hit[in][i] = h;
            }
            if (g0 && hit_t_aov) hit_t_aov[out_idx] = have ? best_t : -R(1.0);
            }
#if !EU_ISECT_PREFETCH
            c_next = deal.next();
            if (!g0 && c_next != EU_WF_NO_BATCH && c_next * 64u + lane < total) {
                i_next = wf_map_index(pref, B.n_seg, B.seg_cap, c_next * 64u + lane);
                wf_load_ray<D>(B, in, i_next, o_next, d_next);
            }
#endif
            SHP(cnt, 10);     /* per batch: entity loop's end, result store */
        }
    }
    wf_flush_counters(counters, cnt);
    WF_PROF_END(B, 0, gen);
}

template <int D, int HSCAP>
__global__ __launch_bounds__(EU_WF_BLOCK, EU_ISECT_WAVES) void eu_wf_intersect_kernel(const uint64_t *__restrict__ scene_g, uint32_t hs_cap, uint32_t gen,
                                                                      EuWfBuffers B, EuDevCounters *counters) {
    extern __shared__ uint64_t lds_dyn[];
    const EuDevCamera cam = {};
    const EuDevFrame fr = {};
    wf_intersect_body<D, HSCAP, EuInterp<D>, false>(scene_g, hs_cap, gen, cam, fr, B, counters, nullptr, lds_dyn);
}
template <int D, int HSCAP>
__global__ __launch_bounds__(EU_WF_BLOCK, EU_ISECT_WAVES) void eu_wf_intersect0_kernel(const uint64_t *__restrict__ scene_g, uint32_t hs_cap, EuDevCamera cam, EuDevFrame fr,
                                                                       EuWfBuffers B, EuDevCounters *counters, eu_f64 *__restrict__ hit_t_aov) {
    extern __shared__ uint64_t lds_dyn[];
    wf_intersect_body<D, HSCAP, EuInterp<D>, true>(scene_g, hs_cap, 0u, cam, fr, B, counters, hit_t_aov, lds_dyn);
}

/* ------------------------------------------------------------------ ComposableSurface::get_color up to the recursive calls */
/* FUSE: the kernel goes on to INTERSECT the rays it has just queued (generation g + 1: they sit in this workgroup's own segment), so
 * that a generation costs one launch instead of two -- one prologue, one kernel tail -- and the chain of dependent launches of a
 * frame is a third shorter (a 64x64 frame, or 3d_hallways' thin late generations, are nothing but that chain).  FUSE = -1: no; else
 * the hit stack of the intersect part (0: in LDS with capacity hs_cap, over the dynamic LDS the shading part no longer needs; > 0: a
 * private array of that many entries). */
template <int D, bool SCENE_LDS, class P, bool G0, int FUSE = -1>
EU_DEV void wf_shade_body(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t gen, uint32_t max_depth, real time_s, const EuDevCamera &cam, const EuDevFrame &fr,
                          const EuWfBuffers &B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ hit_t_aov, eu_f64 *__restrict__ point_rgb,
                          uint64_t *lds_dyn, uint32_t hs_cap = 0) {
    constexpr bool g0 = G0;
    WF_PROF_BEGIN();
    if (!g0 && B.work[gen * EU_WORK_PER_GEN + EU_WORK_TOTAL] == 0u) {      /* an empty generation: this workgroup's output segment is empty too */
        if (threadIdx.x == 0) B.seg_count[(gen + 1) * B.n_seg + blockIdx.x] = 0u;
        WF_PROF_END(B, 1, gen);
        return;
    }
    __shared__ uint32_t seg_fill;
    if (threadIdx.x == 0) seg_fill = 0;
    __syncthreads();
    /* The hit entity differs from lane to lane, so the interpreter reads the surface / colour-program records with per-lane
     * addresses: from a copy of the flat scene in LDS (the L1 is swept by the ray streams, a global read of the
     * scene usually goes to L2).  (A scene-specialised policy has the records as constants.) */
    const uint64_t *scene_base = scene_g;
    real *color_stack = (real *)lds_dyn;       /* surface_color's operand stack: color_depth RGBA entries per lane (dynamic LDS) */
    if constexpr (SCENE_LDS) {
        for (uint32_t i = threadIdx.x; i < scene_words; i += EU_WF_BLOCK) lds_dyn[i] = scene_g[i];
        __syncthreads();
        scene_base = lds_dyn;
        color_stack = (real *)(lds_dyn + scene_words);
    }
    EuScene S;
    S.init(scene_base);
    S.wrt = scene_g;
    LaneCounters cnt = {0, 0, 0, 0};
#ifdef EU_PROFILE_SHADE_WAVE
    __shared__ unsigned long long prof_rows[EU_WF_BLOCK / 64][17];
    cnt.prof = prof_rows[threadIdx.x >> 6];
    if ((threadIdx.x & 63) < 16) cnt.prof[threadIdx.x & 63] = 0;
    if ((threadIdx.x & 63) == 16) cnt.prof[16] = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_wave_barrier();
#endif
    const uint32_t in = gen & 1u, outb = (gen + 1) & 1u;
    const uint32_t child_depth = max_depth - gen - 1;
    const bool last_gen = child_depth == 0;
    const uint32_t out_base = blockIdx.x * B.seg_cap;
    const uint32_t node_base = gen * B.ray_cap;       /* node id of queue slot q: node_base + q */
    __shared__ uint32_t pref[EU_WF_MAX_SEG + 1];
    __shared__ uint32_t wave_tot[4];
    int cam_ent = -1;
    uint32_t total;
    if (g0) { cam_ent = P::material_at(S, cam.location); total = fr.n_tiles * 64u; }
    else total = wf_build_prefix(B.seg_count + gen * B.n_seg, B.n_seg, pref, wave_tot);
    /* Rays are taken in windows of EU_WF_WIN per workgroup and counting-sorted in LDS by the entity they
     * hit, so that a wave shades (mostly) one surface: a wall ray costs ~500 instructions, a glass ray
     * (Fresnel + Snell + rotation) ~2500, and unsorted they would share waves. */
    __shared__ uint32_t sorted[EU_WF_WIN];
    __shared__ uint32_t hist[EU_WF_KEYS], offs[EU_WF_KEYS];
    __shared__ uint32_t n_sorted, next_win, next_chunk;
    {
        /* the window: every workgroup gets the same number of equally large windows (multiples of one batch; the counting sort's
         * arrays hold EU_WF_WIN rays) -- with windows of a fixed 2048 rays a generation of 1.1 M rays was 557 windows for 768 workgroups,
         * a quarter of which had nothing to do while the others shaded 2048 rays each.  Where windows are dealt they are made
         * smaller, EU_WF_DEAL_FACTOR per workgroup, down to EU_WF_WIN_MIN: the window is the unit in which the work is balanced */
        uint32_t win;
        {
            const uint32_t per_wg = (total + gridDim.x - 1) / gridDim.x;
#if EU_WF_DEAL_SHADE
            uint32_t target = (per_wg + EU_WF_DEAL_FACTOR - 1) / EU_WF_DEAL_FACTOR;
            if (target < EU_WF_WIN_MIN) target = EU_WF_WIN_MIN < per_wg ? EU_WF_WIN_MIN : per_wg;
#else
            const uint32_t target = per_wg;
#endif
            const uint32_t k = (target + EU_WF_WIN - 1) / EU_WF_WIN;      /* windows the target is cut into */
            win = (((target + (k ? k : 1u) - 1) / (k ? k : 1u)) + EU_WF_BLOCK - 1) / EU_WF_BLOCK * EU_WF_BLOCK;
            if (win < EU_WF_BLOCK) win = EU_WF_BLOCK;
#if !EU_WF_EQUAL_WIN      /* A/B: rounds 1-3's rule -- EU_WF_WIN, halved while the launch would leave workgroups without a window */
            win = EU_WF_WIN;
            while (win > EU_WF_BLOCK && total <= gridDim.x * (win / 2)) win >>= 1;
#endif
        }
        const uint32_t n_win = (total + win - 1) / win;
#if EU_WF_DEAL_SHADE
        const bool dealt = n_win > gridDim.x;
        uint32_t *win_ctr = B.work + gen * EU_WORK_PER_GEN + EU_WORK_WINDOWS;
#endif
        for (uint32_t w = blockIdx.x; w < n_win;) {
#if EU_WF_DEAL_SHADE
            uint32_t ticket = 0;
            if (dealt && threadIdx.x == 0) ticket = atomicAdd(win_ctr, 1u);
#endif
            if (threadIdx.x < EU_WF_KEYS) hist[threadIdx.x] = 0;
            __syncthreads();
            uint32_t myq[EU_WF_WIN / EU_WF_BLOCK], mykey[EU_WF_WIN / EU_WF_BLOCK], myrank[EU_WF_WIN / EU_WF_BLOCK];
#pragma unroll
            for (uint32_t k = 0; k < EU_WF_WIN / EU_WF_BLOCK; k++) {
                const uint32_t kb = k * EU_WF_BLOCK;
#if EU_WF_SPREAD      /* window w = every n_win-th 256-ray piece of the generation's queue, starting with piece w (the arithmetic is scalar) */
                const uint32_t v = (w + k * n_win) * EU_WF_BLOCK + threadIdx.x;
#else
                const uint32_t v = w * win + kb + threadIdx.x;
#endif
                mykey[k] = 0xffffffffu; myq[k] = 0; myrank[k] = 0;
                if (kb < win && v < total) {
                    myq[k] = g0 ? v : wf_map_index(pref, B.n_seg, B.seg_cap, v);
                    const uint32_t he = B.hit[in][myq[k]].ent;
                    if (he == EU_WF_ENT_DEAD) B.node_kind[node_base + myq[k]] = (uint8_t)TS_NONE;      /* (generation 0 only) nothing to shade, nothing to resolve */
                    else {
                        /* (round 4 measured the keys in the order "most expensive surface first", so that what a wave waits for at the window's
                         * last barrier is a chunk of wall rays, not of glass rays: 1.31-1.35 against 1.30 ms on 3d_room, the others within noise) */
                        mykey[k] = he < EU_WF_KEYS - 1 ? he : EU_WF_KEYS - 1;
                        myrank[k] = atomicAdd(&hist[mykey[k]], 1u);
                    }
                }
            }
            __syncthreads();
            /* (round 4 measured the exclusive scan of the key counts done by half a wave instead of this loop, and a thread's later queue slots found by a
             * short walk from its first instead of the ten-step search: 1.41 against 1.27 ms on 3d_room, 8.06 against 8.37 Gray/s with frames in flight --
             * six more VGPRs in a kernel that sits at 128; removed) */
            if (threadIdx.x == 0) { uint32_t run = 0; for (uint32_t k = 0; k < EU_WF_KEYS; k++) { offs[k] = run; run += hist[k]; } n_sorted = run; next_chunk = 0; }
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < EU_WF_WIN / EU_WF_BLOCK; k++) if (mykey[k] != 0xffffffffu) sorted[offs[mykey[k]] + myrank[k]] = myq[k];
            __syncthreads();
            const uint32_t n_live = n_sorted;
            /* The sorted window is shaded in chunks of one wave's 64 rays, and the waves of the workgroup TAKE chunks (a counter in LDS)
             * instead of owning every fourth one: a chunk of glass rays costs five times a chunk of wall rays, the sort puts them
             * side by side, and the other three waves would wait for the unlucky one at the window's last barrier.
             * (Round 4 also measured a software pipeline here -- the next sub-batch's hit / routing / ray records requested before the
             * current one is shaded: 19 more VGPRs, no gain: 1.694 vs 1.685 ms on 3d_room; what the kernel waits for is not these loads.) */
            const uint32_t n_chunks = (n_live + 63u) >> 6;
#if !EU_SHADE_TAKE_CHUNKS
            uint32_t own_chunk = threadIdx.x >> 6;
#endif
            for (;;) {
#if EU_SHADE_TAKE_CHUNKS
            uint32_t chunk = 0;
            if ((threadIdx.x & 63u) == 0) chunk = atomicAdd(&next_chunk, 1u);
            chunk = (uint32_t)__builtin_amdgcn_readfirstlane((int)chunk);
#else       /* A/B: wave j owns chunks j, j + 4, ... */
            const uint32_t chunk = own_chunk;
            own_chunk += EU_WF_BLOCK / 64;
#endif
            if (chunk >= n_chunks) break;
            const uint32_t sidx = chunk * 64u + (threadIdx.x & 63u);
            const bool live = sidx < n_live;
            const uint32_t i = live ? sorted[sidx] : 0u;
            WF_STAMP(0);
            const uint32_t nid = node_base + i;
            /* children of this ray: 0 = transmission, 1 = reflection; c_sm = slot | delivery mode << 1 (trace_nodes.h) */
            uint32_t n_child = 0;
            uint32_t fin_kind = TS_NONE, fin_parent = 0, fin_sm = 0;      /* last generation: the node this lane finishes itself */
            real fin_ratio = R(0.0);
            uint32_t fin_spx = 0;
            Rgba fin_inter = {R(0.0), R(0.0), R(0.0), R(0.0)};
            real c_o[2][D], c_d[2][D];
            uint32_t c_ent[2] = {0, 0}, c_parent[2] = {0, 0}, c_sm[2] = {0, 0};
            bool bg_miss = false;
            if (live) {
                WF_STAMP(1);
                uint32_t parent, ent, sm;
                real o[D], d[D];
                const EuWfHit hit = B.hit[in][i];
                uint32_t hit_ent = hit.ent;
                if (g0) {
                    const EuPrimary pr = wf_primary_ray<D, P>(S, cam, fr, cam_ent, i, o, d);
                    parent = pr.out_idx; ent = (uint32_t)cam_ent; sm = (uint32_t)(TS_MODE_ROOT << 1);      /* the colour goes straight to the pixel */
                    if (pr.status != EU_PR_RAY) {      /* (hit_ent == EU_WF_ENT_SPECIAL) a pixel without a ray */
                        hit_ent = EU_WF_ENT_SPECIAL;
                        if (hit_t_aov) hit_t_aov[pr.out_idx] = -R(1.0);
                        if (pr.status == EU_PR_PAD) rgba[pr.out_idx] = 0u;
                        else if (pr.status == EU_PR_CROSS) {
                            rgba[pr.out_idx] = 0xff0000ffu;
                            if (point_rgb) { point_rgb[0] = R(1.0); point_rgb[1] = R(0.0); point_rgb[2] = R(0.0); }
                        } else {      /* trace_screen_point's checkerboard (universe/mod.rs:387-395) */
                            const bool black = (((int)pr.px_x / 8 + (int)pr.px_y / 8) % 2) == 0;
                            rgba[pr.out_idx] = black ? 0xff000000u : 0xffff00ffu;
                            if (point_rgb) { point_rgb[0] = black ? R(0.0) : R(1.0); point_rgb[1] = R(0.0); point_rgb[2] = black ? R(0.0) : R(1.0); }
                        }
                    }
                } else {
                    const uint2 pa = B.ray_pa[in][i];
                    parent = pa.x; ent = pa.y & 0xffffu; sm = (pa.y >> 16) & 7u;
                    wf_load_ray<D>(B, in, i, o, d);
                }
                uint32_t node_kind = TS_NONE;
                if (hit_ent == EU_WF_ENT_SPECIAL) {
                } else if (hit_ent == EU_WF_ENT_MISS) {
                    /* nothing hit: the background colour goes to the parent; handled as a depth-0 "child" below */
#pragma unroll
                    for (int k = 0; k < D; k++) { c_o[0][k] = o[k]; c_d[0][k] = d[k]; }
                    c_parent[0] = parent; c_sm[0] = sm;
                    n_child = 1; bg_miss = true;
                } else {
                    const real best_t = hit.t;
                    const uint32_t best_code = hit.code;
                    HitCtx<D> c;
                    c.finish(best_t, o, d);
                    P::hit_normal(S, hit_ent, best_code, o, d, c.loc, c.normal);
                    c.classify();
                    WF_STAMP(2);
                    SurfaceEval<D> E;
                    P::surface(S, hit_ent, c, time_s, cnt, color_stack + threadIdx.x, EU_WF_BLOCK, E);
                    const real ratio = E.ratio;
                    WF_STAMP(3);
                    bool have_inter = false, need_trans = false;
                    Rgba inter = {R(0.0), R(0.0), R(0.0), R(0.0)};
                    const uint32_t spx = E.spx;
                    int dest = -1;
                    if (E.have_color) {                                                /* get_intersection_color, surface.rs:62-117 */
                        if (!E.translucent) { inter = E.sc; have_inter = true; }
                        else {
#pragma unroll
                            for (int k = 0; k < D; k++) { c_d[0][k] = E.thr[k]; c_o[0][k] = c.loc[k] + -c.nc[k] * EU_EPS * R(128.0); }
                            dest = c.exiting ? P::material_at(S, c_o[0]) : (int)hit_ent;
                            WF_SUB(11);
                            if (dest >= 0) {
                                P::material_apply(S, ent, c_d[0], true);
                                P::material_apply(S, (uint32_t)dest, c_d[0], false);
                                need_trans = true;
                            }
                        }
                    }
                    WF_STAMP(4);
                    const bool need_refl = !(ratio <= R(0.0));                                /* get_reflection_color, surface.rs:119-139 */
                    const uint32_t rs = need_trans ? 1u : 0u;        /* reflection goes to child slot rs in the arrays */
                    if (need_refl) {
                        const real dn = vdot<D>(c.dir, c.nc);
#pragma unroll
                        for (int k = 0; k < D; k++) {
                            const real rd = c.nc[k] * -R(2.0) * dn + c.dir[k];              /* surface.rs:246-256 */
                            const real ro = c.loc[k] + c.nc[k] * EU_EPS * R(128.0);
                            if (rs) { c_d[1][k] = rd; c_o[1][k] = ro; } else { c_d[0][k] = rd; c_o[0][k] = ro; }
                        }
                    }
                    fin_ratio = ratio; fin_spx = spx; fin_inter = inter;
                    EuTsNode *N = B.nodes + nid;
                    /* (the last generation -- no depth left -- queues nothing: its children are background samples this very thread takes,
                     * so it finishes the node in registers, below, and no record is written) */
                    if (need_trans) {      /* the transmitted colour arrives quantised (slot 0), the reflection as it is (slot 1) */
                        node_kind = need_refl ? TS_COMBINE_TRANS : TS_OVER;
                        if (!last_gen) {
                        N->spx = spx; N->parent = parent; N->meta = node_kind | (sm << 8);
                        if (need_refl) N->ratio = ratio;
                        }
                        c_ent[0] = (uint32_t)dest; c_parent[0] = nid; c_sm[0] = 0u | (TS_MODE_U8 << 1);
                        n_child = 1;
                        if (need_refl) { c_ent[1] = ent; c_parent[1] = nid; c_sm[1] = 1u | (TS_MODE_F64 << 1); n_child = 2; }
                    } else if (need_refl) {
                        c_ent[0] = ent;
                        n_child = 1;
                        if (have_inter) {      /* the opaque surface colour waits in the node; the reflection is combined with it on arrival */
                            node_kind = TS_COMBINE_INTER;
                            if (!last_gen) {
                            N->ratio = ratio; N->parent = parent; N->meta = node_kind | (sm << 8);
                            N->c1[0] = inter.r; N->c1[1] = inter.g; N->c1[2] = inter.b; N->c1[3] = inter.a;
                            }
                            c_parent[0] = nid; c_sm[0] = 1u | (TS_MODE_INTER << 1);
                        } else {   /* the reflection colour is the result (surface.rs:153-154): the child reports to our parent */
                            c_parent[0] = parent; c_sm[0] = sm;
                        }
                    } else {
                        if (!have_inter) cnt.errors++;            /* the reference panics here (surface.rs:154) */
                        ts_deliver(B.nodes, parent, sm, inter, cnt, rgba, point_rgb);
                    }
                }
                B.node_kind[nid] = (uint8_t)(node_kind == TS_COMBINE_TRANS && !last_gen ? TS_COMBINE_TRANS : TS_NONE);      /* (only a node with two children is left to the resolve pass: trace_nodes.h) */
                fin_kind = node_kind; fin_parent = parent; fin_sm = sm;
            }
            /* children with no depth left (or plain misses) only sample the background
             * (universe/mod.rs:157,183): one code site for all of them */
            WF_STAMP(5);
            /* (a lane's children are either all background-only or all queued: the miss case has one child) */
            const bool bg_only = n_child != 0 && (bg_miss || child_depth == 0);
            const uint32_t n_queue = bg_only ? 0u : n_child;
            Rgba bgc[2] = {{R(0.0), R(0.0), R(0.0), R(0.0)}, {R(0.0), R(0.0), R(0.0), R(0.0)}};
#pragma unroll 1
            for (uint32_t k = 0; k < 2; k++) {      /* constant indices only: a run-time indexed private array would live in scratch */
                if (bg_only && k < n_child) {
                    real dd[D];
#pragma unroll
                    for (int q = 0; q < D; q++) dd[q] = k ? c_d[1][q] : c_d[0][q];
                    const Rgba bg = ts_background<D, P>(S, dd, cnt);
                    if (k) bgc[1] = bg; else bgc[0] = bg;
                }
            }
            if (bg_only) {
                if (bg_miss || fin_kind == TS_NONE) {      /* a ray that hit nothing, or a reflection that IS the result (surface.rs:153-154): the colour goes where the ray's would */
                    ts_deliver(B.nodes, c_parent[0], c_sm[0], bgc[0], cnt, rgba, point_rgb);
                } else {
                    /* the node's children are both here: what ts_deliver and the resolve pass would do with its record, done in registers
                     * (surface.rs:104-114: surface_palette.over(transition_palette), both quantised; :159-161: combine_palette_color) */
                    Rgba res;
                    if (fin_kind == TS_COMBINE_INTER) res = combine_palette_color(bgc[0], fin_inter, fin_ratio);
                    else {
                        const Rgba over = blend_rgba(EU_BL_OVER, new_u8(fin_spx), new_u8(to_pixel4(bgc[0], cnt)));
                        res = fin_kind == TS_OVER ? over : combine_palette_color(bgc[1], over, fin_ratio);
                    }
                    ts_deliver(B.nodes, fin_parent, fin_sm, res, cnt, rgba, point_rgb);
                }
            }
            WF_STAMP(6);
            uint32_t pos1;
            const uint32_t pos0 = wf_append_local(&seg_fill, n_queue, pos1);
            if (n_queue >= 1) {
                if (pos0 >= B.seg_cap) cnt.errors++, atomicAdd(&counters->overflow, 1ull);
                else wf_store_ray<D>(B, outb, out_base + pos0, c_o[0], c_d[0], c_parent[0], c_ent[0] | (c_sm[0] << 16));
            }
            if (n_queue >= 2) {
                if (pos1 >= B.seg_cap) cnt.errors++, atomicAdd(&counters->overflow, 1ull);
                else wf_store_ray<D>(B, outb, out_base + pos1, c_o[1], c_d[1], c_parent[1], c_ent[1] | (c_sm[1] << 16));
            }
            WF_STAMP(7);
            }   /* chunks */
#if EU_WF_DEAL_SHADE
            if (threadIdx.x == 0) next_win = dealt ? gridDim.x + ticket : n_win;
            __syncthreads();     /* `sorted` is rewritten by the next window */
            w = (uint32_t)__builtin_amdgcn_readfirstlane((int)next_win);
#else
            __syncthreads();     /* `sorted` is rewritten by the next window */
            w += gridDim.x;
#endif
        }
    }
    if (threadIdx.x == 0) next_chunk = 0;      /* (every wave is past its last window's chunks: the counter deals the tail's batches next) */
    __syncthreads();
    WF_SUB(13);
    if constexpr (FUSE >= 0) {
        /* trace_closest of this workgroup's own children (the barrier above made their records visible to all its waves; the same
         * compute unit wrote them, so its L1 holds nothing stale). */
        const uint32_t n_own = seg_fill < B.seg_cap ? seg_fill : B.seg_cap;
        if (n_own != 0u) {
            EuScene S2;
            S2.init(scene_g);      /* (the intersect part reads the scene through scalar loads from device memory, whatever the shading part used) */
            typename eu_conditional<FUSE == 0, HitStackLds, HitStackPriv<(FUSE > 0 ? FUSE : 1)>>::type HS;
            if constexpr (FUSE == 0) {
                const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
                real *hs_t = (real *)(lds_dyn);
                uint32_t *hs_c = (uint32_t *)(hs_t + (EU_WF_BLOCK / 64) * hs_cap * 64);
                HS.t = hs_t + wave * hs_cap * 64 + lane;
                HS.c = hs_c + wave * hs_cap * 64 + lane;
                HS.cap = hs_cap;
            }
            for (;;) {      /* batches of 64 in queue order, taken by whichever wave is free (a batch of rays inside the glass costs several times a batch that misses everything) */
                uint32_t batch = 0;
                if ((threadIdx.x & 63u) == 0) batch = atomicAdd(&next_chunk, 1u);
                batch = (uint32_t)__builtin_amdgcn_readfirstlane((int)batch);
                if (batch * 64u >= n_own) break;
                const uint32_t idx = batch * 64u + (threadIdx.x & 63u);
                if (idx < n_own) {
                    const uint32_t q = out_base + idx;
                    real o[D], d[D];
                    wf_load_ray<D>(B, outb, q, o, d);
                    bool have = false, fail = false;
                    real best_t = R(0.0);
                    uint32_t best_code = 0, best_ent = EU_WF_ENT_MISS;
                    cnt.rays++;
#ifdef EU_PROFILE_SHADE_WAVE      /* (the shape routines' own stamps belong to the EU_PROFILE_SHAPE build) */
                    unsigned long long *const prof_rows_keep = cnt.prof;
                    cnt.prof = nullptr;
#endif
                    P::trace_closest(S2, o, d, HS, cnt, 2, fail, have, best_t, best_code, best_ent);
#ifdef EU_PROFILE_SHADE_WAVE
                    cnt.prof = prof_rows_keep;
#endif
                    EuWfHit h;
                    h.t = best_t; h.code = best_code; h.ent = best_ent;
#if EU_REAL_BITS == 32
                    h.pad = 0;
#endif
                    B.hit[outb][q] = h;
                    WF_SUB(12);
                }
            }
            WF_SUB(14);
        }
    }
    if (threadIdx.x == 0) {
        const uint32_t queued = seg_fill < B.seg_cap ? seg_fill : B.seg_cap;
        B.seg_count[(gen + 1) * B.n_seg + blockIdx.x] = queued;
        if (queued) atomicAdd(B.work + (gen + 1) * EU_WORK_PER_GEN + EU_WORK_TOTAL, queued);
    }
    wf_flush_counters(counters, cnt);
    WF_PROF_END(B, 1, gen);
}

template <int D, bool SCENE_LDS>
__global__ __launch_bounds__(EU_WF_BLOCK, EU_SHADE_WAVES) void eu_wf_shade_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t gen, uint32_t max_depth, real time_s,
                                                                  EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {
    extern __shared__ uint64_t lds_dyn[];
    const EuDevCamera cam = {};
    const EuDevFrame fr = {};
    wf_shade_body<D, SCENE_LDS, EuInterp<D>, false>(scene_g, scene_words, gen, max_depth, time_s, cam, fr, B, counters, rgba, nullptr, point_rgb, lds_dyn);
}
template <int D, bool SCENE_LDS>
__global__ __launch_bounds__(EU_WF_BLOCK, EU_SHADE_WAVES) void eu_wf_shade0_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, EuDevCamera cam, EuDevFrame fr,
                                                                   EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ hit_t_aov, eu_f64 *__restrict__ point_rgb) {
    extern __shared__ uint64_t lds_dyn[];
    wf_shade_body<D, SCENE_LDS, EuInterp<D>, true>(scene_g, scene_words, 0u, cam.max_depth, fr.time_s, cam, fr, B, counters, rgba, hit_t_aov, point_rgb, lds_dyn);
}

/* the fused forms (interpreter: hit stack in LDS only -- scenes whose stack is deeper keep the two-kernel pipeline) */
template <int D, bool SCENE_LDS>
__global__ __launch_bounds__(EU_WF_BLOCK, EU_SHADE_WAVES) void eu_wf_fshade_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t hs_cap, uint32_t gen, uint32_t max_depth, real time_s,
                                                                   EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {
    extern __shared__ uint64_t lds_dyn[];
    const EuDevCamera cam = {};
    const EuDevFrame fr = {};
    wf_shade_body<D, SCENE_LDS, EuInterp<D>, false, 0>(scene_g, scene_words, gen, max_depth, time_s, cam, fr, B, counters, rgba, nullptr, point_rgb, lds_dyn, hs_cap);
}
template <int D, bool SCENE_LDS>
__global__ __launch_bounds__(EU_WF_BLOCK, EU_SHADE_WAVES) void eu_wf_fshade0_kernel(const uint64_t *__restrict__ scene_g, uint32_t scene_words, uint32_t hs_cap, EuDevCamera cam, EuDevFrame fr,
                                                                    EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ hit_t_aov, eu_f64 *__restrict__ point_rgb) {
    extern __shared__ uint64_t lds_dyn[];
    wf_shade_body<D, SCENE_LDS, EuInterp<D>, true, 0>(scene_g, scene_words, 0u, cam.max_depth, fr.time_s, cam, fr, B, counters, rgba, hit_t_aov, point_rgb, lds_dyn, hs_cap);
}

/* ------------------------------------------------------------------ bottom-up resolve of one generation's nodes
 * (surface_palette.over(transition_palette), both quantised to u8: surface.rs:104-114; combine: surface.rs:159-161);
 * a node of generation 0 delivers to its pixel (trace_nodes.h): there is no separate final pass.
 * (Round 3 measured two other forms.  All generations in ONE launch with a grid-wide barrier between them -- release / acquire fences at
 * device scope around a counter: 4.8 instead of 6.2 Gray/s on 3d_room; each fence writes back and invalidates the XCD's L2, which at that
 * moment holds the other band pipeline's ray queues.  One workgroup per queue segment, without the prefix table: 26 us per launch instead
 * of 19; the launch is bound by its node traffic -- 64-byte records, 32-byte deliveries -- and wants the whole chip's worth of waves.) */
__global__ __launch_bounds__(EU_WF_BLOCK) void eu_wf_resolve_kernel(uint32_t gen, uint32_t total0, EuWfBuffers B, EuDevCounters *counters, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {
    WF_PROF_BEGIN();
    if (gen != 0 && B.work[gen * EU_WORK_PER_GEN + EU_WORK_TOTAL] == 0u) { WF_PROF_END(B, 2, gen); return; }
    LaneCounters cnt = {0, 0, 0, 0};
    const uint32_t node_base = gen * B.ray_cap;
    const bool g0 = gen == 0;
    __shared__ uint32_t pref[EU_WF_MAX_SEG + 1];
    __shared__ uint32_t wave_tot[4];
    const uint32_t total = g0 ? total0 : wf_build_prefix(B.seg_count + gen * B.n_seg, B.n_seg, pref, wave_tot);
    for (uint32_t v = blockIdx.x * blockDim.x + threadIdx.x; v < total; v += gridDim.x * blockDim.x) {
        const uint32_t nid = node_base + (g0 ? v : wf_map_index(pref, B.n_seg, B.seg_cap, v));
        if (B.node_kind[nid] != (uint8_t)TS_COMBINE_TRANS) continue;
        ts_finish_two(B.nodes, B.nodes + nid, cnt, rgba, point_rgb);
    }
    wf_flush_counters(counters, cnt);
    WF_PROF_END(B, 2, gen);
}

#endif
