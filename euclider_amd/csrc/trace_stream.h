/*
 * trace_stream.h -- the trace loop as ONE persistent kernel of self-contained workgroups (gfx950).
 *
 * The reference recurses per pixel: trace -> trace_closest -> get_color -> trace ...
 * (/root/reference/src/universe/mod.rs:149-184, universe/entity/surface.rs:62-162).  The round-1 pipeline processed
 * the recursion tree of the whole frame one generation (= recursion depth) per kernel launch: 3 launches per
 * generation, each ending in a tail during which most CUs idled, and every generation's rays, hits and colours
 * travelling through HBM.  Here a frame is one launch, and there is no synchronisation between workgroups at all:
 *
 *   - every workgroup owns EU_TS_NCH ray CHUNKS (EU_TS_CH rays each, component-major) in HBM that only it touches:
 *     they stay hot in its XCD's L2 and are recycled through a LIFO free list, so ray traffic rarely reaches HBM;
 *   - a workgroup keeps one OPEN chunk per generation that children are appended to, and a stack of FULL chunks.
 *     Its loop: take the deepest full chunk and process it (trace_closest for all its rays -> counting sort by the
 *     entity hit -> get_color up to the recursive calls, children appended to the next generation's open chunk);
 *     when no chunk is full, pull the next 256-pixel tile from a global counter and append its camera rays to
 *     generation 0; when the pixels are gone, flush the open chunks, shallowest first.  A chunk is therefore
 *     full whatever the generation's size is frame-wide (3d_hallways: 9 of 12 generations hold < 2 % of the rays),
 *     and the only partially filled batches are the <= max_depth flushed ones per workgroup at the very end;
 *   - deepest-first bounds the chunks in use: <= 2 full ones per generation + 1 open + 3 spares < EU_TS_NCH.  A queue
 *     cannot overflow, whatever the fan-out of the recursion;
 *   - a ray whose colour needs its children's colours (over / combine, surface.rs:104-114,159-161) leaves a 64-byte
 *     tree NODE in a pool shared by all workgroups (chunks of EU_TS_NCN nodes handed out by one atomic each, linked
 *     per workgroup and generation).  A finished colour is DELIVERED to its parent in the form the parent will use
 *     it: the transmitted colour already quantised to u8 (4 bytes, surface.rs:104-112 does that first thing), the
 *     reflection next to an opaque surface colour already combined with it, a primary ray's colour straight to the
 *     RGBA8 pixel (fg.over(white), to_pixel: universe/mod.rs:263-269,342);
 *   - when its rays are done, the workgroup resolves its own nodes, deepest generation first (coalesced sweeps
 *     over its node chunks, block barriers only).  A node's children were appended, traced and shaded by the
 *     workgroup that created the node, so nothing here ever waits for another workgroup.
 *
 * The order in which rays are processed does not matter: every step is a pure function of the ray, so the
 * result is bit-identical to the depth-first recursion.
 */
#ifndef EU_TRACE_STREAM_H
#define EU_TRACE_STREAM_H

#include <type_traits>

#include "trace_device.h"
#include "trace_nodes.h"

#define EU_TS_BLOCK 256
#ifndef EU_TS_CH
#define EU_TS_CH 512            /* rays per chunk = rays intersected, sorted and shaded together */
#endif
#define EU_TS_SUB (EU_TS_CH / EU_TS_BLOCK)
#define EU_TS_NCH 64            /* ray chunks owned by one workgroup */
#define EU_TS_NCN EU_TS_CH      /* node slots per node chunk (a shade phase creates at most EU_TS_CH nodes) */
#define EU_TS_NONE 0xffffffffu
#define EU_TS_KEYS 32
#define EU_TS_ROW 16            /* counter words per workgroup row: rays, bg, nan, errors, abort, then (diagnostic build -DEU_TS_PROFILE) clock shares */
#ifdef EU_TS_PROFILE
#define TS_CLK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); prof[(k)] += now_ - last_; last_ = now_; } while (0)
#else
#define TS_CLK(k) do { } while (0)
#endif
static_assert(EU_TS_CH <= 2048 && EU_TS_CH % EU_TS_BLOCK == 0, "rank field: 11 bits");
#ifndef EU_TS_WAVES
#define EU_TS_WAVES 3           /* waves per SIMD the kernel is compiled for (168 VGPRs) */
#endif

enum { TS_WORK_DONE = 0, TS_WORK_GENERATE = 1, TS_WORK_PROCESS = 2, TS_WORK_ABORT = 3 };

struct EuTsPool {
    double *ray_od;             /* [n_wg * NCH][2 D][CH] origin then direction, component-major inside a chunk */
    uint32_t *ray_parent;       /* [n_wg * NCH][CH] */
    uint32_t *ray_aux;          /* entity the ray travels in (bits 0..15) | slot/mode of the delivery (bits 16..18) */
    double *hit_t;              /* [n_wg][CH] hits of the chunk being processed */
    uint32_t *hit_code;
    EuTsNode *nodes;            /* [n_node_chunks * NCN] */
    uint32_t *nchunk_prev;      /* [n_node_chunks] previous chunk of the same workgroup and generation */
    unsigned long long *wg_counters;   /* [n_wg][EU_TS_ROW] */
    uint32_t n_node_chunks, n_wg;
    uint32_t nch, node_chunk_base;   /* node chunk ids handed out start here (import mode: the pool lies behind the wavefront pipeline's slots in one array) */          /* ray chunks per workgroup (<= EU_TS_NCH): 3 per generation of the frame's max_depth + 4 */
};

struct TsState {                /* per workgroup, LDS */
    uint32_t open_chunk[EU_MAX_DEPTH], open_fill[EU_MAX_DEPTH];
    uint32_t node_head[EU_MAX_DEPTH], node_fill[EU_MAX_DEPTH];
    uint32_t node_spare;
    uint32_t ready[EU_TS_NCH], n_ready;       /* full chunks: local chunk | generation << 8; the top is the deepest */
    uint32_t free_list[EU_TS_NCH], n_free;
    uint32_t work, cur_chunk, cur_gen, cur_count, cur_tile, next_tile;
    uint32_t imp_off, imp_count;               /* import mode: progress inside the queue segment cur_tile */
    uint32_t app_pos, app_chunk[3];           /* ray append stream of the next generation */
    uint32_t napp_pos, napp_chunk[2];         /* node append stream of this generation */
    uint32_t hist[EU_TS_KEYS], offs[EU_TS_KEYS];
    uint32_t sorted[EU_TS_CH];                /* index in the chunk | entity hit << 16 (0xffff: none) */
    uint32_t krk[EU_TS_CH];                   /* per ray of the chunk: sort key (5 bits) | rank inside the key << 5 | entity hit << 16 */
    unsigned long long wg_cnt[4];
};

/* wave-aggregated reservation of n (0..2) slots per lane on an LDS counter: first slots of all lanes, then second slots */
EU_DEV uint32_t ts_reserve(uint32_t *lds_counter, uint32_t n, uint32_t &second) {
    const unsigned long long m1 = __ballot(n >= 1), m2 = __ballot(n >= 2);
    const uint32_t rank1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
    const uint32_t rank2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
    const uint32_t total = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
    uint32_t base = 0;
    if ((threadIdx.x & 63) == 0 && total) base = atomicAdd(lds_counter, total);
    base = __builtin_amdgcn_readfirstlane(base);
    second = base + (uint32_t)__popcll(m1) + rank2;
    return base + rank1;
}

/* ---- thread 0's bookkeeping (between block barriers) */
EU_DEV uint32_t ts_free_pop(TsState &st) {
    if (st.n_free == 0) { st.work = TS_WORK_ABORT; return 0; }      /* cannot happen (see the bound above); never corrupt memory */
    return st.free_list[--st.n_free];
}
EU_DEV void ts_open_append(TsState &st, uint32_t gen) {
    st.app_pos = st.open_fill[gen];
    st.app_chunk[0] = st.open_chunk[gen] != EU_TS_NONE ? st.open_chunk[gen] : ts_free_pop(st);
    st.app_chunk[1] = ts_free_pop(st);
    st.app_chunk[2] = ts_free_pop(st);
}
EU_DEV void ts_close_append(TsState &st, uint32_t gen) {
    const uint32_t total = st.app_pos, full = total / EU_TS_CH, rem = total % EU_TS_CH;
    for (uint32_t k = 0; k < full; k++) {
        if (st.n_ready >= EU_TS_NCH) { st.work = TS_WORK_ABORT; return; }
        st.ready[st.n_ready++] = st.app_chunk[k] | (gen << 8);
    }
    uint32_t used = full;
    if (rem) { st.open_chunk[gen] = st.app_chunk[full]; st.open_fill[gen] = rem; used = full + 1; }
    else { st.open_chunk[gen] = EU_TS_NONE; st.open_fill[gen] = 0; }
    for (uint32_t k = used; k < 3; k++) st.free_list[st.n_free++] = st.app_chunk[k];
}
EU_DEV uint32_t ts_node_chunk_alloc(TsState &st, EuDevCounters *counters, const EuTsPool &P) {
    const uint32_t id = (uint32_t)atomicAdd(&counters->node_chunks, 1ull);
    if (id >= P.n_node_chunks) { st.work = TS_WORK_ABORT; return 0; }
    return P.node_chunk_base + id;
}
EU_DEV void ts_open_nodes(TsState &st, uint32_t gen, EuDevCounters *counters, const EuTsPool &P) {
    if (st.node_head[gen] == EU_TS_NONE) {
        const uint32_t id = ts_node_chunk_alloc(st, counters, P);
        if (st.work == TS_WORK_ABORT) return;
        P.nchunk_prev[id] = EU_TS_NONE;
        st.node_head[gen] = id; st.node_fill[gen] = 0;
    }
    if (st.node_spare == EU_TS_NONE) {
        st.node_spare = ts_node_chunk_alloc(st, counters, P);
        if (st.work == TS_WORK_ABORT) { st.node_spare = EU_TS_NONE; return; }
    }
    st.napp_pos = st.node_fill[gen];
    st.napp_chunk[0] = st.node_head[gen];
    st.napp_chunk[1] = st.node_spare;
}
EU_DEV void ts_close_nodes(TsState &st, uint32_t gen, const EuTsPool &P) {
    const uint32_t total = st.napp_pos;
    if (total > EU_TS_NCN) {        /* the head chunk is full, the spare becomes the head */
        P.nchunk_prev[st.node_spare] = st.node_head[gen];
        st.node_head[gen] = st.node_spare;
        st.node_fill[gen] = total - EU_TS_NCN;
        st.node_spare = EU_TS_NONE;
    } else st.node_fill[gen] = total;
}

/* thread 0: put the finished step's output away, then decide what the workgroup does next */
/* Where a workgroup's rays come from.  Frame mode: 256-pixel tiles of the frame (camera rays, generation 0).  Import mode (the
 * wavefront pipeline's FINISH step): the segments of that pipeline's queue of generation `gen0`, taken over 256 rays at a time;
 * the workgroup then finishes those rays and all their descendants, and delivers into the pipeline's nodes and pixels. */
struct TsSource {
    uint32_t n_tiles;           /* work tiles of the frame | queue segments */
    uint32_t gen0;              /* generation the incoming rays belong to (0 | the hand-over generation) */
    const uint32_t *seg_count;  /* import mode: rays in each segment of the queue; nullptr in frame mode */
};

EU_DEV void ts_schedule(TsState &st, uint32_t max_depth, const TsSource &src, EuDevCounters *counters, const EuTsPool &P) {
    if (st.work == TS_WORK_PROCESS) {
        if (st.cur_gen + 1 < max_depth) ts_close_append(st, st.cur_gen + 1);
        ts_close_nodes(st, st.cur_gen, P);
        st.free_list[st.n_free++] = st.cur_chunk;
    } else if (st.work == TS_WORK_GENERATE && max_depth > src.gen0) ts_close_append(st, src.gen0);
    if (st.work == TS_WORK_ABORT) return;
    if (st.n_ready) {                               /* the deepest full chunk */
        const uint32_t e = st.ready[--st.n_ready];
        st.work = TS_WORK_PROCESS; st.cur_chunk = e & 0xffu; st.cur_gen = e >> 8; st.cur_count = EU_TS_CH;
    } else if (!src.seg_count && st.next_tile < src.n_tiles) {       /* more pixels */
        st.work = TS_WORK_GENERATE; st.cur_tile = st.next_tile;
    } else {
        st.work = TS_WORK_DONE;
        if (src.seg_count) {                        /* more queued rays to take over?  (skips empty segments) */
            while (st.imp_off >= st.imp_count && st.next_tile < src.n_tiles) {
                st.cur_tile = st.next_tile; st.imp_off = 0; st.imp_count = src.seg_count[st.cur_tile];
                st.next_tile = (uint32_t)atomicAdd(&counters->next_item, 1ull);
            }
            if (st.imp_off < st.imp_count) st.work = TS_WORK_GENERATE;
        }
        if (st.work == TS_WORK_DONE) {              /* flush: the shallowest open chunk (its children top up the deeper ones) */
            for (uint32_t g = 0; g < max_depth; g++) {
                if (st.open_fill[g]) {
                    st.work = TS_WORK_PROCESS; st.cur_chunk = st.open_chunk[g]; st.cur_gen = g; st.cur_count = st.open_fill[g];
                    st.open_chunk[g] = EU_TS_NONE; st.open_fill[g] = 0;
                    break;
                }
            }
        }
    }
    if (st.work == TS_WORK_PROCESS) {
        if (st.cur_gen + 1 < max_depth) ts_open_append(st, st.cur_gen + 1);
        ts_open_nodes(st, st.cur_gen, counters, P);
        for (uint32_t k = 0; k < EU_TS_KEYS; k++) st.hist[k] = 0;
    } else if (st.work == TS_WORK_GENERATE && max_depth > src.gen0) ts_open_append(st, src.gen0);
}

/* Everything the kernel is told, as ONE by-value argument.  The kernel never touches the argument itself: every phase reads
 * the fields it needs from the kernarg segment through a pointer the compiler must treat as new each time (ts_params), with
 * scalar loads.  A persistent loop otherwise invites the compiler to hoist every loop-invariant load (68 camera words, the
 * scene header twice, ten pool pointers) in front of the loop and keep it all live: the first build of this kernel spilled
 * 484 SGPRs and 69 VGPRs that way. */
struct EuTsParams {
    const uint64_t *scene_g;
    uint32_t scene_words, hs_cap;
    EuDevCamera cam;
    EuDevFrame fr;
    EuTsPool P;
    EuDevCounters *counters;
    uint32_t *rgba;
    eu_f64 *hit_t_aov, *point_rgb;
    /* import mode (import_gen != 0xffffffff): the wavefront pipeline's queue of that generation */
    uint32_t import_gen, imp_n_seg, imp_seg_cap, imp_ray_cap;
    const double *imp_ray;            /* the wavefront queue's records (EuWfRay<D>, trace_wavefront.h) */
    const uint2 *imp_ray_pa;
    const uint32_t *imp_seg_count;
    uint32_t *imp_seg_count_rows;      /* the pipeline's [EU_MAX_DEPTH + 1][n_seg] table: rows behind import_gen are cleared (they describe an older frame) */
    EuDevCounters *stats_counters;     /* not null: add rays / background samples / would-panic counts (and an abort) to the pipeline's accounting */
};
typedef const EuTsParams __attribute__((address_space(4))) *TsParamsPtr;
EU_DEV TsParamsPtr ts_params() {
    auto p = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return (TsParamsPtr)p;
}
EU_DEV EuTsPool ts_pool(TsParamsPtr q) {
    EuTsPool P;
    P.ray_od = q->P.ray_od; P.ray_parent = q->P.ray_parent; P.ray_aux = q->P.ray_aux; P.hit_t = q->P.hit_t; P.hit_code = q->P.hit_code;
    P.nodes = q->P.nodes; P.nchunk_prev = q->P.nchunk_prev; P.wg_counters = q->P.wg_counters;
    P.n_node_chunks = q->P.n_node_chunks; P.n_wg = q->P.n_wg; P.nch = q->P.nch; P.node_chunk_base = q->P.node_chunk_base;
    return P;
}
EU_DEV TsSource ts_source(TsParamsPtr q) {
    TsSource src;
    const uint32_t ig = q->import_gen;
    if (ig == 0xffffffffu) { src.n_tiles = (q->fr.n_tiles + 3u) / 4u; src.gen0 = 0; src.seg_count = nullptr; }
    else { src.n_tiles = q->imp_n_seg; src.gen0 = ig; src.seg_count = q->imp_seg_count; }
    return src;
}

template <int D> EU_DEV void ts_store_ray(const EuTsPool &P, const TsState &st, uint32_t pos, const double *o, const double *d, uint32_t parent, uint32_t aux) {
    const uint32_t gch = blockIdx.x * P.nch + st.app_chunk[pos / EU_TS_CH], off = pos % EU_TS_CH;
    double *od = P.ray_od + (size_t)gch * (2 * D * EU_TS_CH) + off;
#pragma unroll
    for (int k = 0; k < D; k++) { od[k * EU_TS_CH] = o[k]; od[(D + k) * EU_TS_CH] = d[k]; }
    P.ray_parent[(size_t)gch * EU_TS_CH + off] = parent;
    P.ray_aux[(size_t)gch * EU_TS_CH + off] = aux;
}

/* ---------------------------------------------------------------- camera rays of one work tile (camera.rs:164-185, mod.rs:253-271) */
template <int D> EU_DEV void ts_generate(TsState &st, LaneCounters &cnt, int cam_ent) {
    TsParamsPtr q = ts_params();
    const uint32_t tid = threadIdx.x;
    const EuTsPool P = ts_pool(q);
    uint32_t *const rgba = q->rgba;
    eu_f64 *const hit_t_aov = q->hit_t_aov, *const point_rgb = q->point_rgb;
    const uint32_t max_depth = q->cam.max_depth;
    uint32_t nt = 0;
    if (tid == 0) nt = (uint32_t)atomicAdd(&q->counters->next_item, 1ull);      /* the tile after this one: the round trip hides behind the work */
    const unsigned long long total_items = (unsigned long long)q->fr.n_tiles * 64ull;
    const unsigned long long item = (unsigned long long)st.cur_tile * EU_TS_BLOCK + tid;
    const uint32_t rows = q->fr.local_rows, width = q->fr.width, height = q->fr.height;
    bool have_ray = false;
    double o[D], d[D];
    uint32_t out_idx = 0, ent_u = 0;
    do {
        if (item >= total_items) break;
        uint32_t px_x, px_y, ry;
        if (q->fr.single_pixel) {      /* Environment::trace_screen_point: exactly one item */
            if (item != 0) break;
            px_x = q->fr.single_x; px_y = q->fr.single_y; ry = 0; out_idx = 0;
        } else {
            const uint32_t tile = (uint32_t)(item >> 6), within = (uint32_t)(item & 63);   /* 8x8 pixel tiles: coherent waves */
            const uint32_t tiles_x = q->fr.tiles_x;
            px_x = (tile % tiles_x) * 8 + (within & 7);
            ry = (tile / tiles_x) * 8 + (within >> 3);
            if (px_x >= width || ry >= rows) break;
            out_idx = ry * width + px_x;
            const uint32_t strip_count = q->fr.strip_count;
            if (strip_count > 1) {   /* interleaved 8-row strips: this rank owns strips s with s % count == index */
                const uint32_t gstrip = (ry >> 3) * strip_count + q->fr.strip_index;
                px_y = q->fr.row_begin + gstrip * 8 + (ry & 7);
                if (px_y >= q->fr.row_end) {   /* padding rows of the last strip: defined contents */
                    rgba[out_idx] = 0u;
                    if (hit_t_aov) hit_t_aov[out_idx] = -R(1.0);
                    break;
                }
            } else px_y = q->fr.row_begin + ry;
        }
        if (hit_t_aov) hit_t_aov[out_idx] = -R(1.0);
        /* Environment::render's cross-hair (universe/mod.rs:321-333) */
        const uint32_t hw = width / 2, hh = height / 2;
        if (q->fr.debug_crosshair && ((px_x == hw && (px_y == hh - 1 || px_y == hh + 1)) || (px_y == hh && (px_x == hw - 1 || px_x == hw + 1)))) {
            rgba[out_idx] = 0xff0000ffu;
            if (point_rgb) { point_rgb[0] = R(1.0); point_rgb[1] = R(0.0); point_rgb[2] = R(0.0); }
            break;
        }
        const int sw = (int)width, sh = (int)height;
        const double rel_x = (double)((int)px_x - sw / 2) + (double)(1 - sw % 2) / R(2.0);
        const double rel_y = (double)((int)px_y - sh / 2) + (double)(1 - sh % 2) / R(2.0);
        const double dist = q->cam.dist;
        double dl[D];
#pragma unroll
        for (int i = 0; i < D; i++) {
            const double loc = q->cam.location[i];
            const double center = loc + q->cam.forward[i] * dist;
            const double p = center + (q->cam.up[i] * rel_y) + (q->cam.right[i] * rel_x);
            dl[i] = p - loc;
            o[i] = loc;
        }
        vnormalize<D>(dl, d);
        if (cam_ent < 0) {   /* trace_screen_point's checkerboard (universe/mod.rs:387-395) */
            const bool black = (((int)px_x / 8 + (int)px_y / 8) % 2) == 0;
            rgba[out_idx] = black ? 0xff000000u : 0xffff00ffu;
            if (point_rgb) { point_rgb[0] = black ? R(0.0) : R(1.0); point_rgb[1] = R(0.0); point_rgb[2] = black ? R(0.0) : R(1.0); }
            break;
        }
        EuScene SG;
        SG.init(q->scene_g);
        material_apply<D>(SG, SG.entity((uint32_t)cam_ent).material, d, false);
        if (max_depth == 0) {   /* trace() with depth 0 goes straight to the background */
            ts_deliver(P.nodes, out_idx, TS_MODE_ROOT << 1, ts_background<D>(SG, d, cnt), cnt, rgba, point_rgb);
            break;
        }
        ent_u = (uint32_t)cam_ent;
        have_ray = true;
    } while (false);
    uint32_t second;
    const uint32_t pos = ts_reserve(&st.app_pos, have_ray ? 1u : 0u, second);
    if (have_ray) ts_store_ray<D>(P, st, pos, o, d, out_idx, ent_u | ((uint32_t)(TS_MODE_ROOT << 1) << 16));
    if (tid == 0) st.next_tile = nt;
}

/* ---------------------------------------------------------------- import mode: up to 256 rays of the queue segment cur_tile move into this
 * workgroup's open chunk of their generation (records as they are: the parent is a node or pixel of the wavefront pipeline) */
template <int D> EU_DEV void ts_import(TsState &st) {
    TsParamsPtr q = ts_params();
    const uint32_t tid = threadIdx.x;
    const EuTsPool P = ts_pool(q);
    const uint32_t i = st.imp_off + tid;
    const bool have = i < st.imp_count;
    const size_t slot = (size_t)st.cur_tile * q->imp_seg_cap + i;
    double o[D], d[D];
    uint32_t parent = 0, aux = 0;
    if (have) {
        const double *od = q->imp_ray + slot * (size_t)(2 * D);
#pragma unroll
        for (int k = 0; k < D; k++) { o[k] = od[k]; d[k] = od[D + k]; }
        const uint2 pa = q->imp_ray_pa[slot];
        parent = pa.x; aux = pa.y;
    }
    uint32_t second;
    const uint32_t pos = ts_reserve(&st.app_pos, have ? 1u : 0u, second);
    if (have) ts_store_ray<D>(P, st, pos, o, d, parent, aux);
    if (tid == 0) st.imp_off += EU_TS_BLOCK;
}

/* ---------------------------------------------------------------- trace_closest (universe/mod.rs:85-147) for the rays of the current chunk:
 * first hit of every surfaced entity, strict minimum.  Leaves (t, code) in the workgroup's hit row and each lane's sort keys in LDS. */
template <int D, int HSCAP> EU_DEV void ts_intersect(TsState &st, LaneCounters &cnt, uint64_t *lds_dyn) {
    TsParamsPtr q = ts_params();
    const uint32_t tid = threadIdx.x;
    EuScene SG;                  /* wave-uniform addresses: the scene arrives through scalar loads */
    SG.init(q->scene_g);
    const uint32_t gen = st.cur_gen, n = st.cur_count;
    const uint32_t gch = blockIdx.x * q->P.nch + st.cur_chunk;
    const double *const od = q->P.ray_od + (size_t)gch * (2 * D * EU_TS_CH);
    double *const my_hit_t = q->P.hit_t + (size_t)blockIdx.x * EU_TS_CH;
    uint32_t *const my_hit_code = q->P.hit_code + (size_t)blockIdx.x * EU_TS_CH;
    typename std::conditional<HSCAP == 0, HitStackLds, HitStackPriv<(HSCAP ? HSCAP : 1)>>::type HS;
    if constexpr (HSCAP == 0) {
        const uint32_t wave = tid >> 6, lane = tid & 63, hs_cap = q->hs_cap;
        double *hs_t = (double *)(lds_dyn);
        uint32_t *hs_c = (uint32_t *)(hs_t + (EU_TS_BLOCK / 64) * hs_cap * 64);
        HS.t = hs_t + wave * hs_cap * 64 + lane;
        HS.c = hs_c + wave * hs_cap * 64 + lane;
        HS.cap = hs_cap;
    }
    double o_next[D], d_next[D];
#pragma unroll
    for (int k = 0; k < D; k++) { o_next[k] = R(0.0); d_next[k] = R(0.0); }
    if (tid < n) {
#pragma unroll
        for (int k = 0; k < D; k++) { o_next[k] = od[k * EU_TS_CH + tid]; d_next[k] = od[(D + k) * EU_TS_CH + tid]; }
    }
#pragma unroll 1
    for (uint32_t sub = 0; sub < EU_TS_SUB; sub++) {
        const uint32_t i = sub * EU_TS_BLOCK + tid;
        if (i < n) {
            double o[D], d[D];
#pragma unroll
            for (int k = 0; k < D; k++) { o[k] = o_next[k]; d[k] = d_next[k]; }
            if (sub + 1 < EU_TS_SUB && i + EU_TS_BLOCK < n) {      /* the next ray's loads are in flight while this one is intersected */
#pragma unroll
                for (int k = 0; k < D; k++) { o_next[k] = od[k * EU_TS_CH + i + EU_TS_BLOCK]; d_next[k] = od[(D + k) * EU_TS_CH + i + EU_TS_BLOCK]; }
            }
            cnt.rays++;
            bool have = false;
            double best_t = R(0.0);
            uint32_t best_code = 0, best_ent = 0xffffu;
            for (uint32_t e = 0; e < SG.n_entities; e++) {
                const EuScene::EntityView E = SG.entity(e);
                if (E.surface < 0) continue;
                if (E.bound != 0xffffffffu && ray_misses_bound<D>(SG.bounds(E.bound, D), o, d)) continue;
                double t = R(0.0); uint32_t code = 0;
                const uint32_t nh = eval_shape<D>(SG, E.shape_first, E.shape_root, o, d, HS, cnt, t, code);
                if (nh == 0) continue;
                if (!have || best_t > t) { have = true; best_t = t; best_code = code; best_ent = e; }
            }
            my_hit_t[i] = best_t;
            my_hit_code[i] = best_code;
            if (gen == 0) {
                eu_f64 *const hit_t_aov = q->hit_t_aov;
                if (hit_t_aov) hit_t_aov[q->P.ray_parent[(size_t)gch * EU_TS_CH + i]] = have ? best_t : -R(1.0);
            }
            const uint32_t key = best_ent < EU_TS_KEYS - 1 ? best_ent : EU_TS_KEYS - 1;
            st.krk[i] = key | (atomicAdd(&st.hist[key], 1u) << 5) | (best_ent << 16);
        }
    }
}

/* ---------------------------------------------------------------- ComposableSurface::get_color up to the recursive calls (surface.rs:62-162)
 * for the rays of the current chunk, in sorted order: finished colours are delivered, the others leave a node and 1-2 child rays */
template <int D, bool SCENE_LDS> EU_DEV void ts_shade(TsState &st, LaneCounters &cnt, uint64_t *lds_dyn) {
    TsParamsPtr q = ts_params();
    const uint32_t tid = threadIdx.x;
    const EuTsPool P = ts_pool(q);
    uint32_t *const rgba = q->rgba;
    eu_f64 *const point_rgb = q->point_rgb;
    const double time_s = q->fr.time_s;
    const uint32_t gen = st.cur_gen, n = st.cur_count;
    const uint32_t gch = blockIdx.x * P.nch + st.cur_chunk;
    const double *const od = P.ray_od + (size_t)gch * (2 * D * EU_TS_CH);
    const uint32_t *const rpar = P.ray_parent + (size_t)gch * EU_TS_CH;
    const uint32_t *const raux = P.ray_aux + (size_t)gch * EU_TS_CH;
    const double *const my_hit_t = P.hit_t + (size_t)blockIdx.x * EU_TS_CH;
    const uint32_t *const my_hit_code = P.hit_code + (size_t)blockIdx.x * EU_TS_CH;
    /* The hit entity differs from lane to lane, so surface / colour-program records are read with per-lane addresses: from the
     * LDS copy (the L1 is swept by the ray streams); the table offsets come from the header in global memory (scalar registers). */
    EuScene S;
    S.init(q->scene_g);
    double *color_stack = (double *)lds_dyn;       /* surface_color's operand stack: color_depth RGBA entries per lane */
    if constexpr (SCENE_LDS) { S.w = lds_dyn; color_stack = (double *)(lds_dyn + q->scene_words); }
    const uint32_t child_depth = q->cam.max_depth - gen - 1;
    double *const my_cst = color_stack + tid;      /* this lane's colour-operand stack, lane-interleaved */
#pragma unroll 1
    for (uint32_t sub = 0; sub * EU_TS_BLOCK < n; sub++) {
        const uint32_t sidx = sub * EU_TS_BLOCK + tid;
        const bool live = sidx < n;
        const uint32_t sorted_e = live ? st.sorted[sidx] : 0u;
        const uint32_t i = sorted_e & 0xffffu, hit_ent = sorted_e >> 16;
        /* children of this ray: 0 = transmission, 1 = reflection */
        uint32_t n_child = 0;
        double c_o[2][D], c_d[2][D];
        uint32_t c_ent[2] = {0, 0}, c_parent[2] = {0, 0}, c_sm[2] = {0, 0};
        uint32_t node_kind = TS_NONE, node_spx = 0, my_parent = 0, my_sm = 0;
        double node_ratio = R(0.0);
        bool bg_miss = false;
        if (live) {
            const uint32_t parent = rpar[i];
            const uint32_t aux = raux[i];
            const uint32_t ent = aux & 0xffffu, sm = (aux >> 16) & 7u;
            my_parent = parent; my_sm = sm;
            double o[D], d[D];
#pragma unroll
            for (int k = 0; k < D; k++) { o[k] = od[k * EU_TS_CH + i]; d[k] = od[(D + k) * EU_TS_CH + i]; }
            if (hit_ent == 0xffffu) {
                /* nothing hit: the background colour goes to the parent; handled as a depth-0 "child" below */
#pragma unroll
                for (int k = 0; k < D; k++) { c_o[0][k] = o[k]; c_d[0][k] = d[k]; }
                c_parent[0] = parent; c_sm[0] = sm;
                n_child = 1; bg_miss = true;
            } else {
                const double best_t = my_hit_t[i];
                const uint32_t best_code = my_hit_code[i];
                HitCtx<D> c;
                c.finish(best_t, o, d);
                hit_normal<D>(S, best_code, o, d, c.loc, c.normal);
                c.classify();
                const EuScene::EntityView HE = S.entity(hit_ent);
                const EuFlatSurface *F = S.surface((uint32_t)HE.surface);
                double ratio = reflection_ratio<D>(F, c);
                ratio = rust_max(rust_min(ratio, R(1.0)), R(0.0));                          /* surface.rs:145-147 */
                bool have_inter = false, need_trans = false;
                Rgba inter = {R(0.0), R(0.0), R(0.0), R(0.0)};
                uint32_t spx = 0;
                int dest = -1;
                if (!(ratio >= R(1.0))) {                                                /* get_intersection_color, surface.rs:62-117 */
                    const Rgba sc = surface_color<D>(S, F, c, time_s, cnt, my_cst, EU_TS_BLOCK);
                    spx = to_pixel4(sc, cnt);
                    if ((spx >> 24) == 255u) { inter = sc; have_inter = true; }
                    else {
                        threshold_direction<D>(F, c, c_d[0]);
#pragma unroll
                        for (int k = 0; k < D; k++) c_o[0][k] = c.loc[k] + -c.nc[k] * EU_EPS * R(128.0);
                        dest = c.exiting ? material_at<D>(S, c_o[0]) : (int)hit_ent;
                        if (dest >= 0) {
                            material_apply<D>(S, S.entity(ent).material, c_d[0], true);
                            material_apply<D>(S, S.entity((uint32_t)dest).material, c_d[0], false);
                            need_trans = true;
                        }
                    }
                }
                const bool need_refl = !(ratio <= R(0.0));                                /* get_reflection_color, surface.rs:119-139 */
                const uint32_t rs = need_trans ? 1u : 0u;        /* reflection goes to child slot rs in the arrays */
                if (need_refl) {
                    const double dn = vdot<D>(c.dir, c.nc);
#pragma unroll
                    for (int k = 0; k < D; k++) {
                        const double rd = c.nc[k] * -R(2.0) * dn + c.dir[k];              /* surface.rs:246-256 */
                        const double ro = c.loc[k] + c.nc[k] * EU_EPS * R(128.0);
                        if (rs) { c_d[1][k] = rd; c_o[1][k] = ro; } else { c_d[0][k] = rd; c_o[0][k] = ro; }
                    }
                }
                if (need_trans) {
                    node_kind = need_refl ? TS_COMBINE_TRANS : TS_OVER;
                    node_spx = spx; node_ratio = ratio;
                    c_ent[0] = (uint32_t)dest; c_sm[0] = 0u | (TS_MODE_U8 << 1);
                    n_child = 1;
                    if (need_refl) { c_ent[1] = ent; c_sm[1] = 1u | (TS_MODE_F64 << 1); n_child = 2; }
                } else if (need_refl) {
                    c_ent[0] = ent;
                    n_child = 1;
                    if (have_inter) {      /* the opaque surface colour waits in the node (via this lane's idle colour stack) for the reflection */
                        node_kind = TS_COMBINE_INTER; node_ratio = ratio;
                        my_cst[0] = inter.r; my_cst[EU_TS_BLOCK] = inter.g; my_cst[2 * EU_TS_BLOCK] = inter.b; my_cst[3 * EU_TS_BLOCK] = inter.a;
                        c_sm[0] = 1u | (TS_MODE_INTER << 1);
                    } else {   /* the reflection colour is the result (surface.rs:153-154): the child reports to our parent */
                        c_parent[0] = parent; c_sm[0] = sm;
                    }
                } else {
                    if (!have_inter) cnt.errors++;            /* the reference panics here (surface.rs:154) */
                    ts_deliver(P.nodes, parent, sm, inter, cnt, rgba, point_rgb);
                }
            }
        }
        /* tree node of this ray, if its colour needs its children's */
        uint32_t dummy;
        const uint32_t npos = ts_reserve(&st.napp_pos, node_kind != TS_NONE ? 1u : 0u, dummy);
        if (node_kind != TS_NONE) {
            const uint32_t nid = st.napp_chunk[npos / EU_TS_NCN] * EU_TS_NCN + (npos % EU_TS_NCN);
            EuTsNode *N = P.nodes + nid;
            N->ratio = node_ratio; N->spx = node_spx; N->parent = my_parent; N->meta = node_kind | (my_sm << 8);
            if (node_kind == TS_COMBINE_INTER) { N->c1[0] = my_cst[0]; N->c1[1] = my_cst[EU_TS_BLOCK]; N->c1[2] = my_cst[2 * EU_TS_BLOCK]; N->c1[3] = my_cst[3 * EU_TS_BLOCK]; }
            c_parent[0] = nid; c_parent[1] = nid;
        }
        /* children with no depth left (or plain misses) only sample the background (universe/mod.rs:157,183): one code
         * site for all of them (a lane's children are either all background-only or all queued: the miss case has one) */
        const bool bg_only = n_child != 0 && (bg_miss || child_depth == 0);
        const uint32_t n_queue = bg_only ? 0u : n_child;
#pragma unroll 1
        for (uint32_t k = 0; k < 2; k++) {      /* constant indices only: a run-time indexed private array would live in scratch */
            if (bg_only && k < n_child) {
                double dd[D];
#pragma unroll
                for (int qq = 0; qq < D; qq++) dd[qq] = k ? c_d[1][qq] : c_d[0][qq];
                ts_deliver(P.nodes, k ? c_parent[1] : c_parent[0], k ? c_sm[1] : c_sm[0], ts_background<D>(S, dd, cnt), cnt, rgba, point_rgb);
            }
        }
        uint32_t pos1;
        const uint32_t pos0 = ts_reserve(&st.app_pos, n_queue, pos1);
        if (n_queue >= 1) ts_store_ray<D>(P, st, pos0, c_o[0], c_d[0], c_parent[0], c_ent[0] | (c_sm[0] << 16));
        if (n_queue >= 2) ts_store_ray<D>(P, st, pos1, c_o[1], c_d[1], c_parent[1], c_ent[1] | (c_sm[1] << 16));
    }
}

/* ---------------------------------------------------------------- this workgroup's tree nodes, deepest generation first
 * (surface_palette.over(transition_palette), both quantised to u8: surface.rs:104-114; combine: surface.rs:159-161) */
EU_DEV void ts_resolve(TsState &st, LaneCounters &cnt) {
    TsParamsPtr q = ts_params();
    const EuTsPool P = ts_pool(q);
    uint32_t *const rgba = q->rgba;
    eu_f64 *const point_rgb = q->point_rgb;
    for (uint32_t g = q->cam.max_depth; g-- > 0;) {
        uint32_t chunk = st.node_head[g], count = st.node_fill[g];
        while (chunk != EU_TS_NONE) {
            for (uint32_t k = threadIdx.x; k < count; k += EU_TS_BLOCK) {
                const EuTsNode *N = P.nodes + (size_t)chunk * EU_TS_NCN + k;
                const uint32_t meta = N->meta, kind = meta & 0xffu;
                Rgba res = {N->c1[0], N->c1[1], N->c1[2], N->c1[3]};
                if (kind != TS_COMBINE_INTER) {
                    const Rgba over = blend_rgba(EU_BL_OVER, new_u8(N->spx), new_u8(N->c0px));
                    res = kind == TS_OVER ? over : combine_palette_color(res, over, N->ratio);
                }
                ts_deliver(P.nodes, N->parent, (meta >> 8) & 7u, res, cnt, rgba, point_rgb);
            }
            chunk = P.nchunk_prev[chunk]; count = EU_TS_NCN;
        }
        __syncthreads();
    }
}

template <int D, int HSCAP /* 0: per-lane hit stack in LDS (capacity hs_cap); else a private array of HSCAP entries */, bool SCENE_LDS>
__global__ __launch_bounds__(EU_TS_BLOCK, EU_TS_WAVES) void eu_ts_kernel(EuTsParams prm_unused) {
    extern __shared__ uint64_t lds_dyn[];       /* intersect phase: the hit stacks; shade phase: scene copy + colour-operand stacks */
    __shared__ TsState st;
    const uint32_t tid = threadIdx.x;
    LaneCounters cnt = {0, 0, 0, 0};
#ifdef EU_TS_PROFILE      /* wave 0's clock per section (including its waits at the barriers that end the section): 0 generate, 1 intersect, 2 sort, 3 shade, 4 schedule, 5 resolve, 6 start-up; [8] generate steps, [9] chunks, [10] rays in chunks */
    unsigned long long prof[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_amdgcn_s_memtime();
#endif
    int cam_ent;
    {
        TsParamsPtr q = ts_params();
        if (tid < EU_MAX_DEPTH) { st.open_chunk[tid] = EU_TS_NONE; st.open_fill[tid] = 0; st.node_head[tid] = EU_TS_NONE; st.node_fill[tid] = 0; }
        const uint32_t nch = q->P.nch;
        if (tid < nch) st.free_list[tid] = nch - 1 - tid;      /* local chunk ids; the lowest is popped first */
        /* every camera ray starts at the camera (get_ray_point, d3/entity/camera.rs:147-153): material_at(origin) is one value per frame */
        EuScene SG;
        SG.init(q->scene_g);
        double loc[D];
#pragma unroll
        for (int k = 0; k < D; k++) loc[k] = q->cam.location[k];
        cam_ent = material_at<D>(SG, loc);
        __syncthreads();
        if (tid == 0) {
            st.n_ready = 0; st.n_free = nch; st.node_spare = EU_TS_NONE; st.work = TS_WORK_DONE;
            st.imp_off = 0; st.imp_count = 0;
            st.next_tile = (uint32_t)atomicAdd(&q->counters->next_item, 1ull);
            if (q->import_gen != 0xffffffffu && blockIdx.x == 0) {      /* queue lengths behind the hand-over generation belong to an older frame */
                uint32_t *rows = q->imp_seg_count_rows;
                const uint32_t n_seg = q->imp_n_seg;
                for (uint32_t k = (q->import_gen + 1) * n_seg; k < (EU_MAX_DEPTH + 1) * n_seg; k++) rows[k] = 0u;
            }
            ts_schedule(st, q->cam.max_depth, ts_source(q), q->counters, ts_pool(q));
        }
        __syncthreads();
    }
    TS_CLK(6);
    for (;;) {
        const uint32_t work = st.work;
        if (work == TS_WORK_DONE || work == TS_WORK_ABORT) break;
        if (work == TS_WORK_GENERATE) {
            if (ts_params()->import_gen != 0xffffffffu) ts_import<D>(st);
            else ts_generate<D>(st, cnt, cam_ent);
            __syncthreads();
            TS_CLK(0);
#ifdef EU_TS_PROFILE
            prof[8]++;
#endif
        } else {
#ifdef EU_TS_PROFILE
            prof[9]++; prof[10] += st.cur_count;
#endif
            ts_intersect<D, HSCAP>(st, cnt, lds_dyn);
            __syncthreads();
            TS_CLK(1);
            /* counting sort by the entity hit: a wave shades (mostly) one surface.  A wall ray costs ~500 instructions, a glass
             * ray (Fresnel + Snell + rotation) ~2500; unsorted they would share waves. */
            if (tid < 64) {
                const uint32_t v = tid < EU_TS_KEYS ? st.hist[tid] : 0u;
                uint32_t inc = v;
                for (int off = 1; off < EU_TS_KEYS; off <<= 1) { const uint32_t y = __shfl_up(inc, off); if ((int)tid >= off) inc += y; }
                if (tid < EU_TS_KEYS) st.offs[tid] = inc - v;
            }
            if constexpr (SCENE_LDS) {      /* the hit stacks are dead: the shade phase's copy of the scene moves into their place */
                TsParamsPtr q = ts_params();
                const uint64_t *scene_g = q->scene_g;
                const uint32_t scene_words = q->scene_words;
                for (uint32_t k = tid; k < scene_words; k += EU_TS_BLOCK) lds_dyn[k] = scene_g[k];
            }
            __syncthreads();
            for (uint32_t i = tid; i < st.cur_count; i += EU_TS_BLOCK) {
                const uint32_t e = st.krk[i];
                st.sorted[st.offs[e & 31u] + ((e >> 5) & 0x7ffu)] = i | (e & 0xffff0000u);
            }
            __syncthreads();
            TS_CLK(2);
            ts_shade<D, SCENE_LDS>(st, cnt, lds_dyn);
            __syncthreads();
            TS_CLK(3);
        }
        if (tid == 0) {
            TsParamsPtr q = ts_params();
            ts_schedule(st, q->cam.max_depth, ts_source(q), q->counters, ts_pool(q));
        }
        __syncthreads();
        TS_CLK(4);
    }
    if (st.work != TS_WORK_ABORT) ts_resolve(st, cnt);
    TS_CLK(5);

    /* counters: one row per workgroup, summed by the host (same-address atomics drain at ~90 per microsecond) */
    if (tid < 4) st.wg_cnt[tid] = 0ull;
    __syncthreads();
    {
        unsigned long long v0 = cnt.rays, v1 = cnt.bg, v2 = cnt.nan_px, v3 = cnt.errors;
        for (int off = 32; off > 0; off >>= 1) {
            v0 += __shfl_down(v0, off); v1 += __shfl_down(v1, off); v2 += __shfl_down(v2, off); v3 += __shfl_down(v3, off);
        }
        if ((tid & 63) == 0) {
            if (v0) atomicAdd(&st.wg_cnt[0], v0);
            if (v1) atomicAdd(&st.wg_cnt[1], v1);
            if (v2) atomicAdd(&st.wg_cnt[2], v2);
            if (v3) atomicAdd(&st.wg_cnt[3], v3);
        }
    }
    __syncthreads();
    {
        TsParamsPtr q = ts_params();
        unsigned long long *row = q->P.wg_counters + (size_t)blockIdx.x * EU_TS_ROW;
        if (q->stats_counters) {      /* the wavefront pipeline's accounting: a few atomics per workgroup */
            EuDevCounters *c = q->stats_counters;
            if (tid == 0 && st.wg_cnt[0]) atomicAdd(&c->rays, st.wg_cnt[0]);
            if (tid == 1 && st.wg_cnt[1]) atomicAdd(&c->bg_samples, st.wg_cnt[1]);
            if (tid == 2 && st.wg_cnt[2]) atomicAdd(&c->nan_pixels, st.wg_cnt[2]);
            if (tid == 3 && st.wg_cnt[3]) atomicAdd(&c->errors, st.wg_cnt[3]);
            if (tid == 4 && st.work == TS_WORK_ABORT) atomicAdd(&c->overflow, 1ull);
        }
        if (tid < 4) row[tid] = st.wg_cnt[tid];
        if (tid == 4) row[4] = st.work == TS_WORK_ABORT ? 1ull : 0ull;
#ifdef EU_TS_PROFILE
        if (tid == 0) for (int k = 0; k < 11; k++) row[5 + k] = prof[k];
#endif
    }
}

#endif
