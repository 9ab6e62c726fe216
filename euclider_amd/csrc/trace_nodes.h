/*
 * trace_nodes.h -- the tree nodes of the trace recursion and the delivery of finished colours, used by the wavefront
 * pipeline (trace_wavefront.h).
 *
 * A ray whose colour needs its children's colours (ComposableSurface::get_color: over / combine,
 * /root/reference/src/universe/entity/surface.rs:104-114,159-161) leaves one 64-byte NODE.  A finished colour is DELIVERED to
 * whoever waits for it in the form that one will use: the transmitted colour already quantised to u8 (4 bytes; surface.rs:104-112
 * quantises it first thing), the reflection next to an opaque surface colour already combined with it, a primary ray's colour
 * straight to the RGBA8 pixel (fg.over(white), to_pixel: universe/mod.rs:263-269,342).  Round 1 delivered every colour as four
 * doubles into a 64-byte child array per node next to four more arrays (84 bytes per node) and finished pixels in a pass of its own.
 */
#ifndef EU_TRACE_NODES_H
#define EU_TRACE_NODES_H

#include "trace_device.h"

enum { TS_NONE = 0, TS_OVER = 2, TS_COMBINE_TRANS = 3, TS_COMBINE_INTER = 4 };
/* how a colour is handed to its parent: (slot | mode << 1), carried in bits 16..18 of a ray's aux word */
enum { TS_MODE_F64 = 0, TS_MODE_U8 = 1, TS_MODE_INTER = 2, TS_MODE_ROOT = 3 };

struct EuTsNode {               /* 64 bytes */
    real c1[4];               /* slot 1: the reflection's colour (COMBINE_TRANS) | the surface colour, replaced by the combined result when the reflection arrives (COMBINE_INTER) */
    real ratio;
    uint32_t c0px;              /* slot 0: the transmitted colour, quantised by whoever delivers it */
    uint32_t spx;               /* the surface colour, quantised */
    uint32_t parent;            /* node id, or the pixel's index in the frame buffer */
    uint32_t meta;              /* kind | (slot | mode << 1) of the parent << 8 */
};


template <int D, class P> EU_DEV Rgba ts_background(const EuScene &S, const real *d, LaneCounters &cnt) {
    /* background().get_color(&direction.to_point()) (universe/mod.rs:183) */
    cnt.bg++;
    real pt[D];
#pragma unroll
    for (int i = 0; i < D; i++) pt[i] = R(0.0) + d[i];
    return P::background(S, pt, cnt);
}

/* hand a finished colour to whoever waits for it */
EU_DEV void ts_deliver(EuTsNode *nodes, uint32_t parent, uint32_t slot_mode, const Rgba &c, LaneCounters &cnt,
                       uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {
    const uint32_t mode = slot_mode >> 1;
    if (mode == TS_MODE_ROOT) {             /* trace_unknown: fg.over(white) un-premultiplied, then Rgb::to_pixel (universe/mod.rs:263-269,342) */
        const Rgba white = {R(1.0), R(1.0), R(1.0), R(1.0)};
        const Rgba out = from_premultiplied(blend_pre(EU_BL_OVER, into_premultiplied(c), into_premultiplied(white)));
        rgba[parent] = to_u8(out.r, cnt) | (to_u8(out.g, cnt) << 8) | (to_u8(out.b, cnt) << 16) | 0xff000000u;
        if (point_rgb) { point_rgb[0] = out.r; point_rgb[1] = out.g; point_rgb[2] = out.b; }
        return;
    }
    EuTsNode *N = nodes + parent;
    if (mode == TS_MODE_U8) { N->c0px = to_pixel4(c, cnt); return; }       /* transition_palette = Rgba::new_u8(transition.to_pixel()), surface.rs:104-112 */
    Rgba v = c;
    if (mode == TS_MODE_INTER) {            /* combine_palette_color(reflection, intersection, ratio), surface.rs:159-161 */
        const Rgba inter = {N->c1[0], N->c1[1], N->c1[2], N->c1[3]};
        v = combine_palette_color(c, inter, N->ratio);
    }
    N->c1[0] = v.r; N->c1[1] = v.g; N->c1[2] = v.b; N->c1[3] = v.a;
}



#endif
