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
    real c1[4];               /* slot 1: the reflection's colour (COMBINE_TRANS) | the opaque surface colour the reflection will be combined with (COMBINE_INTER) */
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

/* Hand a finished colour to whoever waits for it -- and, where that is a node with ONE child (TS_OVER: the surface colour over the
 * transmitted colour; TS_COMBINE_INTER: the reflection combined with the opaque surface colour), finish that node on the spot and carry
 * its colour further up: such a node's record was written by an earlier launch (or by this very thread, when the child had no depth
 * left), everything else it needs has just arrived, and nobody else will ever touch it.  Only a node with TWO children
 * (TS_COMBINE_TRANS) has to wait: its slots are filled here and the bottom-up resolve pass of its generation finishes it.  Rounds 1-3
 * parked every colour in its parent and resolved all nodes generation by generation: a scene without reflections (3d_hallways: depth
 * 12) paid twelve dependent launches for nodes that never wait for anything. */
EU_DEV void ts_deliver(EuTsNode *nodes, uint32_t parent, uint32_t slot_mode, const Rgba &c_in, LaneCounters &cnt,
                       uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {
    Rgba c = c_in;
    for (;;) {
        const uint32_t mode = slot_mode >> 1;
        if (mode == TS_MODE_ROOT) {             /* trace_unknown: fg.over(white) un-premultiplied, then Rgb::to_pixel (universe/mod.rs:263-269,342) */
            const Rgba white = {R(1.0), R(1.0), R(1.0), R(1.0)};
            const Rgba out = from_premultiplied(blend_pre(EU_BL_OVER, into_premultiplied(c), into_premultiplied(white)));
            rgba[parent] = to_u8(out.r, cnt) | (to_u8(out.g, cnt) << 8) | (to_u8(out.b, cnt) << 16) | 0xff000000u;
            if (point_rgb) { point_rgb[0] = out.r; point_rgb[1] = out.g; point_rgb[2] = out.b; }
            return;
        }
        EuTsNode *N = nodes + parent;
        if (mode == TS_MODE_F64) {              /* the reflection's slot of a node that also waits for a transmitted colour */
            N->c1[0] = c.r; N->c1[1] = c.g; N->c1[2] = c.b; N->c1[3] = c.a;
            return;
        }
        const uint32_t meta = N->meta;
        if (mode == TS_MODE_U8) {               /* transition_palette = Rgba::new_u8(transition.to_pixel()), surface.rs:104-112 */
            const uint32_t px = to_pixel4(c, cnt);
            if ((meta & 0xffu) != (uint32_t)TS_OVER) { N->c0px = px; return; }      /* (TS_COMBINE_TRANS: the reflection is still to come, or came) */
            c = blend_rgba(EU_BL_OVER, new_u8(N->spx), new_u8(px));                 /* surface_palette.over(transition_palette), surface.rs:113-114 */
        } else {                                /* TS_MODE_INTER: combine_palette_color(reflection, intersection, ratio), surface.rs:159-161 */
            const Rgba inter = {N->c1[0], N->c1[1], N->c1[2], N->c1[3]};
            c = combine_palette_color(c, inter, N->ratio);
        }
        parent = N->parent;
        slot_mode = (meta >> 8) & 7u;
    }
}

/* a node with two children, once both colours are there (the resolve pass) */
EU_DEV void ts_finish_two(EuTsNode *nodes, const EuTsNode *N, LaneCounters &cnt, uint32_t *__restrict__ rgba, eu_f64 *__restrict__ point_rgb) {
    const Rgba refl = {N->c1[0], N->c1[1], N->c1[2], N->c1[3]};
    const Rgba over = blend_rgba(EU_BL_OVER, new_u8(N->spx), new_u8(N->c0px));
    ts_deliver(nodes, N->parent, (N->meta >> 8) & 7u, combine_palette_color(refl, over, N->ratio), cnt, rgba, point_rgb);
}



#endif
