/*
 * eu_render.c -- the C ABI (include/euclider_amd.h) from plain C: load a reference scene file, walk the camera forward
 * through it (Camera::update) and render the frames with frames in flight (eu_sequence_*); the last frame is written as a
 * binary PPM in the reference's row order (row 0 = bottom of the screen, universe/mod.rs:351-356).
 *
 *   gcc -O2 -Iinclude examples/eu_render.c -Leuclider_amd -leuclider_amd -Wl,-rpath,$PWD/euclider_amd -o eu_render
 *   ./eu_render scenes/3d_room.json 640 360 8 out.ppm 10
 *
 * Textures: no decoder is linked here, so eu_load_opts.load_texture stays NULL and the loader substitutes its documented
 * procedural grid for every image the scene names (a Rust host would pass the `image` crate's decoder).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "euclider_amd.h"

static char *read_file(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)n + 1);
    if (buf && fread(buf, 1, (size_t)n, f) != (size_t)n) { free(buf); buf = NULL; }
    fclose(f);
    if (buf) { buf[n] = 0; *len = (size_t)n; }
    return buf;
}

int main(int argc, char **argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s scene.json width height max_depth out.ppm [frames]\n", argv[0]); return 2; }
    const uint32_t width = (uint32_t)atoi(argv[2]), height = (uint32_t)atoi(argv[3]);
    const int frames = argc > 6 ? atoi(argv[6]) : 1;
    size_t len = 0;
    char *json = read_file(argv[1], &len);
    if (!json) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    char err[512] = "";
    eu_load_opts opts;
    memset(&opts, 0, sizeof opts);
    eu_scene *scene = NULL;
    eu_renderer *renderer = NULL;
    eu_sequence *seq = NULL;
    int rc = eu_scene_from_json(json, len, &opts, &scene, err, sizeof err);
    if (rc != EU_OK) { fprintf(stderr, "scene: %d %s\n", rc, err); return 1; }
    rc = eu_renderer_create(scene, 0, &renderer, err, sizeof err);
    if (rc != EU_OK) { fprintf(stderr, "renderer: %d %s (there is no CPU fallback)\n", rc, err); return 1; }
    eu_camera cam;
    eu_scene_default_camera(scene, &cam);
    cam.max_depth = (uint32_t)atoi(argv[4]);
    rc = eu_sequence_create(renderer, width, height, 2, &seq);
    if (rc != EU_OK) { fprintf(stderr, "sequence: %d\n", rc); return 1; }

    eu_frame fr;
    memset(&fr, 0, sizeof fr);
    fr.width = width; fr.height = height; fr.row_begin = 0; fr.row_end = height;
    eu_input in;
    memset(&in, 0, sizeof in);
    in.keys = EU_KEY_W;                 /* walk forward, 16 ms per frame at the reference's speed 10 */
    in.delta_time_ms = 16;

    const uint8_t *rgb = NULL;
    uint32_t w = 0, rows = 0;
    eu_stats st;
    unsigned long long rays = 0;
    int in_flight = 0;
    for (int k = 0; k < frames; k++) {
        if (k > 0) {
            rc = eu_camera_update(renderer, &cam, &in);
            if (rc != EU_OK) { fprintf(stderr, "camera update: %d\n", rc); return 1; }
        }
        if (in_flight == 2) {
            rc = eu_sequence_next(seq, &rgb, &w, &rows, &st);
            if (rc != EU_OK) { fprintf(stderr, "frame: %d\n", rc); return 1; }
            rays += st.rays; in_flight--;
        }
        fr.time_ms = (uint64_t)k * 16u;
        rc = eu_sequence_submit(seq, &cam, &fr);
        if (rc != EU_OK) { fprintf(stderr, "submit: %d\n", rc); return 1; }
        in_flight++;
    }
    while (in_flight) {
        rc = eu_sequence_next(seq, &rgb, &w, &rows, &st);
        if (rc != EU_OK) { fprintf(stderr, "frame: %d\n", rc); return 1; }
        rays += st.rays; in_flight--;
    }
    FILE *out = fopen(argv[5], "wb");
    if (!out) { fprintf(stderr, "cannot write %s\n", argv[5]); return 2; }
    fprintf(out, "P6\n%u %u\n255\n", w, rows);
    fwrite(rgb, 3, (size_t)w * rows, out);
    fclose(out);
    printf("%s: %d frame(s) %ux%u depth %u, %llu rays, camera at (%.17g, %.17g, %.17g, %.17g)\n", eu_version(), frames, w, rows, cam.max_depth,
           rays, cam.location[0], cam.location[1], cam.location[2], cam.location[3]);
    eu_sequence_destroy(seq);
    eu_renderer_destroy(renderer);
    eu_scene_free(scene);
    free(json);
    return 0;
}
