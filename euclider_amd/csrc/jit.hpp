/*
 * jit.hpp -- scene-specialised trace kernels, generated and compiled when a renderer is created.
 *
 * The ahead-of-time kernels (trace_wavefront.h instantiated with EuInterp<D>) answer every question about the scene by
 * walking the flat scene at run time: entity records, shape-program ops, colour programs and RPN code are decoded per ray
 * batch (the reference does the same through trait objects and boxed iterators: /root/reference/src/universe/mod.rs:61-147,
 * universe/entity/shape.rs:548-584).  For ONE scene all of that is known when the scene is loaded.  jit_generate_source()
 * writes a scene policy (struct EuJit) in which every entity's shape program is a straight line of calls with constant kinds,
 * counts and parameters, every surface a function with its provider records as constants, every LinearSpace expression an
 * arithmetic expression; the same kernel bodies are instantiated with it and compiled by hiprtc for gfx950.  Both policies
 * call the same arithmetic (trace_device.h), so the frames are bit-identical (tests/test_gpu_jit.py).
 */
#ifndef EU_JIT_HPP
#define EU_JIT_HPP

#include <atomic>
#include <memory>
#include <string>
#include <vector>

#include "scene_host.hpp"

namespace euclider {

struct JitPlan {
    int dim = 3;
    bool hs_lds = true;          /* intersect kernel: per-lane hit stack in LDS (capacity hs_cap) or private (HSCAP = hs_cap entries) */
    uint32_t hs_cap = 8;
    std::string source;          /* HIP source of the translation unit */
    std::vector<std::string> extra_flags;      /* caller's tuning flags (eu_renderer_opts.jit_flags), part of the key */
    std::string key;             /* hex digest of everything the code object depends on */
    bool fused = true;           /* the module holds the fused kernels (intersect0, fshade0, fshade: one launch per generation) or the two-kernel pipeline's
                                  * (intersect0, intersect, shade0, shade: eu_renderer_opts.flags & EU_RENDERER_NO_FUSE) */
    /* what did not fit the budgets below and is traced / shaded from the flat scene inside the specialised kernels */
    uint32_t n_straight_ops = 0, n_interp_entities = 0, n_generic_surfaces = 0;
    bool color_stack = false;    /* the shade kernels need surface_color's operand stack in dynamic LDS (color_depth RGBA entries per lane) */
};

/* Straight-line code for every entity compiles in time that grows faster than the scene (3d_room: 10 shape ops, 6 s; 4d_cylinders: 248
 * ops, 73 s for the fused kernels; a random scene of 266 ops: 166 s; 600 ops: more than ten minutes).  So the generator has budgets:
 * entities get straight-line code, in entity order, while the shape operations of those that have it stay within kJitOpsBudget, and
 * the first kJitSurfacesBudget distinct surfaces get a function of their own.  The rest of a larger scene is traced and shaded by
 * the interpreter's routines INSIDE the specialised kernels (rounds 2-3 gave such a scene the interpreter kernels altogether). */
constexpr uint32_t kJitOpsBudget = 256, kJitSurfacesBudget = 48;

/* Pure host code (no HIP call): the specialised translation unit for this scene. */
JitPlan jit_generate(const FlatScene &flat, const std::string &extra_flags = std::string(), bool fused = true);

struct JitBuild {
    std::vector<char> code;      /* gfx950 code object */
    bool from_cache = false;
    double compile_ms = 0.0;
    std::string log;
};

/* Compiles the plan with hiprtc (works without a GPU) or fetches the code object from the cache directories: `cache_dir`
 * (read / write; empty: $XDG_CACHE_HOME/euclider_amd or ~/.cache/euclider_amd) and the read-only directory `jit_cache` next
 * to the library (kernels compiled at build time travel with it).  Returns 0 or a negative EU_ERR_* code (log says why). */
int jit_build(const JitPlan &plan, const std::string &cache_dir, JitBuild &out, bool cache_only = false);
/* A cached code object the runtime would not load (a truncated or foreign file): dropped from memory and from the user's cache directory. */
void jit_forget(const JitPlan &plan, const std::string &cache_dir);

/* EU_SPECIALIZE_ASYNC: the compilation runs on a worker thread of this library (one for the process: jobs are served in order) while the
 * renderer traces with the interpreter kernels; the renderer polls `done` when a frame is launched.  A job whose renderer is gone before
 * its turn is dropped; the one in progress always finishes (its code object lands in the cache).  At process exit the library waits for
 * the job in progress (hiprtc must not be torn down under it). */
struct JitJob {
    std::shared_ptr<const FlatScene> flat;
    std::string cache_dir, flags;
    bool fused = true;
    JitPlan plan;
    JitBuild build;
    int rc = 0;
    std::string key;             /* the plan's key: renderers of the same scene (sequence slots, one per device) share ONE job */
    std::atomic<bool> done{false};
    std::atomic<int> waiters{0};  /* renderers that still want the result */
};
std::shared_ptr<JitJob> jit_submit(std::shared_ptr<const FlatScene> flat, const std::string &cache_dir, const std::string &flags, const std::string &key, bool fused);

}  // namespace euclider

#endif
