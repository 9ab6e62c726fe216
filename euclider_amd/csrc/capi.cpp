/*
 * capi.cpp -- scene half of the C ABI (include/euclider_amd.h): parse, flatten, inspect.
 * The render half lives in renderer.hip.  Nothing here traces rays.
 */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "../../include/euclider_amd.h"
#include "scene_host.hpp"
#include "jit.hpp"

using namespace euclider;

static void set_err(char *err, size_t errlen, const std::string &msg) {
    if (err && errlen) snprintf(err, errlen, "%s", msg.c_str());
}

extern "C" void *eu_alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
extern "C" void eu_free(void *p) { free(p); }
extern "C" const char *eu_version(void) { return "euclider_amd 0.1 (gfx950)"; }

extern "C" int eu_scene_from_json(const char *json, size_t len, const eu_load_opts *opts, eu_scene **out, char *err, size_t errlen) {
    if (!json || !out) return EU_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    try {
        Parser parser = Parser::make_default(opts);
        auto universe = parser.parse(json, len);
        eu_scene *s = new eu_scene();
        s->universe = universe;
        s->flat = flatten(*universe);
        *out = s;
        return EU_OK;
    } catch (const ParserError &e) {
        set_err(err, errlen, e.what());
        return EU_ERR_PARSE;
    } catch (const std::bad_alloc &) {
        set_err(err, errlen, "out of memory");
        return EU_ERR_PARSE;
    } catch (...) {
        set_err(err, errlen, "unexpected failure while loading the scene");
        return EU_ERR_PARSE;
    }
}

extern "C" void eu_scene_free(eu_scene *s) { delete s; }

extern "C" int eu_scene_get_info(const eu_scene *s, eu_scene_info *info) {
    if (!s || !info) return EU_ERR_INVALID_ARGUMENT;
    *info = s->flat.info;
    return EU_OK;
}

extern "C" int eu_scene_default_camera(const eu_scene *s, eu_camera *out) {
    if (!s || !out) return EU_ERR_INVALID_ARGUMENT;
    *out = s->universe->camera;
    return EU_OK;
}

extern "C" const void *eu_scene_flat(const eu_scene *s, size_t *bytes) {
    if (!s) return nullptr;
    if (bytes) *bytes = s->flat.words.size() * 8;
    return s->flat.words.data();
}

/* ---- scene-specialised kernels (jit.hpp): source and ahead-of-time compilation, no GPU involved ---- */
extern "C" int eu_scene_jit_source_opts(const eu_scene *scene, const char *jit_flags, unsigned renderer_flags, char **source, char *key) {
    if (!scene || !source) return EU_ERR_INVALID_ARGUMENT;
    const euclider::JitPlan plan = euclider::jit_generate(scene->flat, jit_flags ? jit_flags : "", !(renderer_flags & EU_RENDERER_NO_FUSE));
    char *buf = (char *)eu_alloc(plan.source.size() + 1);
    if (!buf) return EU_ERR_INVALID_ARGUMENT;
    memcpy(buf, plan.source.c_str(), plan.source.size() + 1);
    *source = buf;
    if (key) snprintf(key, 40, "%s", plan.key.c_str());
    return EU_OK;
}

extern "C" int eu_scene_jit_source(const eu_scene *scene, char **source, char *key) { return eu_scene_jit_source_opts(scene, nullptr, 0u, source, key); }

extern "C" int eu_scene_jit_precompile_opts(const eu_scene *scene, const char *cache_dir, const char *jit_flags, unsigned renderer_flags, eu_jit_info *info, char *err, size_t errlen) {
    if (!scene) return EU_ERR_INVALID_ARGUMENT;
    const euclider::JitPlan plan = euclider::jit_generate(scene->flat, jit_flags ? jit_flags : "", !(renderer_flags & EU_RENDERER_NO_FUSE));
    euclider::JitBuild b;
    const int rc = euclider::jit_build(plan, cache_dir ? cache_dir : "", b);
    if (info) {
        memset(info, 0, sizeof *info);
        info->requested = 1; info->from_cache = b.from_cache ? 1 : 0; info->hit_stack_entries = plan.hs_cap; info->compile_ms = b.compile_ms;
        snprintf(info->key, sizeof info->key, "%s", plan.key.c_str());
    }
    if (rc != EU_OK || !b.log.empty()) set_err(err, errlen, b.log);      /* (on success: the compiler's warnings and the generator's notes, if any) */
    return rc;
}
extern "C" int eu_scene_jit_precompile(const eu_scene *scene, const char *cache_dir, eu_jit_info *info, char *err, size_t errlen) {
    return eu_scene_jit_precompile_opts(scene, cache_dir, nullptr, 0u, info, err, errlen);
}

