"""Host-side mirror of the reference's loader + `Environment` seam, over the C ABI.

    Parser().parse(json_text)            <-> scene::Parser::default().parse::<Box<Environment>>()
                                             (/root/reference/src/scene.rs:564,1466-1478)
    Environment.max_depth()              <-> Environment::max_depth        (universe/mod.rs:290)
    Environment.trace_screen_point(...)  <-> Environment::trace_screen_point (universe/mod.rs:291-299)
    Environment.render(dimensions, time, threads, context) -> RawImage2d
                                         <-> Environment::render           (universe/mod.rs:300-357)

All tracing happens in libeuclider_amd.so on the GPU; this file holds no arithmetic.
"""
import ctypes as C
import os

import numpy as np

from . import _capi
from . import textures as _textures


# Process-wide defaults for Environment.renderer() (tests switch the whole suite between the interpreter and the specialised kernels
# with these; an Environment's own configure() wins)
DEFAULT_RENDERER_OPTS = {}


class ParserError(Exception):
    """scene.rs:524-552; `kind` is the variant name."""

    def __init__(self, message):
        super().__init__(message)
        self.kind = message.split(":", 1)[0]


class EuError(RuntimeError):
    def __init__(self, code, message=""):
        super().__init__("euclider_amd error %d %s" % (code, message))
        self.code = code


class SimulationContext:
    """simulation.rs:167-186 (the fields render() reads)."""

    def __init__(self, resolution=1, debugging=False, pressed_keys=(), delta_mouse=(0, 0)):
        self.resolution = resolution
        self.debugging = debugging
        self.pressed_keys = set(pressed_keys)      # names of glutin VirtualKeyCode: "W", "S", "A", "D", "LShift", ...
        self.delta_mouse = tuple(delta_mouse)

    def key_mask(self):
        return sum(_capi.KEYS[k] for k in self.pressed_keys)


class RawImage2d:
    """glium::texture::RawImage2d<u8> with ClientFormat::U8U8U8 (universe/mod.rs:351-356):
    `data` is (height, width, 3) uint8, row 0 first (= bottom row on screen)."""

    def __init__(self, data, width, height):
        self.data, self.width, self.height, self.format = data, width, height, "U8U8U8"


def _duration_to_ms(time):
    """(time * 1000).as_secs() for a duration given in seconds (d3/entity/surface.rs:32)."""
    return int(float(time) * 1000.0) if not isinstance(time, int) else time * 1000


class Environment:
    def __init__(self, scene_handle, info, camera, substituted_textures, low_precision=False):
        self.low_precision = low_precision      # F = f32 (the reference's `low_precision` build): every call goes to that library
        self._L = _capi.lib(low_precision)
        self._scene = scene_handle
        self.info = info
        self.camera = camera          # mutable pose (the reference mutates it in Camera::update)
        self.substituted_textures = substituted_textures
        self._renderers = {}

    # -- lifetime
    def close(self):
        L = self._L
        for m in getattr(self, "_multis", {}).values():
            L.eu_multi_destroy(m)
        self._multis = {}
        for r in self._renderers.values():
            L.eu_renderer_destroy(r)
        self._renderers = {}
        if self._scene:
            L.eu_scene_free(self._scene)
            self._scene = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def dim(self):
        return self.info.dim

    # How renderers are created (eu_renderer_opts): set before the first render.
    #   specialize: "off" (the ahead-of-time kernels interpret the flat scene), "sync" (kernels specialised for this scene are compiled
    #               with hiprtc when the renderer is created, a few seconds the first time, cached on disk), "async" (the same on a worker
    #               thread: frames use the interpreter kernels until the code object is ready), None = the library default
    #   kernel: None / "wavefront" / "stack";  streams, ray_factor, band_pixels: 0 = default;  shade_scene_global: test hook
    def configure(self, specialize=None, kernel=None, streams=0, ray_factor=0.0, band_pixels=0, cache_dir=None, shade_scene_global=False,
                  jit_flags=None, flags=0, band_grid_permille=0, split_pixels=0):
        if self._renderers:
            raise RuntimeError("configure() before the first renderer exists")
        self._opts = dict(specialize=specialize, kernel=kernel, streams=streams, ray_factor=ray_factor, band_pixels=band_pixels,
                          cache_dir=cache_dir, shade_scene_global=shade_scene_global, jit_flags=jit_flags, flags=flags,
                          band_grid_permille=band_grid_permille, split_pixels=split_pixels)
        return self

    def _renderer_opts(self):
        o = {k: v for k, v in DEFAULT_RENDERER_OPTS.items() if k != "on_specialize"}
        o.update({k: v for k, v in getattr(self, "_opts", {}).items() if v not in (None, 0, 0.0, False)})
        spec = {None: _capi.EU_SPECIALIZE_AUTO, "auto": _capi.EU_SPECIALIZE_AUTO, "off": _capi.EU_SPECIALIZE_OFF, "sync": _capi.EU_SPECIALIZE_SYNC,
                "async": _capi.EU_SPECIALIZE_ASYNC}[o.get("specialize")]
        kern = {None: _capi.EU_KERNEL_AUTO, "wavefront": _capi.EU_KERNEL_WAVEFRONT, "stack": _capi.EU_KERNEL_STACK}[o.get("kernel")]
        cache = o.get("cache_dir")
        flags = o.get("jit_flags")
        return _capi.RendererOpts(C.sizeof(_capi.RendererOpts), kern, spec, int(o.get("streams") or 0), float(o.get("ray_factor") or 0.0),
                                  int(o.get("band_pixels") or 0), cache.encode() if cache else None,
                                  (_capi.EU_RENDERER_SHADE_SCENE_GLOBAL if o.get("shade_scene_global") else 0) | int(o.get("flags") or 0), 0,
                                  flags.encode() if flags else None, int(o.get("band_grid_permille") or 0), int(o.get("split_pixels") or 0))

    def renderer(self, device=0):
        if device not in self._renderers:
            L = self._L
            out = C.c_void_p()
            err = C.create_string_buffer(512)
            opts = self._renderer_opts()
            rc = L.eu_renderer_create_opts(self._scene, device, C.byref(opts), C.byref(out), err, len(err))
            if rc != _capi.EU_OK:
                raise EuError(rc, err.value.decode())
            self._renderers[device] = out
            # a process-wide default of "sync" may come with a check (tests: the specialised pass must not pass on the interpreter
            # kernels a failed compilation falls back to); an Environment's own configure(specialize=...) is the caller's business
            check = DEFAULT_RENDERER_OPTS.get("on_specialize")
            if check is not None and opts.specialize == _capi.EU_SPECIALIZE_SYNC and getattr(self, "_opts", {}).get("specialize") is None \
                    and opts.kernel != _capi.EU_KERNEL_STACK:
                check(self, self.jit_info(device))
        return self._renderers[device]

    def jit_info(self, device=0):
        """What eu_renderer_jit_info says about this renderer's scene-specialised kernels."""
        info = _capi.JitInfo()
        rc = self._L.eu_renderer_jit_info(self.renderer(device), C.byref(info))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        return {"requested": bool(info.requested), "active": bool(info.active), "from_cache": bool(info.from_cache),
                "hit_stack_entries": info.hit_stack_entries, "compile_ms": info.compile_ms, "key": info.key.decode(),
                "log": (self._L.eu_renderer_jit_log(self.renderer(device)) or b"").decode(errors="replace")}

    def jit_source(self, jit_flags=None, renderer_flags=0):
        """The HIP source of this scene's specialised kernels (no GPU needed); jit_flags / renderer_flags as in eu_renderer_opts."""
        src, key = C.c_void_p(), C.create_string_buffer(40)
        rc = self._L.eu_scene_jit_source_opts(self._scene, jit_flags.encode() if jit_flags else None, renderer_flags, C.byref(src), key)
        if rc != _capi.EU_OK:
            raise EuError(rc)
        try:
            return C.string_at(src.value).decode(), key.value.decode()
        finally:
            self._L.eu_free(src)

    def jit_precompile(self, cache_dir=None, jit_flags=None, renderer_flags=0):
        """Compile this scene's specialised kernels for gfx950 into the cache (no GPU needed); returns eu_jit_info as a dict."""
        info, err = _capi.JitInfo(), C.create_string_buffer(1 << 16)
        rc = self._L.eu_scene_jit_precompile_opts(self._scene, cache_dir.encode() if cache_dir else None, jit_flags.encode() if jit_flags else None, renderer_flags,
                                                 C.byref(info), err, len(err))
        if rc != _capi.EU_OK:
            raise EuError(rc, err.value.decode())
        return {"from_cache": bool(info.from_cache), "compile_ms": info.compile_ms, "key": info.key.decode(),
                "hit_stack_entries": info.hit_stack_entries, "log": err.value.decode(errors="replace")}

    # -- Environment trait
    def max_depth(self):
        return self.camera.max_depth

    def _frame(self, width, height, time_ms, debugging, rows=None, strips=None):
        r0, r1 = rows if rows is not None else (0, height)
        index, count = strips if strips is not None else (0, 0)
        return _capi.Frame(width, height, r0, r1, time_ms, 1 if debugging else 0, count, index, 0)

    def frame(self, width, height, time=0.0, debugging=False, rows=None, strips=None):
        return self._frame(width, height, _duration_to_ms(time), debugging, rows, strips)

    @staticmethod
    def local_rows(frame):
        return _capi.lib().eu_frame_local_rows(C.byref(frame))      # (integer arithmetic: the same in both builds)

    def trace_screen_point(self, time, max_depth, screen_x, screen_y, screen_width, screen_height, debug=False, device=0):
        cam = _capi.Camera.from_buffer_copy(self.camera)
        cam.max_depth = max_depth
        fr = self._frame(screen_width, screen_height, _duration_to_ms(time), False)
        rgb = (C.c_double * 3)()
        rc = self._L.eu_trace_screen_point(self.renderer(device), C.byref(cam), C.byref(fr), screen_x, screen_y, rgb)
        if rc != _capi.EU_OK:
            raise EuError(rc)
        return tuple(rgb)

    def trace_path_unknown(self, distance, location, direction, device=0):
        """Universe::trace_path_unknown (universe/mod.rs:273-286) -> (location, direction) or None.  Runs on the GPU."""
        D = self.dim
        loc = (C.c_double * 4)(*([float(x) for x in location] + [0.0] * (4 - D)))
        dr = (C.c_double * 4)(*([float(x) for x in direction] + [0.0] * (4 - D)))
        ol, od, found = (C.c_double * 4)(), (C.c_double * 4)(), C.c_int32(0)
        rc = self._L.eu_trace_path(self.renderer(device), loc, dr, float(distance), ol, od, C.byref(found))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        return (tuple(ol)[:D], tuple(od)[:D]) if found.value else None

    def update(self, delta_time, context=None, device=0, mouse_sensitivity=0.0, speed=0.0):
        """Environment::update (universe/mod.rs:359,399-405): Camera::update with the context's keys and mouse delta.
        Mutates self.camera.  Raises EuError(EU_ERR_UNIMPLEMENTED) where the reference panics with unimplemented!()."""
        context = context or SimulationContext()
        inp = _capi.Input(context.key_mask(), int(context.delta_mouse[0]), int(context.delta_mouse[1]), 0,
                          _duration_to_ms(delta_time), mouse_sensitivity, speed)
        moves = inp.delta_time_ms != 0 and (inp.keys & (0xff if self.dim == 4 else 0x3f)) != 0
        rc = self._L.eu_camera_update(self.renderer(device) if moves else None, C.byref(self.camera), C.byref(inp))
        if rc != _capi.EU_OK:
            raise EuError(rc)

    def render(self, dimensions, time=0.0, threads=0, context=None, device=0, want_hit_t=False, rows=None, strips=None):
        """Environment::render.  `threads` is accepted for signature parity and ignored (the GPU
        kernel replaces the thread pool).  Returns RawImage2d; `.stats` and `.hit_t` are extras."""
        context = context or SimulationContext()
        width, height = dimensions
        bw, bh = width // context.resolution, height // context.resolution     # universe/mod.rs:308-309
        fr = self._frame(bw, bh, _duration_to_ms(time), context.debugging, rows, strips)
        nrows = self.local_rows(fr)
        rgb = np.zeros((nrows, bw, 3), dtype=np.uint8)
        hit = np.zeros((nrows, bw), dtype=np.float64) if want_hit_t else None
        st = _capi.Stats()
        rc = self._L.eu_render(self.renderer(device), C.byref(self.camera), C.byref(fr), rgb.ctypes.data,
                                   hit.ctypes.data if hit is not None else None, C.byref(st))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        img = RawImage2d(rgb, bw, nrows)
        img.stats = {"rays": st.rays, "bg_samples": st.bg_samples, "nan_pixels": st.nan_pixels, "errors": st.errors}
        img.hit_t = hit
        return img

    def _multi(self, devices):
        key = tuple(int(d) for d in devices)
        L = self._L
        if not hasattr(self, "_multis"):
            self._multis = {}
        if key not in self._multis:
            out = C.c_void_p()
            err = C.create_string_buffer(512)
            arr = (C.c_int * len(key))(*key)
            opts = self._renderer_opts()
            rc = L.eu_multi_create_opts(self._scene, arr, len(key), C.byref(opts), C.byref(out), err, len(err))
            if rc != _capi.EU_OK:
                raise EuError(rc, err.value.decode())
            self._multis[key] = out
            self._multi_pending = getattr(self, "_multi_pending", {})
            self._multi_pending[key] = []
        return key, self._multis[key]

    def render_multi_begin(self, dimensions, devices, time=0.0, context=None, rows=None):
        """eu_render_multi_begin: queue the frame on all devices and return; at most two frames in flight per device list."""
        context = context or SimulationContext()
        width, height = dimensions
        bw, bh = width // context.resolution, height // context.resolution
        fr = self._frame(bw, bh, _duration_to_ms(time), context.debugging, rows, None)
        key, m = self._multi(devices)
        rc = self._L.eu_render_multi_begin(m, C.byref(self.camera), C.byref(fr))
        if rc != _capi.EU_OK:
            raise EuError(rc, self._L.eu_multi_error(m).decode())
        self._multi_pending[key].append((bw, fr.row_end - fr.row_begin))

    def render_multi_end(self, devices):
        """eu_render_multi_end: the oldest frame begun on this device list."""
        key, m = self._multi(devices)
        if not self._multi_pending[key]:
            raise EuError(_capi.EU_ERR_INVALID_ARGUMENT, "no frame in flight")
        bw, nrows = self._multi_pending[key].pop(0)
        rgb = np.zeros((nrows, bw, 3), dtype=np.uint8)
        st = _capi.Stats()
        rc = self._L.eu_render_multi_end(m, rgb.ctypes.data, None, C.byref(st))
        if rc != _capi.EU_OK:
            raise EuError(rc, self._L.eu_multi_error(m).decode())
        img = RawImage2d(rgb, bw, nrows)
        img.stats = {"rays": st.rays, "bg_samples": st.bg_samples, "nan_pixels": st.nan_pixels, "errors": st.errors}
        return img

    def render_multi(self, dimensions, devices, time=0.0, context=None, rows=None):
        """Environment::render with the frame's 8-row strips dealt round-robin over `devices` (eu_render_multi: one
        process, one renderer per listed device, packed strips gathered on devices[0], row order restored there)."""
        context = context or SimulationContext()
        width, height = dimensions
        bw, bh = width // context.resolution, height // context.resolution
        fr = self._frame(bw, bh, _duration_to_ms(time), context.debugging, rows, None)
        key, m = self._multi(devices)
        L = self._L
        nrows = fr.row_end - fr.row_begin
        rgb = np.zeros((nrows, bw, 3), dtype=np.uint8)
        st = _capi.Stats()
        rc = L.eu_render_multi(m, C.byref(self.camera), C.byref(fr), rgb.ctypes.data, None, C.byref(st))
        if rc != _capi.EU_OK:
            raise EuError(rc, L.eu_multi_error(m).decode())
        img = RawImage2d(rgb, bw, nrows)
        img.stats = {"rays": st.rays, "bg_samples": st.bg_samples, "nan_pixels": st.nan_pixels, "errors": st.errors}
        return img

    def render_device(self, frame, rgba_ptr, hit_t_ptr=None, stream=None, device=0, camera=None):
        """Asynchronous render into caller-owned DEVICE memory (e.g. a torch tensor's data_ptr())."""
        cam = camera if camera is not None else self.camera
        rc = self._L.eu_render_device(self.renderer(device), C.byref(cam), C.byref(frame), stream, rgba_ptr, hit_t_ptr)
        if rc != _capi.EU_OK:
            raise EuError(rc)

    def stats(self, device=0):
        st = _capi.Stats()
        rc = self._L.eu_renderer_stats(self.renderer(device), C.byref(st))
        if rc != _capi.EU_OK:
            raise EuError(rc, (self._L.eu_renderer_error(self.renderer(device)) or b"").decode())
        return {"rays": st.rays, "bg_samples": st.bg_samples, "nan_pixels": st.nan_pixels, "errors": st.errors}

    def kernel_ms(self, device=0):
        ms = C.c_float()
        rc = self._L.eu_renderer_kernel_ms(self.renderer(device), C.byref(ms))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        return ms.value

    def retraces(self, device=0):
        """Frames `render` had to trace a second time (queue / node-pool overflow): the slow path, see eu_renderer_retraces."""
        n = C.c_uint64()
        rc = self._L.eu_renderer_retraces(self.renderer(device), C.byref(n))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        return n.value

    def wg_profile(self, device=0, max_records=1 << 20):
        """Diagnostics (jit_flags -DEU_PROFILE_WG): first call switches the recording on and returns None; later calls return an
        (n, 4) uint64 array {kind | gen << 8 | block << 32, start, end, hw id | grid << 32} of the launches since the previous call."""
        import numpy as np
        r = self.renderer(device)
        if not getattr(self, "_wg_prof_on", False):
            rc = self._L.eu_renderer_debug_wg_profile(r, None, 0, None)
            if rc != _capi.EU_OK:
                raise EuError(rc)
            self._wg_prof_on = True
            return None
        buf = np.zeros((max_records, 4), dtype=np.uint64)
        n = C.c_size_t(0)
        rc = self._L.eu_renderer_debug_wg_profile(r, buf.ctypes.data, max_records, C.byref(n))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        return buf[:n.value].copy()

    def kernel_ms_history(self, n, device=0):
        buf = (C.c_float * max(1, n))()
        got = self._L.eu_renderer_kernel_ms_history(self.renderer(device), buf, n)
        if got < 0:
            raise EuError(got)
        return [buf[i] for i in range(got)]

    def pack_rgb_device(self, rgba_ptr, rgb_ptr, pixels, stream=None, device=0):
        rc = self._L.eu_pack_rgb_device(self.renderer(device), rgba_ptr, rgb_ptr, pixels, stream)
        if rc != _capi.EU_OK:
            raise EuError(rc)


class FrameSequence:
    """The frame loop around Environment::render (simulation.rs:93-150) with `slots` frames in flight: `submit` queues a
    frame (trace + pack + asynchronous read-back into pinned memory) and returns at once, `next` waits for the oldest one.
    The camera pose is sampled at submit time, so `env.update(...)` may move it between submits."""

    def __init__(self, env, max_dimensions, slots=2, device=0):
        self.env, self.device = env, device
        self._h = C.c_void_p()
        rc = self.env._L.eu_sequence_create(env.renderer(device), int(max_dimensions[0]), int(max_dimensions[1]), int(slots),
                                            C.byref(self._h))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        self.in_flight = 0

    def submit(self, dimensions, time=0.0, context=None):
        context = context or SimulationContext()
        bw, bh = dimensions[0] // context.resolution, dimensions[1] // context.resolution     # universe/mod.rs:308-309
        fr = self.env._frame(bw, bh, _duration_to_ms(time), context.debugging)
        rc = self.env._L.eu_sequence_submit(self._h, C.byref(self.env.camera), C.byref(fr))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        self.in_flight += 1

    def next(self, copy=True):
        """RawImage2d of the oldest frame in flight.  With copy=False `.data` aliases a pinned image of the sequence, valid until
        the NEXT call of next() (the sequence rotates slots + 1 images), however many frames are submitted in between."""
        ptr, w, rows, st = C.c_void_p(), C.c_uint32(), C.c_uint32(), _capi.Stats()
        rc = self.env._L.eu_sequence_next(self._h, C.byref(ptr), C.byref(w), C.byref(rows), C.byref(st))
        if rc != _capi.EU_OK:
            raise EuError(rc)
        self.in_flight -= 1
        buf = (C.c_uint8 * (w.value * rows.value * 3)).from_address(ptr.value)
        data = np.frombuffer(buf, dtype=np.uint8).reshape(rows.value, w.value, 3)
        img = RawImage2d(data.copy() if copy else data, w.value, rows.value)
        img.stats = {"rays": st.rays, "bg_samples": st.bg_samples, "nan_pixels": st.nan_pixels, "errors": st.errors}
        return img

    def close(self):
        if self._h:
            self.env._L.eu_sequence_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class Parser:
    """scene::Parser (scene.rs:554-1478).  `texture_dirs`: where relative texture paths are resolved
    (the reference resolves them against the process CWD, scene.rs:1053)."""

    def __init__(self, texture_dirs=None, random_seed=0, texture_overrides=None, low_precision=False):
        self.low_precision = low_precision      # Cargo.toml:18-20 `low_precision`: F = f32
        self.texture_dirs = list(texture_dirs) if texture_dirs is not None else [os.getcwd()]
        self.random_seed = random_seed
        self.texture_overrides = dict(texture_overrides or {})

    @staticmethod
    def default():
        return Parser()

    def parse(self, text):
        L = _capi.lib(self.low_precision)
        keep = []

        def loader(user, path, pw, ph, prgba):
            try:
                p = path.decode()
                img = self.texture_overrides.get(p)
                if img is None:
                    img = _textures.load_rgba(p, self.texture_dirs)
                if img is None:
                    return 1        # the C++ loader substitutes the procedural grid and counts it
                img = np.ascontiguousarray(img, dtype=np.uint8)
                h, w = img.shape[0], img.shape[1]
                buf = L.eu_alloc(w * h * 4)
                C.memmove(buf, img.ctypes.data, w * h * 4)
                pw[0], ph[0] = w, h
                prgba[0] = buf
                return 0
            except Exception:
                return 2

        cb = _capi.TEXTURE_LOADER(loader)
        keep.append(cb)
        opts = _capi.LoadOpts(cb, None, self.random_seed, 0)
        out = C.c_void_p()
        err = C.create_string_buffer(1024)
        data = text.encode() if isinstance(text, str) else bytes(text)
        rc = L.eu_scene_from_json(data, len(data), C.byref(opts), C.byref(out), err, len(err))
        if rc == _capi.EU_ERR_PARSE:
            raise ParserError(err.value.decode())
        if rc != _capi.EU_OK:
            raise EuError(rc, err.value.decode())
        info = _capi.SceneInfo()
        L.eu_scene_get_info(out, C.byref(info))
        cam = _capi.Camera()
        L.eu_scene_default_camera(out, C.byref(cam))
        return Environment(out, info, cam, 0, self.low_precision)

    def parse_file(self, path):
        with open(path) as f:
            text = f.read()
        root = os.path.dirname(os.path.dirname(os.path.abspath(path)))
        saved = self.texture_dirs
        self.texture_dirs = saved + [root]
        try:
            return self.parse(text)
        finally:
            self.texture_dirs = saved
