"""ctypes binding of include/euclider_amd.h (libeuclider_amd.so, built in-tree by csrc/Makefile).

The library is the product; this module only declares its symbols.  There is no Python or CPU
implementation of the trace path: if the shared library is missing the import fails loudly.
"""
import ctypes as C
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EU_LIB_PATH") or os.path.join(HERE, "libeuclider_amd.so")    # EU_LIB_PATH: A/B builds (diagnostics)
LIB_PATH_F32 = os.path.join(HERE, "libeuclider_amd_f32.so")      # F = f32: the reference's `low_precision` cargo feature is a second binary

EU_OK = 0
EU_ERR_INVALID_ARGUMENT = -1
EU_ERR_PARSE = -2
EU_ERR_NO_DEVICE = -3
EU_ERR_HIP = -4
EU_ERR_CAPACITY = -5
EU_ERR_TEXTURE = -6
EU_ERR_UNIMPLEMENTED = -7
EU_ERR_PATH_STEPS = -8
EU_ERR_BUSY = -9

EU_CAMERA_PITCH_YAW_3, EU_CAMERA_FREE_3, EU_CAMERA_FREE_4 = 0, 1, 2
KEYS = {name: 1 << i for i, name in enumerate(
    ["W", "S", "A", "D", "LShift", "LControl", "Q", "E", "C", "M", "I", "O", "K", "L"])}


class Camera(C.Structure):
    _fields_ = [("dim", C.c_int32), ("fov_deg", C.c_uint32), ("max_depth", C.c_uint32), ("kind", C.c_uint32),
                ("location", C.c_double * 4), ("forward", C.c_double * 4), ("up", C.c_double * 4),
                ("left", C.c_double * 4)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32),
                ("time_ms", C.c_uint64), ("debug_crosshair", C.c_int32), ("strip_count", C.c_uint32),
                ("strip_index", C.c_uint32), ("reserved", C.c_uint32)]


class Input(C.Structure):
    _fields_ = [("keys", C.c_uint32), ("delta_mouse_x", C.c_int32), ("delta_mouse_y", C.c_int32),
                ("reserved", C.c_uint32), ("delta_time_ms", C.c_uint64), ("mouse_sensitivity", C.c_double),
                ("speed", C.c_double)]


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("bg_samples", C.c_uint64), ("nan_pixels", C.c_uint64), ("errors", C.c_uint64)]


class SceneInfo(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_entities", C.c_uint32), ("n_shape_ops", C.c_uint32), ("n_leaves", C.c_uint32),
                ("n_materials", C.c_uint32), ("n_surfaces", C.c_uint32), ("n_color_ops", C.c_uint32),
                ("n_textures", C.c_uint32), ("hit_cap", C.c_uint32), ("list_depth", C.c_uint32),
                ("flat_bytes", C.c_uint32)]


class RendererOpts(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("kernel", C.c_uint32), ("specialize", C.c_uint32), ("streams", C.c_uint32),
                ("ray_factor", C.c_double), ("band_pixels", C.c_uint64), ("cache_dir", C.c_char_p), ("flags", C.c_uint32),
                ("reserved", C.c_uint32), ("jit_flags", C.c_char_p), ("band_grid_permille", C.c_uint32), ("split_pixels", C.c_uint32)]


class JitInfo(C.Structure):
    _fields_ = [("requested", C.c_int32), ("active", C.c_int32), ("from_cache", C.c_int32), ("hit_stack_entries", C.c_uint32),
                ("compile_ms", C.c_double), ("key", C.c_char * 40)]


EU_KERNEL_AUTO, EU_KERNEL_WAVEFRONT, EU_KERNEL_STACK = 0, 1, 2
EU_SPECIALIZE_AUTO, EU_SPECIALIZE_OFF, EU_SPECIALIZE_SYNC, EU_SPECIALIZE_ASYNC = 0, 1, 2, 3
EU_RENDERER_SHADE_SCENE_GLOBAL = 1
EU_RENDERER_NO_FUSE = 2

TEXTURE_LOADER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                             C.POINTER(C.c_void_p))


class LoadOpts(C.Structure):
    _fields_ = [("load_texture", TEXTURE_LOADER), ("user", C.c_void_p), ("random_seed", C.c_uint32),
                ("reserved", C.c_uint32)]


# every symbol include/euclider_amd.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "eu_alloc": (C.c_void_p, [C.c_size_t]),
    "eu_free": (None, [C.c_void_p]),
    "eu_scene_from_json": (C.c_int, [C.c_char_p, C.c_size_t, C.POINTER(LoadOpts), C.POINTER(C.c_void_p), C.c_char_p,
                                      C.c_size_t]),
    "eu_scene_free": (None, [C.c_void_p]),
    "eu_scene_get_info": (C.c_int, [C.c_void_p, C.POINTER(SceneInfo)]),
    "eu_scene_default_camera": (C.c_int, [C.c_void_p, C.POINTER(Camera)]),
    "eu_scene_flat": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "eu_frame_local_rows": (C.c_uint32, [C.POINTER(Frame)]),
    "eu_device_count": (C.c_int, []),
    "eu_renderer_create": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "eu_renderer_create_opts": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(RendererOpts), C.POINTER(C.c_void_p), C.c_char_p,
                                          C.c_size_t]),
    "eu_renderer_destroy": (None, [C.c_void_p]),
    "eu_renderer_jit_info": (C.c_int, [C.c_void_p, C.POINTER(JitInfo)]),
    "eu_renderer_error": (C.c_char_p, [C.c_void_p]),
    "eu_renderer_jit_log": (C.c_char_p, [C.c_void_p]),
    "eu_scene_jit_source": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p]),
    "eu_scene_jit_precompile": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(JitInfo), C.c_char_p, C.c_size_t]),
    "eu_scene_jit_source_opts": (C.c_int, [C.c_void_p, C.c_char_p, C.c_uint, C.POINTER(C.c_void_p), C.c_char_p]),
    "eu_scene_jit_precompile_opts": (C.c_int, [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint, C.POINTER(JitInfo), C.c_char_p, C.c_size_t]),
    "eu_render_device": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_void_p, C.c_void_p,
                                    C.c_void_p]),
    "eu_pack_rgb_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "eu_renderer_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "eu_renderer_kernel_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "eu_renderer_kernel_ms_history": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.c_int]),
    "eu_renderer_retraces": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "eu_renderer_debug_phases": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "eu_renderer_debug_generations": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "eu_renderer_debug_wg_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "eu_render": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_void_p, C.c_void_p,
                            C.POINTER(Stats)]),
    "eu_trace_screen_point": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_int32, C.c_int32,
                                         C.POINTER(C.c_double)]),
    "eu_sequence_create": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "eu_sequence_destroy": (None, [C.c_void_p]),
    "eu_sequence_submit": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame)]),
    "eu_sequence_next": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                    C.POINTER(Stats)]),
    "eu_multi_create": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "eu_multi_create_opts": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_int, C.POINTER(RendererOpts), C.POINTER(C.c_void_p),
                                       C.c_char_p, C.c_size_t]),
    "eu_multi_destroy": (None, [C.c_void_p]),
    "eu_render_multi": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame), C.c_void_p, C.POINTER(C.c_void_p),
                                  C.POINTER(Stats)]),
    "eu_render_multi_begin": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Frame)]),
    "eu_render_multi_end": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(Stats)]),
    "eu_multi_error": (C.c_char_p, [C.c_void_p]),
    "eu_trace_path": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double,
                                 C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "eu_camera_update": (C.c_int, [C.c_void_p, C.POINTER(Camera), C.POINTER(Input)]),
    "eu_selftest_math": (C.c_int, [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "eu_version": (C.c_char_p, []),
}

_lib = None
_lib_f32 = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so.7 (same soname as /opt/rocm's).  Whichever copy a process loads
    first serves both; torch does not find the GPU when it ends up on the system copy, so a process that loads this library
    BEFORE torch initialises would lose torch's GPU.  Loading torch's copy first (when there is one) makes the order
    irrelevant.  EU_HIP_RUNTIME=system skips this."""
    if os.environ.get("EU_HIP_RUNTIME") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def lib(low_precision=False):
    """The product library; low_precision=True: the F = f32 build (same ABI: poses, hit distances and colours stay f64 at the boundary)."""
    global _lib, _lib_f32
    if low_precision:
        if _lib_f32 is None:
            if not os.path.exists(LIB_PATH_F32):
                raise ImportError("%s is missing: build it with `make -C euclider_amd/csrc`" % LIB_PATH_F32)
            _share_hip_runtime_with_torch()
            L = C.CDLL(LIB_PATH_F32)
            for name, (res, args) in SYMBOLS.items():
                f = getattr(L, name)
                f.restype = res
                f.argtypes = args
            _lib_f32 = L
        return _lib_f32
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `make -C euclider_amd/csrc` (or __graft_entry__.build()); "
                              "there is no fallback implementation" % LIB_PATH)
        _share_hip_runtime_with_torch()
        L = C.CDLL(LIB_PATH)
        host_only = os.environ.get("EU_LIB_HOST_ONLY") == "1"      # the sanitizer build (csrc/Makefile `asan`) holds the host side only
        for name, (res, args) in SYMBOLS.items():
            if host_only and not hasattr(L, name):
                continue
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib
