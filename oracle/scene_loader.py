"""ORACLE (test infrastructure, NOT product code).

ctypes binding of oracle/libeo_oracle.so plus a JSON scene walker that issues the oracle's
constructor calls.  It follows the reference's scene format and constructor registry
(/root/reference/src/scene.rs:457-515 positional/keyed forms, :620-1408 registry,
:1430-1478 single-key constructor objects) but is written independently of the product's
C++ loader (euclider_amd/csrc/scene_loader.cpp) so that the two can be compared.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libeo_oracle.so")


class Camera(C.Structure):
    _fields_ = [("dim", C.c_int), ("location", C.c_double * 4), ("forward", C.c_double * 4),
                ("up", C.c_double * 4), ("left", C.c_double * 4), ("fov_deg", C.c_uint32),
                ("max_depth", C.c_uint32)]


class Frame(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("row_begin", C.c_uint32),
                ("row_end", C.c_uint32), ("time_ms", C.c_uint64), ("debug_crosshair", C.c_int)]


class Input(C.Structure):
    _fields_ = [("keys", C.c_uint32), ("delta_mouse_x", C.c_int32), ("delta_mouse_y", C.c_int32),
                ("delta_time_ms", C.c_uint64), ("mouse_sensitivity", C.c_double), ("speed", C.c_double)]


CAMERA_KINDS = {"PitchYawCamera3": 0, "FreeCamera3": 1, "FreeCamera4": 2}
KEYS = {name: 1 << i for i, name in enumerate(
    ["W", "S", "A", "D", "LShift", "LControl", "Q", "E", "C", "M", "I", "O", "K", "L"])}


class Stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("bg_samples", C.c_uint64), ("nan_pixels", C.c_uint64),
                ("errors", C.c_uint64), ("spins", C.c_uint64)]


class Intersection(C.Structure):
    _fields_ = [("location", C.c_double * 4), ("direction", C.c_double * 4),
                ("normal", C.c_double * 4), ("distance", C.c_double)]


class Camera32(C.Structure):
    """eo_camera of the F = f32 build (libeo_oracle_f32.so: `double` is `float` there, API included)"""
    _fields_ = [("dim", C.c_int), ("location", C.c_float * 4), ("forward", C.c_float * 4),
                ("up", C.c_float * 4), ("left", C.c_float * 4), ("fov_deg", C.c_uint32),
                ("max_depth", C.c_uint32)]


def dvec32(vals, n=4):
    arr = (C.c_float * n)()
    for i, v in enumerate(vals):
        arr[i] = float(v)
    return arr


_libs = {}
VARIANTS = {"": "libeo_oracle.so", "flops": "libeo_oracle_flops.so", "libm": "libeo_oracle_libm.so", "f32": "libeo_oracle_f32.so",
            "f32_libm": "libeo_oracle_f32_libm.so", "asan": "libeo_oracle_asan.so"}


def build(force=False, variant=""):
    path = os.path.join(HERE, VARIANTS[variant])
    if force or not os.path.exists(path):
        subprocess.check_call(["make", "-C", HERE, VARIANTS[variant]], stdout=subprocess.DEVNULL)
    return path


def lib(variant=""):
    """The oracle library.  variant "flops": the same restatement with f64 operation counters (eo_flops_take);
    "libm": elementary functions from the platform libm instead of eo_math.h.  Both are measurement builds."""
    if variant in _libs:
        return _libs[variant]
    L = C.CDLL(build(variant=variant))
    _libs[variant] = L
    real = C.c_float if variant in ("f32", "f32_libm") else C.c_double      # F of this build
    CameraT = Camera32 if variant in ("f32", "f32_libm") else Camera
    dp = C.POINTER(real)
    ip = C.POINTER(C.c_int)
    vp = C.c_void_p

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("eo_scene_new", vp, C.c_int)
    sig("eo_scene_free", None, vp)
    sig("eo_last_error", C.c_char_p, vp)
    sig("eo_shape_void", C.c_int, vp)
    sig("eo_shape_sphere", C.c_int, vp, dp, real)
    sig("eo_shape_hyperplane", C.c_int, vp, dp, real)
    sig("eo_shape_hyperplane_with_point", C.c_int, vp, dp, dp)
    sig("eo_shape_hyperplane_with_vectors", C.c_int, vp, dp, dp, dp)
    sig("eo_shape_halfspace", C.c_int, vp, C.c_int, real)
    sig("eo_shape_halfspace_with_point", C.c_int, vp, C.c_int, dp)
    sig("eo_shape_cuboid", C.c_int, vp, dp, dp)
    sig("eo_shape_cylinder", C.c_int, vp, dp, dp, real)
    sig("eo_shape_cylinder_with_height", C.c_int, vp, dp, dp, real, real)
    sig("eo_shape_composable_of", C.c_int, vp, ip, C.c_int, C.c_int)
    sig("eo_material_vacuum", C.c_int, vp)
    sig("eo_transformation_expr", C.c_int, vp, C.c_char_p, C.c_char_p)
    sig("eo_component_transformation", C.c_int, vp, ip, C.c_int)
    sig("eo_material_linear_space", C.c_int, vp, C.c_char_p, ip, C.c_int)
    sig("eo_reflection_ratio_uniform", C.c_int, vp, real)
    sig("eo_reflection_ratio_fresnel", C.c_int, vp, real, real)
    sig("eo_reflection_direction_specular", C.c_int, vp)
    sig("eo_threshold_direction_identity", C.c_int, vp)
    sig("eo_threshold_direction_snell", C.c_int, vp, real)
    sig("eo_blend_function", C.c_int, vp, C.c_char_p, real)
    sig("eo_color_uniform", C.c_int, vp, dp)
    sig("eo_color_blend", C.c_int, vp, C.c_int, C.c_int, C.c_int)
    sig("eo_color_illumination_global", C.c_int, vp, dp, dp)
    sig("eo_color_illumination_directional", C.c_int, vp, dp, dp, dp)
    sig("eo_color_perlin_hue", C.c_int, vp, C.c_uint32, real, real)
    sig("eo_color_texture", C.c_int, vp, C.c_int)
    sig("eo_uv_sphere", C.c_int, vp, dp)
    sig("eo_uv_derank", C.c_int, vp, C.c_int)
    sig("eo_texture_image", C.c_int, vp, C.c_int, C.c_uint32, C.c_uint32, C.c_char_p)
    sig("eo_mapped_texture", C.c_int, vp, C.c_int, C.c_int)
    sig("eo_surface_composable", C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int)
    sig("eo_entity", C.c_int, vp, C.c_int, C.c_int, C.c_int)
    sig("eo_entity_void", C.c_int, vp, C.c_int)
    sig("eo_universe", C.c_int, vp, C.POINTER(CameraT), ip, C.c_int, C.c_int)
    sig("eo_rgba_from_hsva", None, real, real, real, real, dp)
    sig("eo_default_camera", C.c_int, C.c_int, dp, C.POINTER(CameraT))
    sig("eo_scene_camera", C.c_int, vp, C.POINTER(CameraT))
    sig("eo_render", C.c_int, vp, C.POINTER(CameraT), C.POINTER(Frame), C.c_int, C.c_void_p, C.c_void_p,
        C.POINTER(Stats))
    sig("eo_trace_path_unknown", C.c_int, vp, dp, dp, real, dp, dp)
    sig("eo_camera_update", C.c_int, vp, C.c_int, C.POINTER(CameraT), C.POINTER(Input))
    sig("eo_test_intersect", C.c_int, vp, C.c_int, dp, dp, C.POINTER(Intersection), C.c_int)
    sig("eo_test_is_point_inside", C.c_int, vp, C.c_int, dp)
    sig("eo_test_angle_between", real, C.c_int, dp, dp)
    sig("eo_test_combine_palette_color", None, dp, dp, real, dp)
    sig("eo_test_remainder_f", real, real, real)
    sig("eo_test_remainder_i", C.c_int64, C.c_int64, C.c_int64)
    sig("eo_test_material_enter", None, vp, C.c_int, dp, C.c_int)
    sig("eo_test_general_rotation", None, C.c_int, dp, dp, real, dp)
    sig("eo_test_blend", None, C.c_char_p, dp, dp, dp)
    sig("eo_test_fresnel", real, C.c_int, real, real, dp, dp, C.c_int)
    sig("eo_test_snell", None, C.c_int, real, dp, dp, C.c_int, dp)
    sig("eo_test_to_pixel", None, dp, C.POINTER(C.c_uint8))
    sig("eo_test_math", None, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int)
    sig("eo_test_perlin", real, C.c_uint32, dp)
    sig("eo_flops_take", None, C.POINTER(C.c_uint64 * 4))
    sig("eo_build_flags", C.c_int)
    sig("eo_test_ray", None, C.POINTER(CameraT), C.c_int, C.c_int, C.c_int, C.c_int, dp, dp)
    return L


def dvec(vals, n=4):
    arr = (C.c_double * n)()
    for i, v in enumerate(vals):
        arr[i] = float(v)
    return arr


def ivec(vals):
    return (C.c_int * max(1, len(vals)))(*vals)


def flops_take(variant="flops"):
    """Operations the "flops" build counted since the last call: dict add_mul / div / sqrt / transcendental."""
    out = (C.c_uint64 * 4)()
    lib(variant).eo_flops_take(C.byref(out))
    return {"add_mul": int(out[0]), "div": int(out[1]), "sqrt": int(out[2]), "transcendental": int(out[3])}


def default_threads():
    """Threads a render uses when the caller does not say: the cgroup CPU quota / affinity mask, at most 64."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


class ParserError(Exception):
    """Mirrors scene.rs:524-552 (kind is one of the ParserError variants)."""

    def __init__(self, kind, description):
        super().__init__("%s: %s" % (kind, description))
        self.kind = kind
        self.description = description


def procedural_uv_grid(width=1024, height=512):
    """Deterministic RGBA8 stand-in for the reference's missing resources/universe_dim.jpg
    (.MISSING_LARGE_BLOBS).  Same generator as euclider_amd.textures.procedural_uv_grid;
    duplicated on purpose (the oracle does not import the product)."""
    import numpy as np
    y, x = np.mgrid[0:height, 0:width]
    r = (x * 255 // max(1, width - 1)).astype(np.uint8)
    g = (y * 255 // max(1, height - 1)).astype(np.uint8)
    checker = (((x // 32) + (y // 32)) % 2).astype(np.uint8)
    b = (64 + 128 * checker).astype(np.uint8)
    line = ((x % 64 == 0) | (y % 64 == 0))
    img = np.stack([r, g, b, np.full_like(r, 255)], axis=-1)
    img[line] = (255, 255, 255, 255)
    return np.ascontiguousarray(img)


def default_texture_loader(search_dirs):
    def load(path):
        import numpy as np
        from PIL import Image
        for d in search_dirs:
            p = os.path.normpath(os.path.join(d, path))
            if os.path.exists(p):
                im = Image.open(p)
                # image 0.18 get_pixel on a DynamicImage yields RGBA (alpha 255 for RGB / Luma)
                return np.ascontiguousarray(np.asarray(im.convert("RGBA"), dtype=np.uint8))
        return procedural_uv_grid()
    return load


class OracleScene:
    """Walks a scene JSON (scene.rs Appendix-B format) and builds an eo_scene."""

    def __init__(self, text, texture_loader=None, random_seed=0, variant=""):
        self.variant = variant
        self.L = lib(variant)
        self.Camera = Camera32 if variant in ("f32", "f32_libm") else Camera
        self.dvec = dvec32 if variant in ("f32", "f32_libm") else dvec
        self.real_np = "float32" if variant in ("f32", "f32_libm") else "float64"
        self.random_seed = random_seed
        self.texture_loader = texture_loader or default_texture_loader([os.getcwd()])
        try:
            value = json.loads(text)
        except ValueError as e:
            raise ParserError("SyntaxError", str(e))
        key, _ = self._single(value)
        if key.startswith("Universe3"):
            self.dim = 3
        elif key.startswith("Universe4"):
            self.dim = 4
        else:
            raise ParserError("TypeMismatch", "root must be a Universe3/Universe4, got `%s`" % key)
        self.s = self.L.eo_scene_new(self.dim)
        self._build_registry()
        self.construct("Environment", value)

    def __del__(self):
        try:
            if getattr(self, "s", None):
                self.L.eo_scene_free(self.s)
                self.s = None
        except Exception:
            pass

    # -- helpers
    def _chk(self, h):
        if h < 0:
            raise ParserError("CustomError", self.L.eo_last_error(self.s).decode())
        return h

    @staticmethod
    def _single(value):
        if not isinstance(value, dict) or len(value) != 1:
            raise ParserError("InvalidConstructor", "A constructor must be an object containing a single key")
        (k, v), = value.items()
        return k, v

    def construct(self, expected, value):
        key, data = self._single(value)
        if key not in self.reg:
            raise ParserError("NoDeserializer", "No deserializer registered for key `%s`." % key)
        fields, product, fn = self.reg[key]
        if product != expected:
            raise ParserError("TypeMismatch", "The constructor used (`%s`) has an incorrect type for this field "
                                              "(expected %s, produces %s)." % (key, expected, product))
        args = []
        if isinstance(data, dict):
            for name, ty in fields:
                if name not in data:
                    raise ParserError("MissingField", "Missing field %s with key %s" % (ty, name))
                args.append(self.field(ty, data[name]))
        elif isinstance(data, list):
            it = iter(data)
            for name, ty in fields:
                try:
                    item = next(it)
                except StopIteration:
                    raise ParserError("MissingField", "Missing field of type %s (%s)" % (ty, name))
                args.append(self.field(ty, item))
        else:
            raise ParserError("InvalidConstructor", "The constructor data may only be an array or an object")
        return fn(*args)

    def field(self, ty, value):
        if ty == "F":
            if isinstance(value, bool) or not isinstance(value, (int, float)):
                raise ParserError("TypeMismatch", "Expected `floating point number`")
            return float(value)
        if ty in ("u8", "u32"):
            if isinstance(value, bool) or not isinstance(value, (int, float)) or value != int(value) or value < 0 \
                    or value > (255 if ty == "u8" else 2 ** 32 - 1):
                raise ParserError("TypeMismatch", "Expected `%s`" % ty)
            return int(value)
        if ty == "str":
            if not isinstance(value, str):
                raise ParserError("TypeMismatch", "Expected `string`")
            return value
        if ty.startswith("Vec<"):
            inner = ty[4:-1]
            if not isinstance(value, list):
                raise ParserError("TypeMismatch", "Expected an array")
            return [self.field(inner, v) for v in value]
        return self.construct(ty, value)

    # -- registry (scene.rs:620-1408)
    def _build_registry(self):
        L, s, D = self.L, self.s, self.dim
        dvec, Camera = self.dvec, self.Camera      # the build's F (shadow the module's f64 helpers)
        reg = {}

        def add(names, fields, product, fn):
            for n in names:
                reg[n] = (fields, product, fn)

        for d in (3, 4):
            comps = [("x", "F"), ("y", "F"), ("z", "F")] + ([("w", "F")] if d == 4 else [])
            add(["Point%d" % d, "Point%d::new" % d], comps, "Point%d" % d, lambda *c: list(c))
            add(["Vector%d" % d, "Vector%d::new" % d], comps, "Vector%d" % d, lambda *c: list(c))
        add(["Rgba", "Rgba::new"], [("r", "F"), ("g", "F"), ("b", "F"), ("a", "F")], "Rgba", lambda *c: list(c))
        add(["Rgba::new_u8"], [("r", "u8"), ("g", "u8"), ("b", "u8"), ("a", "u8")], "Rgba",
            lambda *c: [float(v) / 255.0 for v in c])

        def from_hsva(h, sat, val, a):
            out = dvec([0, 0, 0, 0])
            L.eo_rgba_from_hsva(h, sat, val, a, out)
            return list(out)
        add(["Rgba::from_hsva"], [("hue", "F"), ("saturation", "F"), ("value", "F"), ("alpha", "F")], "Rgba", from_hsva)

        def ops(name):
            table = {"Union": 0, "Intersection": 1, "Complement": 2, "SymmetricDifference": 3}
            if name not in table:
                raise ParserError("CustomError", "Invalid `SetOperation`: \"%s\"" % name)
            return table[name]
        add(["SetOperation", "SetOperation::new"], [("name", "str")], "SetOperation", ops)

        d = D
        P, V = "Point%d" % d, "Vector%d" % d
        SH, MAT, SURF, ENT = "Shape%d" % d, "Material%d" % d, "Surface%d" % d, "Entity%d" % d
        ck = self._chk
        add(["Void%d" % d, "Void%d::new" % d], [("material", MAT)], ENT, lambda m: ck(L.eo_entity_void(s, m)))
        add(["Void%d::new_with_vacuum" % d], [], ENT, lambda: ck(L.eo_entity_void(s, ck(L.eo_material_vacuum(s)))))
        add(["Entity%dImpl" % d, "Entity%dImpl::new" % d, "Entity%dImpl::new_with_surface" % d],
            [("shape", SH), ("material", MAT), ("surface", SURF)], ENT,
            lambda sh, m, sf: ck(L.eo_entity(s, sh, m, sf)))
        add(["Entity%dImpl::new_without_surface" % d], [("shape", SH), ("material", MAT)], ENT,
            lambda sh, m: ck(L.eo_entity(s, sh, m, -1)))
        add(["VoidShape%d" % d, "VoidShape%d::new" % d], [], SH, lambda: ck(L.eo_shape_void(s)))
        add(["ComposableShape%d" % d, "ComposableShape%d::new" % d, "ComposableShape%d::of" % d],
            [("shapes", "Vec<%s>" % SH), ("operation", "SetOperation")], SH,
            lambda shapes, op: ck(L.eo_shape_composable_of(s, ivec(shapes), len(shapes), op)))
        add(["Sphere%d" % d, "Sphere%d::new" % d], [("center", P), ("radius", "F")], SH,
            lambda c, r: ck(L.eo_shape_sphere(s, dvec(c), r)))
        add(["Hyperplane%d" % d, "Hyperplane%d::new" % d], [("normal", V), ("constant", "F")], SH,
            lambda n, c: ck(L.eo_shape_hyperplane(s, dvec(n), c)))
        add(["Hyperplane%d::new_with_point" % d], [("normal", V), ("point", P)], SH,
            lambda n, p: ck(L.eo_shape_hyperplane_with_point(s, dvec(n), dvec(p))))
        if d == 3:
            add(["Hyperplane3::new_with_vectors"], [("first", V), ("second", V), ("point", P)], SH,
                lambda a, b, p: ck(L.eo_shape_hyperplane_with_vectors(s, dvec(a), dvec(b), dvec(p))))
        add(["HalfSpace%d" % d, "HalfSpace%d::new" % d], [("plane", SH), ("sign", "F")], SH,
            lambda pl, sg: ck(L.eo_shape_halfspace(s, pl, sg)))
        add(["HalfSpace%d::new_with_point" % d], [("plane", SH), ("point", P)], SH,
            lambda pl, p: ck(L.eo_shape_halfspace_with_point(s, pl, dvec(p))))
        add(["HalfSpace3::cuboid" if d == 3 else "HalfSpace4::hypercuboid"], [("center", P), ("dimensions", V)], SH,
            lambda c, dims: ck(L.eo_shape_cuboid(s, dvec(c), dvec(dims))))
        add(["Cylinder%d" % d, "Cylinder%d::new" % d], [("center", P), ("direction", V), ("radius", "F")], SH,
            lambda c, dr, r: ck(L.eo_shape_cylinder(s, dvec(c), dvec(dr), r)))
        add(["Cylinder%d::new_with_height" % d], [("center", P), ("direction", V), ("radius", "F"), ("height", "F")], SH,
            lambda c, dr, r, h: ck(L.eo_shape_cylinder_with_height(s, dvec(c), dvec(dr), r, h)))
        add(["Vacuum%d" % d, "Vacuum%d::new" % d], [], MAT, lambda: ck(L.eo_material_vacuum(s)))
        add(["ComponentTransformationExpr", "ComponentTransformationExpr::new"],
            [("expression", "str"), ("inverse_expression", "str")], "ComponentTransformationExpr",
            lambda a, b: ck(L.eo_transformation_expr(s, a.encode(), b.encode())))
        add(["ComponentTransformation%d" % d, "ComponentTransformation%d::new" % d],
            [("expressions", "Vec<ComponentTransformationExpr>")], "LinearTransformation%d" % d,
            lambda ex: ck(L.eo_component_transformation(s, ivec(ex), len(ex))))
        add(["LinearSpace%d" % d, "LinearSpace%d::new" % d],
            [("legend", "str"), ("transformations", "Vec<LinearTransformation%d>" % d)], MAT,
            lambda lg, tr: ck(L.eo_material_linear_space(s, lg.encode(), ivec(tr), len(tr))))
        add(["uv_sphere_3"], [("center", "Point3")], "UVFn3", lambda c: ck(L.eo_uv_sphere(s, dvec(c))))
        if d == 4:
            add(["uv_derank_4"], [("uvfn", "UVFn3")], "UVFn4", lambda u: ck(L.eo_uv_derank(s, u)))

        def tex(kind):
            def f(path):
                img = self.texture_loader(path)
                h, w = img.shape[0], img.shape[1]
                return ck(L.eo_texture_image(s, kind, w, h, img.tobytes()))
            return f
        add(["texture_image_nearest_neighbor"], [("path", "str")], "Texture", tex(0))
        add(["texture_image_linear"], [("path", "str")], "Texture", tex(1))
        add(["MappedTextureImpl%d" % d, "MappedTextureImpl%d::new" % d], [("uvfn", "UVFn%d" % d), ("texture", "Texture")],
            "MappedTexture%d" % d, lambda u, t: ck(L.eo_mapped_texture(s, u, t)))
        RR, RD, TD, SC = "ReflectionRatio%d" % d, "ReflectionDirection%d" % d, "ThresholdDirection%d" % d, "SurfaceColor%d" % d
        add(["ComposableSurface%d" % d, "ComposableSurface%d::new" % d],
            [("reflection_ratio", RR), ("reflection_direction", RD), ("threshold_direction", TD), ("surface_color", SC)],
            SURF, lambda a, b, c, e: ck(L.eo_surface_composable(s, a, b, c, e)))
        add(["blend_function_ratio"], [("ratio", "F")], "BlendFunction", lambda r: ck(L.eo_blend_function(s, b"ratio", r)))
        for name in ["over", "inside", "outside", "atop", "xor", "plus", "multiply", "screen", "overlay", "darken",
                     "lighten", "dodge", "burn", "hard_light", "soft_light", "difference", "exclusion"]:
            add(["blend_function_" + name], [], "BlendFunction",
                (lambda nm: (lambda: ck(L.eo_blend_function(s, nm.encode(), 0.0))))(name))
        add(["surface_color_blend_%d" % d], [("source", SC), ("destination", SC), ("blend_function", "BlendFunction")], SC,
            lambda a, b, f: ck(L.eo_color_blend(s, a, b, f)))
        add(["surface_color_illumination_global_%d" % d], [("light_color", "Rgba"), ("dark_color", "Rgba")], SC,
            lambda l, k: ck(L.eo_color_illumination_global(s, dvec(l), dvec(k))))
        add(["surface_color_illumination_directional_%d" % d],
            [("direction", V), ("light_color", "Rgba"), ("dark_color", "Rgba")], SC,
            lambda dr, l, k: ck(L.eo_color_illumination_directional(s, dvec(dr), dvec(l), dvec(k))))
        if d == 3:
            add(["surface_color_perlin_hue_seed_3"], [("seed", "u32"), ("size", "F"), ("speed", "F")], SC,
                lambda sd, sz, sp: ck(L.eo_color_perlin_hue(s, sd, sz, sp)))
            add(["surface_color_perlin_hue_random_3"], [("size", "F"), ("speed", "F")], SC,
                lambda sz, sp: ck(L.eo_color_perlin_hue(s, self.random_seed, sz, sp)))
        add(["reflection_ratio_uniform_%d" % d], [("ratio", "F")], RR, lambda r: ck(L.eo_reflection_ratio_uniform(s, r)))
        add(["reflection_ratio_fresnel_%d" % d], [("refractive_index_inside", "F"), ("refractive_index_outside", "F")], RR,
            lambda a, b: ck(L.eo_reflection_ratio_fresnel(s, a, b)))
        add(["reflection_direction_specular_%d" % d], [], RD, lambda: ck(L.eo_reflection_direction_specular(s)))
        add(["threshold_direction_snell_%d" % d], [("refractive_index", "F")], TD,
            lambda n: ck(L.eo_threshold_direction_snell(s, n)))
        add(["threshold_direction_identity_%d" % d], [], TD, lambda: ck(L.eo_threshold_direction_identity(s)))
        add(["surface_color_uniform_%d" % d], [("color", "Rgba")], SC, lambda c: ck(L.eo_color_uniform(s, dvec(c))))
        add(["surface_color_texture_%d" % d], [("mapped_texture", "MappedTexture%d" % d)], SC,
            lambda m: ck(L.eo_color_texture(s, m)))

        def cam(kind, loc=None):
            c = Camera()
            L.eo_default_camera(d, dvec(loc) if loc is not None else None, C.byref(c))
            self.camera_kind = CAMERA_KINDS[kind]      # which Camera::update applies (scene.rs:1353-1408)
            return c
        for kind in (["PitchYawCamera3", "FreeCamera3"] if d == 3 else ["FreeCamera4"]):
            add([kind, kind + "::new"], [], "Camera%d" % d, lambda kind=kind: cam(kind))
            add([kind + "::new_with_location"], [("location", "Point%d" % d)], "Camera%d" % d,
                lambda loc, kind=kind: cam(kind, loc))

        def universe(camera, entities, background):
            if L.eo_universe(s, C.byref(camera), ivec(entities), len(entities), background) != 0:
                raise ParserError("CustomError", L.eo_last_error(s).decode())
            return True
        add(["Universe%d" % d, "Universe%d::new" % d],
            [("camera", "Camera%d" % d), ("entities", "Vec<%s>" % ENT), ("background", "MappedTexture%d" % d)],
            "Environment", universe)
        self.reg = reg

    # -- rendering
    def camera(self):
        c = self.Camera()
        self.L.eo_scene_camera(self.s, C.byref(c))
        return c

    def trace_path_unknown(self, distance, location, direction):
        """Universe::trace_path_unknown -> (location, direction) | None; raises on the step cap."""
        D = self.dim
        ol, od = dvec([0.0] * 4), dvec([0.0] * 4)
        rc = self.L.eo_trace_path_unknown(self.s, dvec(list(location) + [0.0] * (4 - D)),
                                          dvec(list(direction) + [0.0] * (4 - D)), float(distance), ol, od)
        if rc < 0:
            raise RuntimeError("trace_path: step cap")
        return (tuple(ol)[:D], tuple(od)[:D]) if rc == 1 else None

    def camera_update(self, camera, delta_time_ms, keys=(), delta_mouse=(0, 0), mouse_sensitivity=0.0, speed=0.0,
                      kind=None):
        """Camera::update on `camera` (mutated).  Returns 0, or 1 where the reference hits unimplemented!()."""
        inp = Input(sum(KEYS[k] for k in set(keys)), int(delta_mouse[0]), int(delta_mouse[1]), int(delta_time_ms),
                    mouse_sensitivity, speed)
        rc = self.L.eo_camera_update(self.s, self.camera_kind if kind is None else kind, C.byref(camera), C.byref(inp))
        if rc < 0:
            raise RuntimeError("trace_path: step cap")
        return rc

    def render(self, width, height, max_depth=None, time_ms=0, threads=None, rows=None, camera=None,
               debug_crosshair=False, want_hit_t=False):
        import numpy as np
        cam = camera or self.camera()
        if max_depth is not None:
            cam.max_depth = max_depth
        r0, r1 = rows if rows else (0, height)
        fr = Frame(width, height, r0, r1, time_ms, 1 if debug_crosshair else 0)
        rgb = np.zeros(((r1 - r0), width, 3), dtype=np.uint8)
        hit = np.zeros(((r1 - r0), width), dtype=self.real_np) if want_hit_t else None      # F of the build
        st = Stats()
        threads = threads or default_threads()
        rc = self.L.eo_render(self.s, C.byref(cam), C.byref(fr), threads, rgb.ctypes.data,
                              hit.ctypes.data if hit is not None else None, C.byref(st))
        if rc != 0:
            raise RuntimeError("eo_render failed: %d" % rc)
        stats = {"rays": st.rays, "bg_samples": st.bg_samples, "nan_pixels": st.nan_pixels, "errors": st.errors}
        self.last_spins = st.spins      # CSG streams the reference would never finish: the frame is undefined there
        return rgb, hit, stats


def load_scene_file(path, texture_dirs=None, random_seed=0, variant=""):
    with open(path) as f:
        text = f.read()
    dirs = list(texture_dirs or []) + [os.path.dirname(os.path.dirname(os.path.abspath(path))), os.getcwd()]
    return OracleScene(text, default_texture_loader(dirs), random_seed, variant)
