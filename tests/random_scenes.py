"""Random scene generator for differential tests (product vs oracle): emits scene JSON in the reference's format
(scene.rs constructor registry) covering shape / CSG / material / surface combinations the shipped scenes do not."""
import json
import random

BLENDS = ["over", "inside", "outside", "atop", "xor", "plus", "multiply", "screen", "overlay", "darken", "lighten", "dodge",
          "burn", "hard_light", "soft_light", "difference", "exclusion"]
OPS = ["Union", "Intersection", "Complement", "SymmetricDifference"]


class Gen:
    def __init__(self, seed, dim):
        self.r = random.Random(seed)
        self.d = dim
        self.no_complement = False
        self.seed_variants = seed >= 20000          # later generator features only for new seed ranges (committed seeds keep their scenes)

    def num(self, lo, hi, grid=0.25):
        """Mostly grid values (exercise ties / axis-aligned degeneracies), sometimes arbitrary."""
        if self.r.random() < 0.7:
            return round(self.r.uniform(lo, hi) / grid) * grid
        return self.r.uniform(lo, hi)

    def point(self, spread=6.0, ahead=10.0):
        p = [self.num(ahead - spread, ahead + spread)] + [self.num(-spread, spread) for _ in range(self.d - 1)]
        return {"Point%d::new" % self.d: p}

    def vector(self, nonzero=True):
        while True:
            v = [self.num(-1.0, 1.0) for _ in range(self.d)]
            if self.r.random() < 0.4:                       # axis aligned
                k = self.r.randrange(self.d)
                v = [(self.r.choice([-1.0, 1.0]) if i == k else 0.0) for i in range(self.d)]
            if not nonzero or any(abs(x) > 1e-3 for x in v):
                return {"Vector%d::new" % self.d: v}

    def rgba(self, alpha=None):
        a = alpha if alpha is not None else self.r.choice([0.0, 0.25, 0.5, 1.0, 1.0])
        if self.r.random() < 0.3:
            return {"Rgba::from_hsva": [self.num(0.0, 360.0, 15.0), self.num(0.0, 1.0), self.num(0.0, 1.0), a]}
        if self.r.random() < 0.2:
            return {"Rgba::new_u8": [self.r.randrange(256), self.r.randrange(256), self.r.randrange(256), int(a * 255)]}
        return {"Rgba::new": [self.num(0.0, 1.0), self.num(0.0, 1.0), self.num(0.0, 1.0), a]}

    # ---- shapes
    def halfspace(self):
        d = self.d
        plane = {"Hyperplane%d::new_with_point" % d: [self.vector(), self.point()]}
        if self.r.random() < 0.25:
            plane = {"Hyperplane%d::new" % d: [self.vector(), self.num(-12.0, 4.0)]}
        if d == 3 and self.r.random() < 0.15:
            plane = {"Hyperplane3::new_with_vectors": [self.vector(), self.vector(), self.point()]}
        if self.r.random() < 0.5:
            return {"HalfSpace%d::new_with_point" % d: [plane, self.point()]}
        return {"HalfSpace%d::new" % d: [plane, self.r.choice([-1.0, 1.0])]}

    def leaf(self):
        d, k = self.d, self.r.random()
        if k < 0.30:
            return {"Sphere%d::new" % d: [self.point(), self.num(0.5, 4.0)]}
        if k < 0.50:
            return self.halfspace()
        if k < 0.72:
            dims = {"Vector%d::new" % d: [self.num(1.0, 6.0) for _ in range(d)]}
            return {("HalfSpace3::cuboid" if d == 3 else "HalfSpace4::hypercuboid"): [self.point(), dims]}
        if k < 0.86:
            return {"Cylinder%d::new" % d: [self.point(), self.vector(), self.num(0.5, 2.5)]}
        if k < 0.97:
            return {"Cylinder%d::new_with_height" % d: [self.point(), self.vector(), self.num(0.5, 2.5), self.num(1.0, 6.0)]}
        return {"VoidShape%d::new" % d: []}

    def shape(self, depth=0):
        if depth >= 2 or self.r.random() < 0.45:
            return self.leaf()
        n = self.r.choice([2, 2, 2, 3])
        parts = [self.shape(depth + 1) for _ in range(n)]
        ops = [o for o in OPS if o != "Complement"] if self.no_complement else OPS     # Complement can make streams that never end
        return {"ComposableShape%d::of" % self.d: [parts, {"SetOperation": [self.r.choice(ops)]}]}

    # ---- materials / surfaces
    def material(self):
        d = self.d
        if self.r.random() < (0.6 if self.seed_variants else 0.75):
            return {"Vacuum%d::new" % d: []}
        legend = "xyzw"[:d]
        exprs = []
        for c in legend:
            k = self.r.choice([1, 1, 2, 4, 0.5])
            form = self.r.random()
            if self.r.random() < 0.2 and self.seed_variants:           # the evaluator is general (meval subset): exercise it
                o = self.r.choice(legend)
                e, i = self.r.choice([("-%s" % c, "-%s" % c), ("%s + %s / 4" % (c, o), "%s - %s / 4" % (c, o)),
                                      ("abs(%s) * 2" % c, "%s / 2" % c), ("max(%s, %s)" % (c, o), "min(%s, %s)" % (c, o)),
                                      ("%s ^ 2 * 0.1 + %s" % (c, c), "sqrt(abs(%s))" % c), ("sin(%s) + %s * 2" % (c, c), "%s / 2 - cos(%s) * 0" % (c, c)),
                                      ("(%s + 1) * 2 - 2" % c, "%s %% 7 / 2" % c), ("floor(%s * 4) / 4" % c, "ceil(%s) - signum(%s) * 0" % (c, c)),
                                      ("atan2(%s, 2) + %s" % (c, c), "%s * pi / e" % c)])
            elif k == 1:
                e, i = c, c
            elif form < 0.5:
                e, i = "%s * %s" % (c, k), "%s / %s" % (c, k)
            else:
                e, i = "%s / %s" % (c, k), "%s * %s" % (c, k)
            exprs.append({"ComponentTransformationExpr": {"expression": e, "inverse_expression": i}})
        return {"LinearSpace%d" % d: {"legend": legend, "transformations": [{"ComponentTransformation%d" % d: {"expressions": exprs}}]}}

    def color(self, depth=0):
        d, k = self.d, self.r.random()
        if depth < 2 and k < 0.30:
            fn = {"blend_function_" + self.r.choice(BLENDS): []} if self.r.random() < 0.85 else {"blend_function_ratio": [self.num(0.0, 1.0)]}
            return {"surface_color_blend_%d" % d: [self.color(depth + 1), self.color(depth + 1), fn]}
        if k < 0.50:
            return {"surface_color_uniform_%d" % d: [self.rgba()]}
        if k < 0.65:
            return {"surface_color_illumination_global_%d" % d: [self.rgba(), self.rgba()]}
        if k < 0.80:
            return {"surface_color_illumination_directional_%d" % d: [self.vector(), self.rgba(), self.rgba()]}
        if d == 3 and k < 0.90:
            return {"surface_color_perlin_hue_seed_3": [self.r.randrange(1000), self.num(0.5, 4.0), self.num(0.0, 2.0)]}
        return {"surface_color_texture_%d" % d: [self.mapped()]}

    def mapped(self):
        uv = {"uv_sphere_3": [{"Point3::new": [self.num(-2.0, 2.0) for _ in range(3)]}]}
        if self.d == 4:
            uv = {"uv_derank_4": [uv]}
        tex = {self.r.choice(["texture_image_linear", "texture_image_nearest_neighbor"]):
               [self.r.choice(["./resources/pixelcg_uv.jpg", "./resources/simple.png"])]}
        return {"MappedTextureImpl%d::new" % self.d: [uv, tex]}

    def surface(self):
        d = self.d
        ratio = {"reflection_ratio_uniform_%d" % d: [self.r.choice([0, 0, 0.25, 0.5, 1])]}
        if self.r.random() < 0.35:
            ratio = {"reflection_ratio_fresnel_%d" % d: [self.r.choice([1.33, 1.458, 2.4]), 1]}
        thr = {"threshold_direction_identity_%d" % d: []}
        if self.r.random() < 0.35:
            thr = {"threshold_direction_snell_%d" % d: [self.r.choice([1.1, 1.458, 0.8])]}
        return {"ComposableSurface%d" % d: {"reflection_ratio": ratio, "reflection_direction": {"reflection_direction_specular_%d" % d: []},
                                            "threshold_direction": thr, "surface_color": self.color()}}

    def scene(self, n_entities=None):
        d = self.d
        self.no_complement = n_entities is not None      # many-entity scenes: keep every entity's stream finite
        ents = []
        for _ in range(n_entities or self.r.randint(1, 5)):
            if self.r.random() < 0.08:
                ents.append({"Entity%dImpl::new_without_surface" % d: [self.shape(), self.material()]})
            else:
                ents.append({"Entity%dImpl::new_with_surface" % d: [self.shape(), self.material(), self.surface()]})
        if self.r.random() < 0.9:
            ents.append({"Void%d::new_with_vacuum" % d: []})
        cam_kind = "FreeCamera4" if d == 4 else self.r.choice(["PitchYawCamera3", "FreeCamera3"])
        loc = {"Point%d::new" % d: [self.num(-3.0, 3.0)] + [self.num(-2.0, 2.0) for _ in range(d - 1)]}
        camera = {cam_kind + "::new_with_location": [loc]} if self.r.random() < 0.8 else {cam_kind + "::new": []}
        return json.dumps({"Universe%d" % d: {"camera": camera, "entities": ents, "background": self.mapped()}})


def random_scene(seed, dim=None, n_entities=None):
    dim = dim or (3 if seed % 3 else 4)
    return Gen(seed, dim).scene(n_entities), dim
