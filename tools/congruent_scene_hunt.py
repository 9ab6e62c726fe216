"""Differential hunt for the generator's runs of congruent entities (jit.cpp: find_runs): rows of entities with the same random shape program at
different places, other entities between them, random surfaces and cameras; specialised kernels against the oracle, bit for bit.
Usage: [HUNT_F32=1] python tools/congruent_scene_hunt.py <first seed> <last seed>"""
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from euclider_amd import Parser  # noqa: E402
from oracle.scene_loader import OracleScene, default_texture_loader  # noqa: E402

OPS = ["Union", "Intersection", "Complement", "SymmetricDifference"]


def of(shapes, op):
    return {"ComposableShape3::of": [shapes, {"SetOperation": [op]}]}


def template(r):
    """a random shape program as a function of its place"""
    kinds = [r.choice(["box", "sphere", "cyl", "capped", "half"]) for _ in range(r.randint(2, 4))]
    dims = [(r.uniform(-0.6, 0.6), r.uniform(-0.6, 0.6), r.uniform(-0.4, 0.6), r.uniform(0.5, 1.6), [r.uniform(-1, 1), r.uniform(-1, 1), r.uniform(0.2, 1)]) for _ in kinds]
    ops = [r.choice(OPS) for _ in kinds[1:]]
    nested = r.random() < 0.5

    def build(x, y, z):
        leaves = []
        for k, (dx, dy, dz, s, v) in zip(kinds, dims):
            c = [x + dx, y + dy, z + dz]
            if k == "box":
                leaves.append({"HalfSpace3::cuboid": [{"Point3::new": c}, {"Vector3::new": [2 * s, 1.5 * s, 2.5 * s]}]})
            elif k == "sphere":
                leaves.append({"Sphere3::new": [{"Point3::new": c}, s]})
            elif k == "cyl":
                leaves.append({"Cylinder3::new": [{"Point3::new": c}, {"Vector3::new": v}, 0.4 * s]})
            elif k == "capped":
                leaves.append({"Cylinder3::new_with_height": [{"Point3::new": c}, {"Vector3::new": v}, 0.5 * s, 3 * s]})
            else:
                leaves.append({"HalfSpace3::new_with_point": [{"Hyperplane3::new_with_point": [{"Vector3::new": v}, {"Point3::new": c}]},
                                                              {"Point3::new": [c[0] - v[0], c[1] - v[1], c[2] - v[2]]}]})
        if nested and len(leaves) >= 3:
            return of([of(leaves[:2], ops[0])] + leaves[2:], ops[1])
        acc = leaves[0]
        for leaf, op in zip(leaves[1:], ops):
            acc = of([acc, leaf], op)
        return acc
    return build


SURFACES = [
    {"reflection_ratio": {"reflection_ratio_fresnel_3": [1.458, 1]}, "reflection_direction": {"reflection_direction_specular_3": []},
     "threshold_direction": {"threshold_direction_snell_3": [1.458]}, "surface_color": {"surface_color_uniform_3": [{"Rgba::new": [0.1, 0.3, 0.2, 0.25]}]}},
    {"reflection_ratio": {"reflection_ratio_uniform_3": [0.4]}, "reflection_direction": {"reflection_direction_specular_3": []},
     "threshold_direction": {"threshold_direction_identity_3": []},
     "surface_color": {"surface_color_illumination_global_3": [{"Rgba::new": [1, 0.9, 0.5, 1]}, {"Rgba::new": [0.1, 0, 0.2, 1]}]}},
    {"reflection_ratio": {"reflection_ratio_uniform_3": [0]}, "reflection_direction": {"reflection_direction_specular_3": []},
     "threshold_direction": {"threshold_direction_identity_3": []},
     "surface_color": {"surface_color_illumination_directional_3": [{"Vector3::new": [0.3, -0.5, -1]}, {"Rgba::new": [0.5, 0.8, 1, 0.6]}, {"Rgba::new": [0.2, 0.1, 0.1, 1]}]}},
]


def entity(shape, kind):
    return {"Entity3Impl::new_with_surface": [shape, {"Vacuum3::new": []}, {"ComposableSurface3": SURFACES[kind]}]}


def scene(seed):
    r = random.Random(seed)
    ents = []
    runs = 0
    for _ in range(r.randint(1, 3)):
        build = template(r)
        n = r.randint(2, 7)
        x0, y0, z0 = r.uniform(5, 9), r.uniform(-8, 2), r.uniform(-1.5, 1.5)
        sx, sy = r.uniform(2.5, 4.5), r.uniform(-1, 3)
        ents += [entity(build(x0 + sx * k, y0 + sy * k, z0), r.randrange(3)) for k in range(n)]
        runs += 1
        if r.random() < 0.6:
            ents.append(entity({"Sphere3::new": [{"Point3::new": [r.uniform(6, 20), r.uniform(-9, 9), r.uniform(-3, 3)]}, r.uniform(0.5, 2)]}, r.randrange(3)))
    text = json.dumps({"Universe3": {"camera": {"FreeCamera3": []}, "entities": ents + [{"Void3::new_with_vacuum": []}],
                                     "background": {"MappedTextureImpl3::new": [{"uv_sphere_3": [{"Point3::new": [0, 0, 0]}]},
                                                                               {"texture_image_nearest_neighbor": ["./resources/simple.png"]}]}}})
    cam = [r.uniform(-2, 6), r.uniform(-3, 3), r.uniform(-1, 1)]
    return text, cam


lo, hi = int(sys.argv[1]), int(sys.argv[2])
F32 = os.environ.get("HUNT_F32") == "1"      # the low_precision pair: libeuclider_amd_f32.so against libeo_oracle_f32.so
bad, loops, skipped, undefined = [], 0, 0, 0
for seed in range(lo, hi):
    text, cam = scene(seed)
    try:
        osc = OracleScene(text, default_texture_loader([ROOT]), variant="f32" if F32 else "")
        env = Parser(texture_dirs=[ROOT], low_precision=F32).parse(text).configure(specialize="sync")
    except Exception:
        skipped += 1
        continue
    src, _ = env.jit_source()
    loops += src.count("congruent, ops")
    env.camera.max_depth = 5
    ocam = osc.camera()
    for k in range(3):
        env.camera.location[k] = cam[k]
        ocam.location[k] = cam[k]
    orgb, ohit, ost = osc.render(96, 54, max_depth=5, want_hit_t=True, camera=ocam, threads=8)
    try:
        img = env.render((96, 54), want_hit_t=True)
    except Exception as e:
        skipped += 1
        env.close()
        continue
    active = env.jit_info()["active"]
    env.close()
    if osc.last_spins:
        undefined += 1
        continue
    if not active or not np.array_equal(img.data, orgb) or img.stats != ost:
        bad.append((seed, active, int((img.data != orgb).sum()), img.stats, ost))
    print("seed", seed, "loops so far", loops, "bad", len(bad), flush=True)
print("congruent-entity hunt:", "f32" if F32 else "f64", "seeds", lo, hi, "loop bodies emitted", loops, "bad", len(bad), "undefined", undefined, "skipped", skipped)
for b in bad[:10]:
    print(b)
