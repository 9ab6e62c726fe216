"""Differential test on random scenes: GPU (through the C ABI) == oracle, bit for bit, on CSG / material / surface
combinations the shipped scenes do not contain (SymmetricDifference, nested Complements, cylinders with height, hyperplanes
from vectors, LinearSpace around glass, every blend function, nearest-neighbour textures, camera inside solids, ...)."""
import os

import numpy as np
import pytest

from random_scenes import random_scene

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_case(seed, w, h, depth, time_ms=0, n_entities=None, min_flat_bytes=0):
    from euclider_amd import Parser
    from euclider_amd.environment import EuError
    from oracle.scene_loader import OracleScene, default_texture_loader
    text, dim = random_scene(seed, n_entities=n_entities)
    osc = OracleScene(text, default_texture_loader([ROOT]))
    orgb, ohit, ost = osc.render(w, h, max_depth=depth, time_ms=time_ms, want_hit_t=True)
    try:
        env = Parser(texture_dirs=[ROOT]).parse(text)
    except Exception as e:          # a capacity the kernels were compiled for (reported, never silent)
        pytest.skip("scene %d rejected by the product loader: %s" % (seed, e))
    assert env.info.flat_bytes >= min_flat_bytes
    env.camera.max_depth = depth
    try:
        img = env.render((w, h), time=time_ms / 1000.0, want_hit_t=True)
    except EuError as e:
        env.close()
        if e.code == -5:
            pytest.skip("scene %d exceeds a compiled capacity" % seed)
        raise
    env.close()
    if osc.last_spins:
        # a CSG stream the reference never finishes computing (shape.rs:390-392 under an outer operation that keeps asking, e.g.
        # Intersection(Complement(box, missed cylinder), X) takes the Complement's never-advancing element for ever): its render
        # hangs, so no frame is defined.  (The GPU either flags the entity in eu_stats.errors or, when the ray misses the entity's
        # bounding sphere, never evaluates it.)
        return
    diff = np.argwhere(img.data != orgb)
    assert diff.size == 0, "seed %d: %d differing bytes, first at %s: gpu %s oracle %s" % (
        seed, len(diff), diff[0], img.data[tuple(diff[0][:2])], orgb[tuple(diff[0][:2])])
    assert img.stats == ost, (seed, img.stats, ost)
    both_nan = np.isnan(img.hit_t) & np.isnan(ohit)
    assert np.array_equal(img.hit_t[~both_nan], ohit[~both_nan]), seed


@pytest.mark.parametrize("seed", range(0, 120))
def test_random_scene_parity(seed):
    run_case(seed, 48, 36, 5, time_ms=250 * (seed % 5))


@pytest.mark.parametrize("seed", range(20000, 20040))
def test_random_scene_parity_general_expressions(seed):
    """Seeds >= 20000 also draw LinearSpace expressions with functions, powers, remainders and several variables."""
    run_case(seed, 48, 36, 5)


@pytest.mark.parametrize("seed", range(1000, 1012))
def test_random_scene_parity_deeper_and_larger(seed):
    run_case(seed, 160, 90, 8)


@pytest.mark.parametrize("seed", range(300, 340))
def test_random_scene_trace_path(seed):
    """Camera motion (Universe::trace_path_unknown) through random scenes: GPU == oracle, bit for bit."""
    import random
    from euclider_amd import Parser
    from euclider_amd.environment import EuError
    from oracle.scene_loader import OracleScene, default_texture_loader
    text, dim = random_scene(seed)
    osc = OracleScene(text, default_texture_loader([ROOT]))
    try:
        env = Parser(texture_dirs=[ROOT]).parse(text)
    except Exception as e:
        pytest.skip("scene %d rejected by the product loader: %s" % (seed, e))
    rng = random.Random(seed)
    base = list(env.camera.location)[:dim]
    for k in range(12):
        loc = [b + rng.uniform(-4.0, 14.0 if i == 0 else 4.0) for i, b in enumerate(base)]
        d = [rng.gauss(0.0, 1.0) for _ in range(dim)]
        n = sum(x * x for x in d) ** 0.5
        d = [x / n for x in d]
        dist = rng.choice([0.5, 3.0, 12.0, 40.0])
        try:
            o = osc.trace_path_unknown(dist, loc, d)
        except RuntimeError:
            continue                        # step cap in the oracle: the reference would overflow its stack
        try:
            g = env.trace_path_unknown(dist, loc, d)
        except EuError as e:
            assert e.code in (-5, -8), (seed, k, e.code)
            continue
        assert (g is None) == (o is None), (seed, k)
        if g is not None:
            assert g == o, (seed, k, loc, d, dist, g, o)
    env.close()


@pytest.mark.parametrize("seed,n", [(500, 60), (501, 150), (504, 300), (505, 300)])
def test_random_scene_parity_many_entities(seed, n):
    """Scenes far larger than the shipped ones: the flat scene no longer fits the shade kernel's LDS copy (global variant), the
    sort keys of entities >= 31 share a bucket, hundreds of bounds and materials."""
    run_case(seed, 64, 48, 4, n_entities=n, min_flat_bytes=45 * 1024 if n >= 150 else 0)
