"""F = f32: the reference's `low_precision` cargo feature (/root/reference/Cargo.toml:18-20, src/main.rs:46-49) as a second
build of the same sources (libeuclider_amd_f32.so, csrc/eu_real.h) against the oracle's second build (libeo_oracle_f32.so).
Bar: bit-exact, as for f64 -- RGB8 identical, counters identical, primary hit distance identical (as f32).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")

CASES = [
    ("3d_fresnel.json", 256, 256, 4),
    ("3d_room.json", 320, 180, 8),
    ("3d_hallways.json", 320, 180, 12),
    ("4d_frame.json", 320, 180, 8),
    ("4d_cylinders.json", 160, 90, 8),
    ("3d_frame.json", 160, 90, 10),
    ("3d_fresnel_2.json", 128, 128, 10),
    ("3d_photo.json", 160, 90, 10),
    ("4d_fresnel.json", 128, 128, 10),
    ("4d_room.json", 160, 90, 10),
    ("3d_room.json", 127, 63, 10),
]


@pytest.mark.parametrize("scene,w,h,depth", CASES)
def test_low_precision_scene_parity(scene, w, h, depth):
    from euclider_amd import Parser
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, scene)
    env = Parser(low_precision=True).parse_file(path)
    env.camera.max_depth = depth
    img = env.render((w, h), want_hit_t=True)
    env.close()
    orgb, ohit, ost = load_scene_file(path, variant="f32").render(w, h, max_depth=depth, want_hit_t=True)
    diff = np.argwhere(img.data != orgb)
    assert diff.size == 0, "%d differing bytes, first at %s: gpu %s oracle %s" % (
        len(diff), diff[0], img.data[tuple(diff[0][:2])], orgb[tuple(diff[0][:2])])
    assert img.stats == {k: ost[k] for k in img.stats}
    gh = img.hit_t.astype(np.float32)          # the ABI hands distances out as f64; they were computed in f32
    both_nan = np.isnan(gh) & np.isnan(ohit)
    assert np.array_equal(gh[~both_nan], ohit[~both_nan])


def test_low_precision_differs_from_f64_but_not_wildly():
    """The two builds are different arithmetic: not identical, and close."""
    from euclider_amd import Parser
    path = os.path.join(SCENES, "3d_room.json")
    a = Parser().parse_file(path); a.camera.max_depth = 8
    b = Parser(low_precision=True).parse_file(path); b.camera.max_depth = 8
    fa, fb = a.render((320, 180)), b.render((320, 180))
    a.close(); b.close()
    d = np.abs(fa.data.astype(np.int16) - fb.data.astype(np.int16))
    assert (d != 0).any() and (d != 0).mean() < 0.5 and np.median(d) <= 1


@pytest.mark.parametrize("low_precision", [True, False])
def test_low_precision_partitions_and_full_size(low_precision):
    from euclider_amd import Parser
    from oracle.scene_loader import load_scene_file
    path = os.path.join(SCENES, "3d_room.json")
    env = Parser(low_precision=low_precision).parse_file(path)
    env.camera.max_depth = 8
    W, H = 1920, 1080
    full = env.render((W, H))
    again = env.render((W, H))
    parts = [env.render((W, H), rows=(0, 536)), env.render((W, H), rows=(536, H))]
    multi = env.render_multi((W, H), [0, 0, 0])
    env.close()
    orgb, _, ost = load_scene_file(path, variant="f32" if low_precision else "").render(W, H, max_depth=8)

    def where(a, b):
        d = np.argwhere((a != b).any(axis=2))
        return "%d px, rows %s, first %s" % (len(d), sorted(set(d[:, 0].tolist()))[:12], d[:4].tolist())
    assert np.array_equal(full.data, orgb), "full vs oracle: " + where(full.data, orgb)
    assert full.stats["rays"] == ost["rays"]
    assert np.array_equal(full.data, again.data) and full.stats == again.stats
    assert np.array_equal(np.concatenate([p.data for p in parts], 0), full.data)
    assert np.array_equal(multi.data, orgb), "multi vs oracle: " + where(multi.data, orgb)
    assert np.array_equal(multi.data, full.data), "multi vs full: " + where(multi.data, full.data)
    assert multi.stats == full.stats
