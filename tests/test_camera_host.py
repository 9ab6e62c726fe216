"""Camera motion ("next" row f3): the oracle's trace_path against hand-derived values, and the product's host-side
Camera::update rotation maths (through the C ABI, no GPU needed: nothing moves) against the oracle's restatement.

Reference: universe/mod.rs:186-227,273-286; surface.rs:164-197; d3/entity/camera.rs:94-145,299-346;
d4/entity/camera.rs:68-136; util.rs:301-322."""
import ctypes as C
import os
import random

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = os.path.join(ROOT, "scenes")
EPS_OFF = 1e-6 * 128.0          # surface.rs:171-174


def oracle_scene(name):
    from oracle.scene_loader import load_scene_file
    return load_scene_file(os.path.join(SCENES, name))


def test_path_vacuum_is_a_straight_line(oracle_lib):
    osc = oracle_scene("3d_hallways.json")
    loc, d = osc.trace_path_unknown(30.0, [0.0, 0.0, 0.0], [0.6, 0.8, 0.0])
    assert loc == (0.0 + 0.6 * 30.0, 0.0 + 0.8 * 30.0, 0.0) and d == (0.6, 0.8, 0.0)


def test_path_through_stretching_portal(oracle_lib):
    """3d_hallways: the cuboid x in [10,30], y in [-6.5,-3.5] is a LinearSpace with x -> 4x.  Walking 40 along +x from
    (0,-5,0): 10 to the near face, the remaining 30 are spent at 4x speed (20 wide -> 5), the last 25 in vacuum: x = 55
    plus the two surface offsets (one of them stretched 4x on the way)."""
    osc = oracle_scene("3d_hallways.json")
    loc, d = osc.trace_path_unknown(40.0, [0.0, -5.0, 0.0], [1.0, 0.0, 0.0])
    assert d == (1.0, 0.0, 0.0)
    assert abs(loc[0] - 55.0) < 8 * EPS_OFF and loc[1] == -5.0 and loc[2] == 0.0
    inside, d_in = osc.trace_path_unknown(12.0, [0.0, -5.0, 0.0], [1.0, 0.0, 0.0])      # stops inside: 10 + 2*4
    assert abs(inside[0] - 18.0) < 8 * EPS_OFF and d_in == (1.0, 0.0, 0.0)                # direction is handed back exited
    # the squeezing portal (x -> x/4, x in [17.5,22.5] at y = 5): 17.5 outside, 20 to cross 5 at quarter speed, 2.5 left
    sq, _ = osc.trace_path_unknown(40.0, [0.0, 5.0, 0.0], [1.0, 0.0, 0.0])
    assert abs(sq[0] - 25.0) < 8 * EPS_OFF


def test_path_none_without_material(oracle_lib):
    """trace_path_unknown is None when no entity contains the start point (universe/mod.rs:280): 3d_fresnel's universe is a
    single sphere entity plus the Void; every point is inside the Void, so build the None case from a scene without one."""
    from oracle.scene_loader import OracleScene
    text = open(os.path.join(SCENES, "3d_fresnel.json")).read()
    osc = oracle_scene("3d_fresnel.json")
    assert osc.trace_path_unknown(1.0, [0.0, 0.0, 0.0], [1.0, 0.0, 0.0]) is not None
    import json
    js = json.loads(text)
    ents = js["Universe3"]["entities"]
    js["Universe3"]["entities"] = [e for e in ents if not (isinstance(e, dict) and any(k.startswith("Void3") for k in e))]
    assert len(js["Universe3"]["entities"]) < len(ents)
    from oracle.scene_loader import default_texture_loader
    o2 = OracleScene(json.dumps(js), default_texture_loader([ROOT]))
    assert o2.trace_path_unknown(1.0, [0.0, 0.0, 0.0], [1.0, 0.0, 0.0]) is None


def test_camera_kind_is_kept_by_both_loaders(oracle_lib):
    from euclider_amd import Parser, _capi
    for name, kind in [("3d_room.json", None), ("4d_room.json", _capi.EU_CAMERA_FREE_4)]:
        env = Parser().parse_file(os.path.join(SCENES, name))
        osc = oracle_scene(name)
        assert env.camera.kind == osc.camera_kind
        if kind is not None:
            assert env.camera.kind == kind
        env.close()
    text = open(os.path.join(SCENES, "3d_room.json")).read()
    for ctor, kind in [("PitchYawCamera3", _capi.EU_CAMERA_PITCH_YAW_3), ("FreeCamera3", _capi.EU_CAMERA_FREE_3)]:
        t = text.replace("PitchYawCamera3", ctor).replace("FreeCamera3", ctor)
        env = Parser().parse(t)
        assert env.camera.kind == kind
        env.close()


def _vec(c, f, D):
    return list(getattr(c, f))[:D]


@pytest.mark.parametrize("scene,kind", [("3d_room.json", 0), ("3d_room.json", 1), ("4d_room.json", 2)])
def test_rotation_maths_match_oracle(oracle_lib, scene, kind):
    """No translation (delta_time 0 or no movement key) => eu_camera_update needs no renderer.  Bit-for-bit equality of
    the pose after every one of 300 chained updates."""
    from euclider_amd import Parser, SimulationContext
    env = Parser().parse_file(os.path.join(SCENES, scene))
    osc = oracle_scene(scene)
    env.camera.kind = kind
    ocam = osc.camera()
    D = env.dim
    rng = random.Random(5 + kind)
    for it in range(300):
        dm = (rng.randint(-60, 60), rng.randint(-60, 60)) if rng.random() < 0.9 else (0, 0)
        keys = []
        if D == 3:
            keys = [k for k in ("Q", "E") if rng.random() < 0.3]
        else:
            keys = [rng.choice(["C", "M"])] + rng.sample(["I", "O", "K", "L"], 2)
            if rng.random() < 0.1:
                keys.append(rng.choice(["I", "O", "K", "L"]))       # three axes: no rotation
        dt_ms = rng.choice([0, 7, 16, 33]) if D == 3 else rng.choice([7, 16, 33])
        if D == 4:
            keys = [k for k in keys]                                  # no movement keys: W/S/A/D/LShift/LControl/Q/E
        env.update(dt_ms / 1000.0, SimulationContext(pressed_keys=keys, delta_mouse=dm))
        rc = osc.camera_update(ocam, dt_ms, keys, dm, kind=kind)
        assert rc == 0
        for f in ("location", "forward", "up", "left"):
            assert _vec(env.camera, f, D) == _vec(ocam, f, D), (it, f)
    # the pose is still orthonormal
    fw, up = _vec(env.camera, "forward", D), _vec(env.camera, "up", D)
    assert abs(sum(a * a for a in fw) - 1.0) < 1e-9 and abs(sum(a * b for a, b in zip(fw, up))) < 1e-6
    env.close()


def test_pitch_snaps_at_the_poles(oracle_lib):
    """PitchYawCamera3 clamps to straight up / down (d3/entity/camera.rs:119-130); FreeCamera3 does not."""
    from euclider_amd import Parser, SimulationContext
    env = Parser().parse_file(os.path.join(SCENES, "3d_room.json"))
    env.camera.kind = 0
    env.update(0.0, SimulationContext(delta_mouse=(0, -400)))         # pitch by +4 rad > pi/2
    assert _vec(env.camera, "forward", 3) == [0.0, 0.0, 1.0]
    env.update(0.0, SimulationContext(delta_mouse=(0, 1000)))
    assert _vec(env.camera, "forward", 3) == [0.0, 0.0, -1.0]
    env.close()


def test_moving_without_a_renderer_fails_loudly():
    from euclider_amd import Parser, _capi
    env = Parser().parse_file(os.path.join(SCENES, "3d_room.json"))
    inp = _capi.Input(_capi.KEYS["W"], 0, 0, 0, 16, 0.0, 0.0)
    before = _vec(env.camera, "location", 3)
    rc = _capi.lib().eu_camera_update(None, C.byref(env.camera), C.byref(inp))
    assert rc == _capi.EU_ERR_NO_DEVICE
    assert _vec(env.camera, "location", 3) == before
    env.close()
