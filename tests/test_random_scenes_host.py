"""Random scenes (tests/random_scenes.py): both loaders accept them and agree on what they built (no GPU needed)."""
import os

import numpy as np
import pytest

from random_scenes import random_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", range(0, 60))
def test_both_loaders_accept_random_scene(oracle_lib, seed):
    from euclider_amd import Parser
    from oracle.scene_loader import OracleScene, default_texture_loader
    text, dim = random_scene(seed)
    env = Parser(texture_dirs=[ROOT]).parse(text)
    osc = OracleScene(text, default_texture_loader([ROOT]))
    assert env.dim == dim == osc.dim
    assert env.camera.kind == osc.camera_kind
    assert list(env.camera.location)[:dim] == list(osc.camera().location)[:dim]
    assert env.info.n_entities >= 1
    # the oracle renders it (tiny frame) without tripping its own consistency checks
    rgb, _, st = osc.render(8, 6, max_depth=3, threads=2)
    assert rgb.shape == (6, 8, 3) and st["rays"] <= 8 * 6 * (2 ** 3)
    env.close()
