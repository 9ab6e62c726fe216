"""Scene-specialised kernels, host side (no GPU): the generator (csrc/jit.cpp) on every shipped scene and both precisions, a real hiprtc
compilation for gfx950 (hiprtc cross-compiles without a device), the code-object cache, and the size limit beyond which a
renderer keeps the interpreter kernels."""
import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCENES = sorted(glob.glob(os.path.join(ROOT, "scenes", "*.json")))


@pytest.mark.parametrize("low_precision", [False, True])
def test_generator_on_every_shipped_scene(low_precision):
    from euclider_amd import Parser
    keys = set()
    for path in SCENES:
        env = Parser(low_precision=low_precision).parse_file(path)
        src, key = env.jit_source()
        again, key2 = env.jit_source()
        assert src == again and key == key2 and len(key) == 32          # deterministic: the cache is keyed by content
        keys.add(key)
        for name in ("eu_jit_intersect0", "eu_jit_fshade0", "eu_jit_fshade", "struct EuJit", "trace_closest", "material_at"):      # (the default module: the fused pipeline)
            assert name in src, (path, name)
        # one inside-test function per entity root, one trace_closest block per surfaced entity
        assert src.count("/* entity ") >= 1
        env.close()
    assert len(keys) == len(SCENES)


def test_linear_space_expressions_become_arithmetic():
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_hallways.json"))
    src, _ = env.jit_source()
    env.close()
    assert "ctx[0]" in src and "0x1p+2" in src          # `x * 4` (material.rs:99-111): the context component times the literal 4
    assert "eval_rpn" not in src                         # no RPN interpreter in the specialised code


def test_hiprtc_compile_and_cache(tmp_path):
    import json
    from euclider_amd import Parser
    # a scene of its own (the shipped scenes' kernels may already sit in euclider_amd/jit_cache, which is consulted first)
    text = open(os.path.join(ROOT, "scenes", "3d_fresnel.json")).read().replace("1.458", "1.4625")
    assert text != open(os.path.join(ROOT, "scenes", "3d_fresnel.json")).read()
    json.loads(text)
    env = Parser(texture_dirs=[ROOT]).parse(text)
    cache = str(tmp_path / "cache")
    first = env.jit_precompile(cache)
    assert not first["from_cache"] and first["compile_ms"] > 0
    files = os.listdir(cache)
    assert files == [first["key"] + ".hsaco"]
    with open(os.path.join(cache, files[0]), "rb") as f:
        assert f.read(4) == b"\x7fELF"
    env.close()
    env = Parser(texture_dirs=[ROOT]).parse(text)
    second = env.jit_precompile(cache)
    assert second["from_cache"] and second["key"] == first["key"]
    env.close()


def test_large_scenes_are_not_specialised(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from random_scenes import random_scene
    from euclider_amd import Parser
    from euclider_amd.environment import EuError
    text, _ = random_scene(501, n_entities=150)
    env = Parser(texture_dirs=[ROOT]).parse(text)
    assert env.info.n_entities > 48
    with pytest.raises(EuError) as ei:
        env.jit_precompile(str(tmp_path))
    assert ei.value.code == -5 and "too large" in str(ei.value)
    env.close()
