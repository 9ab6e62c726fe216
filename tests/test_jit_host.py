"""Scene-specialised kernels, host side (no GPU): the generator (csrc/jit.cpp) on every shipped scene and both precisions, a real hiprtc
compilation for gfx950 (hiprtc cross-compiles without a device), the code-object cache, and the budgets beyond which the rest of a
scene is traced from the flat scene inside the specialised kernels."""
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


def test_large_scenes_get_mixed_kernels():
    """Beyond the generator's budgets (jit.hpp: 256 shape operations, 48 surfaces) a scene's remaining entities and surfaces are traced and
    shaded from the flat scene INSIDE the specialised kernels, at their place in the entity order (rounds 2-3 left such a scene to the
    interpreter kernels altogether)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import re
    from random_scenes import random_scene
    from euclider_amd import Parser
    text, _ = random_scene(501, n_entities=150)
    env = Parser(texture_dirs=[ROOT]).parse(text)
    assert env.info.n_entities > 48 and env.info.n_shape_ops > 256
    src, key = env.jit_source()
    env.close()
    runs = [(int(a), int(b)) for a, b in re.findall(r"interp_entities<\d>\(S, (\d+)u, (\d+)u,", src)]
    assert runs and all(a < b for a, b in runs) and runs == sorted(runs)
    own = [int(e) for e in re.findall(r"/\* entity (\d+): ops", src)]
    assert own and not any(a <= e < b for e in own for a, b in runs)          # every entity is traced once: by its own code or by a run
    assert "material_at_range<" in src and "::hit_normal<" in src and "surface_color<" in src
    assert src.count("static EU_DEV void surf_") == 48


def test_budgets_are_part_of_the_key_and_the_mixed_kernels_compile(tmp_path):
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_room.json"))
    whole, key = env.jit_source()
    assert "interp_entities" not in whole and "surface_color<" not in whole
    flags = "-DEU_JIT_OPS_BUDGET=3 -DEU_JIT_SURFACES_BUDGET=2"
    mixed, key2 = env.jit_source(flags)
    assert key2 != key
    assert "interp_entities<3>(S, 3u, 8u," in mixed and "material_at_range<3>(S, 3u, 8u, p)" in mixed and mixed.count("static EU_DEV void surf_") == 2
    info = env.jit_precompile(str(tmp_path), flags)
    assert info["key"] == key2 and not info["from_cache"]
    env.close()


def test_congruent_entities_get_one_body():
    """4d_cylinders: entities 1..8 are the same 29 operations on parameters a constant stride apart: one body in a loop over them (jit.cpp:
    find_runs), a quarter of the source of the straight-line form (-DEU_JIT_NO_RUNS, which rounds 2-3 emitted)."""
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "4d_cylinders.json"))
    src, key = env.jit_source()
    flat, key2 = env.jit_source("-DEU_JIT_NO_RUNS")
    env.close()
    assert key != key2
    assert "for (uint32_t ge = 0; ge < 8u; ge++) {   /* entities 1..8: congruent, ops 15..43 each */" in src
    assert "const uint32_t po = ge * 126u, bo = ge * 30u, oo = ge * 29u;" in src
    assert "ge++" not in flat and flat.count("/* entity ") == 9 and src.count("/* entity ") == 1
    assert len(src) * 3 < len(flat)


def test_guard_at_the_register_cliff(tmp_path):
    """jit_build reads the fused shade kernel's register count from the code object's metadata and, just above 128 (four waves per SIMD -> three),
    compiles the module again with launch bounds of four waves.  -DEU_JIT_CLIFF_LO=100 makes 3d_fresnel's kernel count as "just above"."""
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_fresnel.json"))
    plain = env.jit_precompile(str(tmp_path / "a"))
    forced = env.jit_precompile(str(tmp_path / "b"), "-DEU_JIT_CLIFF_LO=100")
    env.close()
    assert "four waves" not in plain["log"]
    assert "eu_jit_fshade: " in forced["log"] and "compiled for four waves per SIMD" in forced["log"], forced["log"]


def test_left_folds_of_congruent_operands_get_one_body():
    """4d_frame's one entity is Complement(box, Union(box, box, box, box)): the Union is a left fold of four congruent chains (ComposableShape::of,
    shape.rs:523-545) and gets ONE push and ONE merge in a loop over the operands from the second on (jit.cpp: fold_len, emit_fold)."""
    from euclider_amd import Parser
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "4d_frame.json"))
    src, key = env.jit_source()
    flat, key2 = env.jit_source("-DEU_JIT_NO_FOLDS")
    env.close()
    assert key != key2
    assert "for (uint32_t fj = 1; fj < 4u; fj++) {   /* operands 1..3 of the fold that starts at op 1 */" in src
    assert "fin_acc_1(S, q, fj, po)" in src and "static EU_DEV bool fin_1(" in src
    assert src.count("push_chain<4>(") == 3 and flat.count("push_chain<4>(") == 5          # the outer box, the fold's first operand, the loop body
    assert "fj++" not in flat
