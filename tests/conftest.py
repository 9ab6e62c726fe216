import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "interpreter_only: a GPU test that exercises an ahead-of-time kernel variant by name; not repeated on the specialised kernels")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import scene_loader
    scene_loader.build()
    return scene_loader.lib()


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


# Every GPU test runs twice: on the ahead-of-time kernels that interpret the flat scene ("interp") and on kernels specialised for the
# test's scene, compiled with hiprtc when its renderer is created ("jit": eu_renderer_opts.specialize = EU_SPECIALIZE_SYNC).  The
# specialised pass costs a compilation per distinct scene (cached by content), so the random-scene tests take every fourth seed there.
def pytest_generate_tests(metafunc):
    if metafunc.definition.get_closest_marker("gpu") is not None and "kernel_path" in metafunc.fixturenames:
        modes = ["interp"] if metafunc.definition.get_closest_marker("interpreter_only") is not None else ["interp", "jit"]
        metafunc.parametrize("kernel_path", modes, indirect=True, scope="function")


@pytest.fixture(autouse=True)
def kernel_path(request):
    mode = getattr(request, "param", None)
    if mode is None:
        yield None
        return
    from euclider_amd import environment
    if mode == "jit":
        callspec = getattr(request.node, "callspec", None)
        seed = callspec.params.get("seed") if callspec is not None else None
        if isinstance(seed, int) and seed % 4 != 0:
            pytest.skip("specialised pass: every fourth random scene")
    saved = dict(environment.DEFAULT_RENDERER_OPTS)
    environment.DEFAULT_RENDERER_OPTS["specialize"] = "sync" if mode == "jit" else "off"
    if mode == "jit":
        # a failed compilation falls back to the interpreter kernels BY DESIGN: without this check every test of the specialised
        # pass would then pass vacuously.  (No scene is too large to specialise: jit.hpp's budgets.)
        def must_be_specialised(env, info):
            if not info["active"]:
                raise AssertionError("specialised pass, but this renderer runs the interpreter kernels: %r" % (info,))
        environment.DEFAULT_RENDERER_OPTS["on_specialize"] = must_be_specialised
    try:
        yield mode
    finally:
        environment.DEFAULT_RENDERER_OPTS.clear()
        environment.DEFAULT_RENDERER_OPTS.update(saved)
