"""The C-ABI library loads, exports every symbol include/euclider_amd.h declares, and has no CPU fallback."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "euclider_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(eu_[a-z0-9_]+)\s*\(", text)) - {"eu_texture_loader"})


def test_library_exports_every_declared_symbol():
    from euclider_amd import _capi
    L = _capi.lib()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "libeuclider_amd.so does not export %s" % n
        assert n in _capi.SYMBOLS, "euclider_amd/_capi.py does not bind %s" % n
    assert set(_capi.SYMBOLS) == set(names)


def test_struct_layouts_match_header():
    from euclider_amd import _capi
    assert C.sizeof(_capi.Camera) == 16 + 4 * 32
    assert C.sizeof(_capi.Frame) == 40
    assert C.sizeof(_capi.Stats) == 32
    assert C.sizeof(_capi.SceneInfo) == 44


def test_no_cpu_fallback_without_gpu():
    from euclider_amd import EuError, Parser, _capi
    if _capi.lib().eu_device_count() > 0:
        pytest.skip("a GPU is present")
    env = Parser().parse_file(os.path.join(ROOT, "scenes", "3d_fresnel.json"))
    with pytest.raises(EuError) as e:
        env.render((8, 8))
    assert e.value.code == _capi.EU_ERR_NO_DEVICE
    env.close()


def test_product_does_not_reference_oracle():
    """The shipped path must not import, include or link anything under oracle/."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "euclider_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hpp", ".cpp", ".hip", "Makefile")):
                txt = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"(import|from)\s+oracle|oracle/|eo_oracle|libeo_", txt):
                    bad.append(f)
    assert bad == []
