"""Elementary functions: the oracle's eo_math.h against glibc (<= 1 ulp), and the product's eu_math.h
against the oracle's, bit for bit, both compiled for the host (the device side is checked in
tests/test_gpu_parity.py::test_device_math_matches_oracle)."""
import ctypes as C
import math
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ulps(a, b):
    ia = a.view(np.int64).copy()
    ib = b.view(np.int64).copy()
    ia[ia < 0] = np.iinfo(np.int64).min - ia[ia < 0]
    ib[ib < 0] = np.iinfo(np.int64).min - ib[ib < 0]
    d = np.abs(ia - ib)
    d[np.isnan(a) & np.isnan(b)] = 0
    return d


def run(L, fn, x, y=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float64)
    out = np.zeros_like(x)
    L.eo_test_math(fn, x.ctypes.data, y.ctypes.data, out.ctypes.data, len(x))
    return out


def test_oracle_math_within_one_ulp_of_libm(oracle_lib):
    rng = np.random.default_rng(3)
    n = 400000
    u = rng.uniform(-1, 1, n)
    a = np.concatenate([rng.uniform(-10, 10, n // 2), rng.uniform(-1e5, 1e5, n // 2)])
    y, z = rng.uniform(-5, 5, n), rng.uniform(-5, 5, n)
    assert ulps(run(oracle_lib, 0, u), np.arccos(u)).max() <= 1
    assert ulps(run(oracle_lib, 1, u), np.arcsin(u)).max() <= 1
    assert ulps(run(oracle_lib, 2, a), np.sin(a)).max() <= 1
    assert ulps(run(oracle_lib, 3, a), np.cos(a)).max() <= 1
    assert ulps(run(oracle_lib, 4, a), np.tan(a)).max() <= 1
    # numpy's arctan2 is its own SIMD routine; compare with glibc's through math.atan2
    ref = np.array([math.atan2(p, q) for p, q in zip(y[:100000], z[:100000])])
    assert ulps(run(oracle_lib, 5, y[:100000], z[:100000]), ref).max() <= 1


def test_oracle_math_special_values(oracle_lib):
    x = np.array([1.0, -1.0, 0.0, 1.5, -1.5, np.nan, 1e-300, 0.5, -0.5])
    assert np.array_equal(np.isnan(run(oracle_lib, 0, x)), np.isnan(np.arccos(x)))
    assert run(oracle_lib, 0, np.array([1.0]))[0] == 0.0
    assert run(oracle_lib, 0, np.array([-1.0]))[0] == math.pi
    assert run(oracle_lib, 5, np.array([0.0]), np.array([-1.0]))[0] == math.pi
    assert math.copysign(1, run(oracle_lib, 5, np.array([-0.0]), np.array([1.0]))[0]) == -1.0


def test_product_math_header_equals_oracle_math_on_host(tmp_path):
    src = tmp_path / "cmp.cpp"
    src.write_text(r'''
#include "%s/euclider_amd/csrc/eu_math.h"
extern "C" {
#include "%s/oracle/eo_math.h"
}
#include <cstdio>
#include <cstdlib>
#include <cstring>
int main() { srand(5); long bad = 0;
  for (int i = 0; i < 2000000; i++) {
    double x = (rand() / (double)RAND_MAX * 2 - 1) * (i %% 3 ? 1.0 : 50.0), y = (rand() / (double)RAND_MAX * 2 - 1) * 5;
    double a[7] = {eu_acos(x), eu_asin(x), eu_sin(x), eu_cos(x), eu_tan(x), eu_atan2(x, y), eu_atan(x)};
    double b[7] = {eo_acos(x), eo_asin(x), eo_sin(x), eo_cos(x), eo_tan(x), eo_atan2(x, y), eo_atan(x)};
    if (memcmp(a, b, sizeof a)) bad++; }
  printf("%%ld\n", bad); return bad != 0; }
''' % (ROOT, ROOT))
    exe = tmp_path / "cmp"
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", str(src), "-o", str(exe)])
    assert subprocess.check_output([str(exe)]).strip() == b"0"
