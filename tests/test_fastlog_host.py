"""csrc/mfpa_fastlog.h (the float64 logarithm of the pickers' pre-processing kernels) compiled for the HOST with gcc -- the very header
the device kernels include -- and measured against the x87 80-bit logl and against numpy's own log (the reference's arithmetic,
afp/audfprint/peak_extractor.py:276, afp/dejavu/fingerprint.py:78)."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r"""
#define MFPA_LOG_HOST
#include "mfpa_fastlog.h"
#include <stdint.h>
void flog(const double* x, double* y, long n) { for (long i = 0; i < n; ++i) y[i] = mfpa_log(x[i]); }
/* worst error in ulps against long double logl over the n arguments, and how many results are not the correctly rounded one */
double worst_ulp(const double* x, long n, long* not_cr) {
  double worst = 0; long bad = 0;
  for (long i = 0; i < n; ++i) {
    const double g = mfpa_log(x[i]);
    const long double w = logl((long double)x[i]);
    const double wd = (double)w;
    if (g != wd) ++bad;
    if (wd == 0) continue;
    int e; frexp(wd, &e);
    const double err = (double)(fabsl((long double)g - w) / ldexpl(1.0L, e - 53));
    if (err > worst) worst = err;
  }
  *not_cr = bad;
  return worst;
}
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    d = tmp_path_factory.mktemp("fastlog")
    c = d / "fastlog_host.c"
    c.write_text(SRC)
    so = d / "libfastlog_host.so"
    subprocess.run([gcc, "-O2", "-mfma", "-shared", "-fPIC", "-I", os.path.join(ROOT, "musicfpaugment_amd", "csrc"), "-o", str(so), str(c), "-lm"],
                   check=True)
    h = ctypes.CDLL(str(so))
    h.worst_ulp.restype = ctypes.c_double
    return h


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def test_fast_log_is_within_0_55_ulp_and_matches_numpy_like_glibc_does(lib):
    rng = np.random.default_rng(1)
    n = 2_000_000
    sets = {"pipeline range (1e-6, 1]": np.exp(-13.9 * rng.random(n)),
            "near 1": 1.0 + (rng.random(n) - 0.5) * 0.25,
            "uniform (0, 1)": np.maximum(rng.random(n), 1e-300),
            "every binade": np.ldexp(0.5 + 0.5 * rng.random(n), rng.integers(-1021, 1024, n))}
    for name, x in sets.items():
        x = np.ascontiguousarray(x)
        bad = ctypes.c_long(0)
        worst = lib.worst_ulp(_ptr(x), ctypes.c_long(n), ctypes.byref(bad))
        assert worst < 0.55, (name, worst)
        assert bad.value < (0.01 if name == "near 1" else 0.004) * n, (name, bad.value)   # >= 99.6 % correctly rounded (99 % in the zone around 1)
        y = np.empty_like(x)
        lib.flog(_ptr(x), _ptr(y), ctypes.c_long(n))
        ref = np.log(x)
        d = y != ref
        # numpy's own log is not correctly rounded either (0.03 % of the pipeline's range, ~2 % of the arguments within 12 % of 1, where
        # this function returns the correctly rounded value for 99.4 %)
        assert d.mean() < (3e-2 if name == "near 1" else 1e-3 if name.startswith("pipeline") else 8e-3), (name, d.mean())
        if d.any():
            assert np.max(np.abs(y[d] - ref[d]) / np.spacing(np.abs(ref[d]))) <= 1.0
    # exact and special values
    x = np.array([1.0, 2.0, 0.5, 1e-6, np.inf, 0.0, -1.0, np.nan, 5e-324, 2.2250738585072014e-308])
    y = np.empty_like(x)
    lib.flog(_ptr(x), _ptr(y), ctypes.c_long(x.size))
    with np.errstate(all="ignore"):
        ref = np.log(x)
    assert y[0] == 0.0 and np.array_equal(y[:4], ref[:4])
    assert y[4] == np.inf and y[5] == -np.inf and np.isnan(y[6]) and np.isnan(y[7]) and np.array_equal(y[8:], ref[8:])


def test_the_table_in_the_header_is_what_the_generator_prints():
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_fastlog_table.py")], capture_output=True, text=True, check=True).stdout
    hdr = open(os.path.join(ROOT, "musicfpaugment_amd", "csrc", "mfpa_fastlog.h")).read()
    assert out.strip() in hdr
