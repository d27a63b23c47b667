"""CPU-only checks of the drop-in boundary: the library loads, exports every
symbol include/shpair.h and include/shstep.h declare, its stateless host helpers agree with scipy,
and without a GPU it refuses to create a context (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest
from scipy.special import sph_harm_y

from shpair import capi, shapes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    names = set()
    for h in ("shpair.h", "shstep.h", "shhalo.h"):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(sh(?:pair|step|halo)_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.library_path())
    names = header_symbols()
    assert len(names) >= 34
    for n in names:
        assert hasattr(lib, n), f"libshpair.so does not export {n}"
    assert sorted(capi.SYMBOLS) == names  # the ctypes binding covers exactly the header


def test_version_and_error_strings():
    lib = capi.load_library()
    assert b"gfx950" in lib.shpair_version()
    for code in (0, -1, -2, -3, -4, -5, -6):
        assert lib.shpair_strerror(code)
    assert b"no CPU fallback" in lib.shpair_strerror(-2)


def test_host_helpers_match_scipy():
    lmax = 7
    anm = shapes.random_shape(lmax, 4, amp=0.3).reshape(-1, 2)
    rng = np.random.default_rng(0)
    for _ in range(50):
        th, ph = np.arccos(rng.uniform(-1, 1)), rng.uniform(0, 2 * np.pi)
        u = [np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)]
        ref = sum(((1 if m == 0 else 2) * (anm[n * (n + 1) // 2 + m, 0] + 1j * anm[n * (n + 1) // 2 + m, 1]) *
                   sph_harm_y(n, m, th, ph)).real for n in range(lmax + 1) for m in range(n + 1))
        assert abs(capi.shape_radius(lmax, anm.ravel(), u) - ref) < 1e-13


def test_default_rmax_agrees_with_oracle(oracle):
    for lmax in (0, 4, 6, 12):
        a = shapes.random_shape(lmax, 9)
        assert abs(capi.shape_default_rmax(lmax, a) - oracle.shape_rmax(lmax, a)) < 1e-14


def test_bad_arguments_to_stateless_helpers():
    lib = capi.load_library()
    r = ctypes.c_double()
    assert lib.shpair_shape_radius(-1, None, None, ctypes.byref(r)) == -1
    assert lib.shpair_shape_default_rmax(99, None, ctypes.byref(r)) == -1


def test_no_gpu_means_no_context(gpu_available):
    if gpu_available:
        pytest.skip("GPU present: covered by the -m gpu tests")
    with pytest.raises(capi.ShPairError) as e:
        capi.ShPair(0)
    assert e.value.code == -2  # SHPAIR_ENODEV: fails loudly, never computes on the CPU


def test_mass_props_helper_agrees_with_oracle(oracle):
    """Stateless host helper of include/shstep.h (no device needed) vs the oracle's own quadrature."""
    for lmax, seed in ((0, 1), (4, 2), (6, 3), (12, 4)):
        a = shapes.random_shape(lmax, seed, amp=0.3)
        got = capi.shape_mass_props(lmax, a)
        ref = oracle.mass_props(lmax, a)
        assert np.abs(got - ref).max() < 1e-13 * max(1.0, np.abs(ref).max())
    out = np.zeros(10)
    assert capi.load_library().shstep_shape_mass_props(-1, out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                                                      out.ctypes.data_as(ctypes.POINTER(ctypes.c_double))) == -1
