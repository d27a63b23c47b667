"""Pins the oracle's SH math against library known answers.
The reference holds no tests or fixtures for this path (mount = README.md only),
so these known answers are what the oracle is pinned by ("parity unpinned")."""
import numpy as np
import pytest
from scipy.special import sph_harm_y

from shpair import shapes


def ref_radius(lmax, anm, th, ph):
    anm = np.asarray(anm).reshape(-1, 2)
    r = 0.0
    for n in range(lmax + 1):
        for m in range(n + 1):
            a = anm[n * (n + 1) // 2 + m, 0] + 1j * anm[n * (n + 1) // 2 + m, 1]
            r += ((1 if m == 0 else 2) * a * sph_harm_y(n, m, th, ph)).real
    return r


@pytest.mark.parametrize("lmax", [0, 1, 2, 4, 6, 9, 12, 16, 20])
def test_radius_matches_scipy_sph_harm(oracle, lmax):
    rng = np.random.default_rng(lmax)
    anm = rng.normal(size=(shapes.nterms(lmax), 2))
    for n in range(lmax + 1):
        anm[n * (n + 1) // 2, 1] = 0.0
    scale = np.abs(anm).sum()
    for _ in range(200):
        th, ph = np.arccos(rng.uniform(-1, 1)), rng.uniform(0, 2 * np.pi)
        u = [np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)]
        assert abs(oracle.sh_eval(lmax, anm.ravel(), u) - ref_radius(lmax, anm, th, ph)) < 1e-13 * scale * (lmax + 1)


@pytest.mark.parametrize("lmax", [0, 3, 6, 12])
def test_radius_at_poles_is_finite_and_exact(oracle, lmax):
    rng = np.random.default_rng(5)
    anm = rng.normal(size=(shapes.nterms(lmax), 2))
    for n in range(lmax + 1):
        anm[n * (n + 1) // 2, 1] = 0.0
    for z in (1.0, -1.0):
        th = 0.0 if z > 0 else np.pi
        r, g = oracle.sh_eval(lmax, anm.ravel(), [0.0, 0.0, z], grad=True)
        assert np.isfinite(r) and np.all(np.isfinite(g))
        assert abs(r - ref_radius(lmax, anm, th, 0.3)) < 1e-12 * np.abs(anm).sum()


@pytest.mark.parametrize("lmax", [1, 4, 6, 12])
def test_gradient_matches_finite_differences_on_the_sphere(oracle, lmax):
    rng = np.random.default_rng(100 + lmax)
    anm = shapes.random_shape(lmax, 3, amp=0.3)
    for _ in range(50):
        u = rng.normal(size=3)
        u /= np.linalg.norm(u)
        t = np.cross(u, rng.normal(size=3))
        t /= np.linalg.norm(t)
        r, g = oracle.sh_eval(lmax, anm, u, grad=True)
        h = 1e-5
        up, um = u + h * t, u - h * t
        fd = (oracle.sh_eval(lmax, anm, up / np.linalg.norm(up)) -
              oracle.sh_eval(lmax, anm, um / np.linalg.norm(um))) / (2 * h)
        assert abs(fd - t @ g) < 1e-7 * (1 + abs(fd))


@pytest.mark.parametrize("n", [1, 2, 3, 8, 10, 16, 32, 64, 128])
def test_gauss_legendre_matches_numpy(oracle, n):
    t, w = oracle.gauss_legendre(n)
    t2, w2 = np.polynomial.legendre.leggauss(n)
    # end weights are ill conditioned in the node position (dw/w ~ 2 dx/(1-x^2)); numpy's own
    # eigenvalue+Newton nodes carry ~1e-14 there at n = 128
    assert np.abs(t - t2).max() < 2e-15 and np.abs(w - w2).max() < (4e-15 if n <= 32 else 5e-14)
    assert abs(w.sum() - 2.0) < 1e-14


def test_default_rmax_bounds_the_shape(oracle):
    for lmax, seed in [(4, 1), (6, 2), (12, 3)]:
        anm = shapes.random_shape(lmax, seed, amp=0.2)
        rmax = oracle.shape_rmax(lmax, anm)
        rng = np.random.default_rng(seed)
        u = rng.normal(size=(20000, 3))
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        r = shapes.sh_radius_np(lmax, anm, u)
        assert r.max() < rmax and rmax < 1.02 * r.max()
    assert abs(oracle.shape_rmax(0, shapes.sphere(1.5)) - 1.01 * 1.5) < 1e-13


def test_numpy_setup_radius_agrees_with_oracle(oracle):
    anm = shapes.random_shape(8, 11)
    rng = np.random.default_rng(0)
    u = rng.normal(size=(100, 3))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    r = shapes.sh_radius_np(8, anm, u)
    ro = np.array([oracle.sh_eval(8, anm, v) for v in u])
    assert np.abs(r - ro).max() < 1e-13


def test_ellipsoid_projection_is_close_to_the_ellipsoid():
    anm = shapes.ellipsoid(1.0, 0.8, 0.6, lmax=8)
    r = shapes.sh_radius_np(8, anm, np.eye(3))
    assert np.allclose(r, [1.0, 0.8, 0.6], atol=0.02)


@pytest.mark.parametrize("lmax", [4, 12, 20])
def test_single_harmonics_against_mpmath_at_50_digits(oracle, lmax):
    """spot values of Y_nm (hence of the Legendre recurrence with its normalisation and
    Condon-Shortley phase) against mpmath at 50 digits, every (n, m) of the order, directions including near-polar
    ones; the library's host helper (shpair_shape_radius) is held to the same values."""
    import mpmath
    from shpair import capi
    mpmath.mp.dps = 50
    rng = np.random.default_rng(7 + lmax)
    dirs = [(np.arccos(rng.uniform(-1, 1)), rng.uniform(0, 2 * np.pi)) for _ in range(3)] + [(1e-3, 0.7), (np.pi - 2e-3, 4.1)]
    nt = shapes.nterms(lmax)
    worst = 0.0
    for n in range(lmax + 1):
        for m in range(n + 1):
            k = n * (n + 1) // 2 + m
            for part in ((0, 1) if m > 0 else (0,)):          # a_nm = 1, then a_nm = i
                anm = np.zeros((nt, 2))
                anm[k, part] = 1.0
                for th, ph in dirs:
                    y = mpmath.spherharm(n, m, mpmath.mpf(float(th)), mpmath.mpf(float(ph)))
                    a = mpmath.mpc(1, 0) if part == 0 else mpmath.mpc(0, 1)
                    ref = float(((1 if m == 0 else 2) * a * y).real)
                    u = [np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)]
                    worst = max(worst, abs(oracle.sh_eval(lmax, anm.ravel(), u) - ref),
                                abs(capi.shape_radius(lmax, anm.ravel(), u) - ref))
    # |Y_nm| <= sqrt((2n+1)/4pi) ~ 1.8 at n = 20; the direction itself is rounded to double first
    assert worst < 2e-13, worst
