"""Synthetic spherical-harmonic shapes for the BASELINE.json configs (setup only).

Coefficient storage follows docs/SPEC.md §1: a_nm for m >= 0, n-major,
anm[2k] = Re, anm[2k+1] = Im with k = n(n+1)/2 + m; scipy `sph_harm_y`
normalisation and phase.  The reference's shape-file reader is ABSENT FROM
MOUNT, so shapes are generated, not read.
"""
import numpy as np


def nterms(lmax):
    return (lmax + 1) * (lmax + 2) // 2


def sh_radius_np(lmax, anm, u):
    """r(u) for unit vectors u[...,3]; plain normalised recurrence, vectorised (setup utility)."""
    anm = np.asarray(anm, dtype=np.float64).reshape(-1, 2)
    u = np.asarray(u, dtype=np.float64)
    x, y, z = u[..., 0], u[..., 1], u[..., 2]
    r = np.zeros_like(z)
    cm, sm = np.ones_like(z), np.zeros_like(z)
    pmm = np.sqrt(1.0 / (4.0 * np.pi))
    for m in range(lmax + 1):
        if m > 0:
            pmm = -pmm * np.sqrt((2.0 * m + 1.0) / (2.0 * m))
        fac = 1.0 if m == 0 else 2.0
        p2 = np.zeros_like(z)
        p1 = np.full_like(z, pmm)
        k = m * (m + 1) // 2 + m
        wr, wi = anm[k, 0] * p1, anm[k, 1] * p1
        for n in range(m + 1, lmax + 1):
            a = np.sqrt((4.0 * n * n - 1.0) / (n * n - m * m))
            b = 0.0 if n - m < 2 else np.sqrt(((2.0 * n + 1.0) * (n + m - 1.0) * (n - m - 1.0)) /
                                              ((n - m) * (n + m) * (2.0 * n - 3.0)))
            p = a * z * p1 - b * p2
            k = n * (n + 1) // 2 + m
            wr = wr + anm[k, 0] * p
            wi = wi + anm[k, 1] * p
            p2, p1 = p1, p
        r = r + fac * (wr * cm - wi * sm)
        cm, sm = cm * x - sm * y, cm * y + sm * x
    return r


def _sphere_grid(nt):
    t, w = np.polynomial.legendre.leggauss(nt)
    ph = 2.0 * np.pi * np.arange(2 * nt) / (2 * nt)
    ct, phg = np.meshgrid(t, ph, indexing="ij")
    st = np.sqrt(1.0 - ct * ct)
    u = np.stack([st * np.cos(phg), st * np.sin(phg), ct], axis=-1)
    wg = np.repeat(w[:, None], 2 * nt, axis=1) * (2.0 * np.pi / (2 * nt))
    return u, wg, np.arccos(ct), phg


def sphere(radius=1.0, lmax=0):
    anm = np.zeros((nterms(lmax), 2))
    anm[0, 0] = radius * np.sqrt(4.0 * np.pi)
    return anm.ravel()


def project(fun, lmax, nt=None):
    """a_nm = integral fun(u) conj(Y_nm) dOmega by Gauss x trapezoid quadrature."""
    from scipy.special import sph_harm_y
    nt = nt or 4 * (lmax + 2)
    u, wg, th, ph = _sphere_grid(nt)
    f = fun(u)
    anm = np.zeros((nterms(lmax), 2))
    for n in range(lmax + 1):
        for m in range(n + 1):
            a = np.sum(f * np.conj(sph_harm_y(n, m, th, ph)) * wg)
            anm[n * (n + 1) // 2 + m] = (a.real, a.imag if m > 0 else 0.0)
    return anm.ravel()


def ellipsoid(a=1.0, b=0.8, c=0.6, lmax=4):
    """Band-limited (order lmax) projection of the ellipsoid radius function (config 1)."""
    def rad(u):
        return 1.0 / np.sqrt((u[..., 0] / a) ** 2 + (u[..., 1] / b) ** 2 + (u[..., 2] / c) ** 2)
    return project(rad, lmax)


def random_shape(lmax, seed, amp=0.1, rmin=0.5):
    """Unit mean radius; a_nm (n>=1) ~ N(0, amp/(n+1)^2), shrunk until r > rmin everywhere (configs 2-5)."""
    rng = np.random.default_rng(seed)
    anm = np.zeros((nterms(lmax), 2))
    anm[0, 0] = np.sqrt(4.0 * np.pi)
    for n in range(1, lmax + 1):
        sd = amp / (n + 1) ** 2 * np.sqrt(4.0 * np.pi)
        for m in range(n + 1):
            anm[n * (n + 1) // 2 + m, 0] = rng.normal(0.0, sd)
            if m > 0:
                anm[n * (n + 1) // 2 + m, 1] = rng.normal(0.0, sd)
    u, _, _, _ = _sphere_grid(6 * (lmax + 1) + 2)
    base = anm.copy()
    scale = 1.0
    for _ in range(40):
        anm[1:] = base[1:] * scale
        if sh_radius_np(lmax, anm.ravel(), u).min() > rmin:
            break
        scale *= 0.8
    return anm.ravel()
