"""Spherical-harmonic shapes: synthetic ones for the BASELINE.json configs, the text file format the PairSH
adapter reads, and a least-squares fit of an expansion to surface points (setup only).

Coefficient storage follows docs/SPEC.md §1: a_nm for m >= 0, n-major,
anm[2k] = Re, anm[2k+1] = Im with k = n(n+1)/2 + m; scipy `sph_harm_y`
normalisation and phase.  The reference's shape-file reader is ABSENT FROM
MOUNT: the file format below is this repo's own.
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


# ---- shape files and fitting (the data format on the input side of the path) --------------------------------
# Text format read by the PairSH adapter (lammps/pair_sh.cpp::load_shapes): first line `lmax`, then one line
# `n m Re(a_nm) Im(a_nm)` per stored coefficient, m >= 0.  The reference's own format is unknown (its reader
# is absent from the mount), so this is the repo's format, not a reimplementation of theirs.

def write_shape_file(path, lmax, anm):
    a = np.asarray(anm, dtype=np.float64).reshape(-1, 2)
    if a.shape[0] != nterms(lmax):
        raise ValueError(f"anm has {a.shape[0]} terms, lmax {lmax} needs {nterms(lmax)}")
    with open(path, "w") as fp:
        fp.write(f"{lmax}\n")
        for n in range(lmax + 1):
            for m in range(n + 1):
                k = n * (n + 1) // 2 + m
                fp.write(f"{n} {m} {float(a[k, 0])!r} {float(a[k, 1])!r}\n")


def read_shape_file(path):
    """Returns (lmax, anm).  Same grammar as PairSH::load_shapes (lammps/pair_sh.cpp): data lines `n m Re Im`, an optional
    first line with lmax alone (else lmax = the largest n), `#` comments; coefficients that the file does not list are
    zero; m < 0 is accepted when a_{n,-m} = (-1)^m conj(a_{n,m}) (a real radius), and fills +m when only -m is listed."""
    with open(path) as fp:
        lines = [ln.split("#")[0].strip() for ln in fp]
    lines = [ln for ln in lines if ln]
    if not lines:
        raise ValueError(f"{path}: empty shape file")
    lmax = None
    if len(lines[0].split()) == 1:
        lmax = int(lines[0])
        lines = lines[1:]
    ent = []
    for ln in lines:
        n, m, re, im = ln.split()
        n, m = int(n), int(m)
        if not (-n <= m <= n) or n < 0 or (lmax is not None and n > lmax):
            raise ValueError(f"{path}: (n, m) = ({n}, {m}) out of range")
        ent.append((n, m, float(re), float(im)))
    if lmax is None:
        if not ent:
            raise ValueError(f"{path}: empty shape file")
        lmax = max(e[0] for e in ent)
    a = np.zeros((nterms(lmax), 2))
    have = np.zeros(nterms(lmax), dtype=bool)
    amax = max([max(abs(e[2]), abs(e[3])) for e in ent] + [0.0])
    for n, m, re, im in ent:
        if m < 0:
            continue
        if m == 0 and abs(im) > 1e-9 * amax:
            raise ValueError(f"{path}: a_{n}0 must be real")
        k = n * (n + 1) // 2 + m
        a[k] = (re, 0.0 if m == 0 else im)
        have[k] = True
    for n, m, re, im in ent:
        if m >= 0:
            continue
        k = n * (n + 1) // 2 - m
        sg = -1.0 if (-m) & 1 else 1.0
        v = (sg * re, -sg * im)
        if have[k]:
            if abs(a[k, 0] - v[0]) > 1e-9 * amax or abs(a[k, 1] - v[1]) > 1e-9 * amax:
                raise ValueError(f"{path}: not a real radius: a_({n},{m}) != (-1)^m conj(a_({n},{-m}))")
        else:
            a[k] = v
            have[k] = True
    return lmax, a.ravel()


def basis_matrix(lmax, u):
    """Real design matrix B with r(u_k) = B[k] @ anm for the SPEC storage (column 2k: Re a, 2k+1: Im a)."""
    u = np.asarray(u, dtype=np.float64)
    nt = nterms(lmax)
    B = np.zeros((u.shape[0], 2 * nt))
    e = np.zeros(2 * nt)
    for c in range(2 * nt):
        if c % 2 == 1 and _m_of(c // 2) == 0:
            continue          # Im a_n0 does not enter r
        e[:] = 0.0
        e[c] = 1.0
        B[:, c] = sh_radius_np(lmax, e, u)
    return B


def _m_of(k):
    n = int((np.sqrt(8 * k + 1) - 1) // 2)
    return k - n * (n + 1) // 2


def fit_points(points, lmax, centre=None, ridge=0.0):
    """Least-squares SH expansion of a star-shaped surface sampled by `points` (N x 3, e.g. the vertices of a
    scanned grain): r(u_k) = |p_k - centre| in direction u_k.  centre defaults to the mean of the points.
    ridge > 0 adds n^2 (n+1)^2 smoothing (useful when N is small for the order).
    Returns (anm, centre, rms residual)."""
    p = np.asarray(points, dtype=np.float64)
    c = p.mean(axis=0) if centre is None else np.asarray(centre, dtype=np.float64)
    d = p - c
    r = np.linalg.norm(d, axis=1)
    if not np.all(r > 0):
        raise ValueError("a point coincides with the centre")
    B = basis_matrix(lmax, d / r[:, None])
    use = np.array([not (cidx % 2 == 1 and _m_of(cidx // 2) == 0) for cidx in range(B.shape[1])])
    A = B[:, use]
    if ridge > 0.0:
        pen = []
        for cidx in np.flatnonzero(use):
            k = cidx // 2
            n = int((np.sqrt(8 * k + 1) - 1) // 2)
            pen.append(np.sqrt(ridge) * n * (n + 1))
        A = np.vstack([A, np.diag(pen)])
        rhs = np.concatenate([r, np.zeros(len(pen))])
    else:
        rhs = r
    sol, *_ = np.linalg.lstsq(A, rhs, rcond=None)
    anm = np.zeros(B.shape[1])
    anm[use] = sol
    res = B @ anm - r
    return anm, c, float(np.sqrt(np.mean(res * res)))
