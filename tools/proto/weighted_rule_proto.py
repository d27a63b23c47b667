"""Prototype (CPU): the candidate 'weighted' cap rule, stated exactly as a kernel could compute it, against
the sharp rule of docs/SPEC.md §2 — errors of the force integral S_n (vector) vs n_q on bed-like contacts.

  g~_kl = s - r_j  if s < R_j  else  s - R_j
  D_l   = |g~_{k,l+1} - g~_{k,l-1}| / 2                      (azimuth, periodic)
  D_k   = |g~_{k+1,l} - g~_{k,l}|   (k < n_q - 1),  |g~_{k,l} - g~_{k-1,l}|  (last ring)
  w_kl  = clip(1/2 - g~_kl / (D_k + D_l), 0, 1)              ([g~ < 0] if D_k + D_l = 0);  0 if s >= R_j
  S_n   = sum w_kl omega_kl A_kl
Evidence for DESIGN.md; not used by the product or the tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
sys.path.insert(0, ROOT)
from shpair import shapes, bed  # noqa: E402
from oracle import oracle as O  # noqa: E402  (prototype only)


def rotmat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def rule(lmax, a, R, qi, qj, d, nq, weighted):
    rho = np.linalg.norm(d)
    cosa = np.sqrt(rho * rho - R * R) / rho if rho * rho - R * R <= R * R else rho / (2 * R)
    c = d / rho
    sg = np.copysign(1.0, c[2])
    aa = -1.0 / (sg + c[2])
    bb = c[0] * c[1] * aa
    e1 = np.array([1 + sg * c[0] ** 2 * aa, sg * bb, -sg * c[0]])
    e2 = np.array([bb, sg + c[1] ** 2 * aa, -c[1]])
    t, w = np.polynomial.legendre.leggauss(nq)
    mu = 0.5 * (1 + cosa) + 0.5 * (1 - cosa) * t
    psi = 2 * np.pi * (np.arange(2 * nq) + 0.5) / (2 * nq)
    MU, PSI = np.meshgrid(mu, psi, indexing="ij")
    OM = np.repeat((0.5 * (1 - cosa) * w)[:, None], 2 * nq, axis=1) * (2 * np.pi / (2 * nq))
    SIG = np.sqrt(1 - MU ** 2)
    U = SIG[..., None] * (np.cos(PSI)[..., None] * e1 + np.sin(PSI)[..., None] * e2) + MU[..., None] * c
    Ri, Rj = rotmat(qi), rotmat(qj)
    Ui = U @ Ri
    ri = shapes.sh_radius_np(lmax, a, Ui)
    Q = (ri[..., None] * U - d) @ Rj
    s = np.linalg.norm(Q, axis=-1)
    rj = shapes.sh_radius_np(lmax, a, Q / s[..., None])
    G = np.where(s < R, s - rj, s - R)
    if weighted:
        Dl = 0.5 * np.abs(np.roll(G, -1, 1) - np.roll(G, 1, 1))
        Dk = np.zeros_like(G)
        if nq > 1:
            Dk[:-1] = np.abs(G[1:] - G[:-1])
            Dk[-1] = np.abs(G[-1] - G[-2])
        den = Dk + Dl
        W = np.where(den > 0, np.clip(0.5 - G / np.where(den > 0, den, 1.0), 0.0, 1.0), (G < 0).astype(float))
        W = np.where(s < R, W, 0.0)      # a node outside B_j is outside, whatever its neighbours say
    else:
        W = (G < 0).astype(float)
    S = np.zeros(3)
    for k, l in zip(*np.nonzero(W > 0)):
        r, gF = O.sh_eval(lmax, a, Ui[k, l], grad=True)
        tg = gF - (Ui[k, l] @ gF) * Ui[k, l]
        A = Ri @ (r * r * Ui[k, l] - r * tg)
        S += W[k, l] * OM[k, l] * A
    return S


if __name__ == "__main__":
    rng = np.random.default_rng(5)
    lmax = 6
    a = shapes.random_shape(lmax, bed.SEED0 + 2)
    R = O.shape_rmax(lmax, a)
    nqs = (6, 8, 10, 12, 16, 24, 32)
    err = {(nq, sm): [] for nq in nqs for sm in (False, True)}
    n = 0
    while n < 30:
        qi = rng.normal(size=4); qi /= np.linalg.norm(qi)
        qj = rng.normal(size=4); qj /= np.linalg.norm(qj)
        d = rng.normal(size=3); d *= rng.uniform(1.75, 1.95) / np.linalg.norm(d)
        ref = 0.5 * (rule(lmax, a, R, qi, qj, d, 160, True) + rule(lmax, a, R, qi, qj, d, 160, False))
        if np.linalg.norm(ref) < 2e-2:
            continue
        n += 1
        for nq in nqs:
            for sm in (False, True):
                err[(nq, sm)].append(np.linalg.norm(rule(lmax, a, R, qi, qj, d, nq, sm) - ref) / np.linalg.norm(ref))
    print("|dS_n|/|S_n|   sharp median / max        weighted median / max")
    for nq in nqs:
        es, ew = np.array(err[(nq, False)]), np.array(err[(nq, True)])
        print(f"n_q {nq:3d}      {np.median(es):.2e} / {es.max():.2e}      {np.median(ew):.2e} / {ew.max():.2e}")
