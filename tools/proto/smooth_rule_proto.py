"""Prototype (CPU, numpy): convergence of the cap rule with a sharp inside test (docs/SPEC.md §2.4 as it is)
against the same rule with each node weighted by the covered fraction of its cell, estimated from the
residual g = s - r_j that the rule computes anyway (w = clip(1/2 - g / (|d_k g| + |d_l g|), 0, 1), index-space
central differences).  Integrand: r_i^2 over the part of i's cap inside j (the scalar part of the vector
area).  Evidence for DESIGN.md's 'next lever'; nothing here is used by the product or the tests."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import shapes, bed  # noqa: E402


def rotmat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def rule(lmax, a, R, qi, qj, d, nq, smooth):
    rho = np.linalg.norm(d)
    cosa = np.sqrt(rho * rho - R * R) / rho if rho * rho - R * R <= R * R else (rho * rho) / (2 * rho * R)
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
    ri = shapes.sh_radius_np(lmax, a, U @ rotmat(qi))
    Q = (ri[..., None] * U - d) @ rotmat(qj)
    s = np.linalg.norm(Q, axis=-1)
    G = s - shapes.sh_radius_np(lmax, a, Q / s[..., None])
    if not smooth:
        W = (G < 0).astype(float)
    else:
        gl = 0.5 * (np.roll(G, -1, 1) - np.roll(G, 1, 1))          # azimuth: periodic
        gk = np.gradient(G, axis=0)                                 # rings: one-sided at the ends
        W = np.clip(0.5 - G / (np.abs(gk) + np.abs(gl) + 1e-300), 0.0, 1.0)
    return (W * OM * ri * ri).sum()


rng = np.random.default_rng(5)
lmax = 6
a = shapes.random_shape(lmax, bed.SEED0 + 2)
R = 1.01 * shapes.sh_radius_np(lmax, a, rng.normal(size=(20000, 3)) / 1.0 if False else
                               (lambda v: v / np.linalg.norm(v, axis=1, keepdims=True))(rng.normal(size=(20000, 3)))).max()
nqs = (6, 8, 10, 12, 16, 24, 32)
err = {(nq, sm): [] for nq in nqs for sm in (False, True)}
n = 0
while n < 40:
    qi = rng.normal(size=4); qi /= np.linalg.norm(qi)
    qj = rng.normal(size=4); qj /= np.linalg.norm(qj)
    d = rng.normal(size=3); d *= rng.uniform(1.75, 1.95) / np.linalg.norm(d)
    ref = rule(lmax, a, R, qi, qj, d, 256, True)
    if ref < 1e-3:
        continue
    n += 1
    for nq in nqs:
        for sm in (False, True):
            err[(nq, sm)].append(abs(rule(lmax, a, R, qi, qj, d, nq, sm) - ref) / ref)
print("n_q   sharp median / max        covered-fraction median / max")
for nq in nqs:
    es, ew = np.array(err[(nq, False)]), np.array(err[(nq, True)])
    print(f"{nq:3d}   {np.median(es):.2e} / {es.max():.2e}      {np.median(ew):.2e} / {ew.max():.2e}")
