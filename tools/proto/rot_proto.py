"""Prototype (numpy) of the cap-frame evaluation of particle i: real-SH rotation by
Z(alpha) X(-90) Z(beta) X(90) Z(gamma) with constant X matrices, then ring tables."""
import sys, os
import numpy as np
from scipy.special import sph_harm_y
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import shapes


def real_sh(l, m, th, ph):
    if m == 0:
        return sph_harm_y(l, 0, th, ph).real
    Y = sph_harm_y(l, abs(m), th, ph)
    s = np.sqrt(2.0) * (-1) ** abs(m)
    return s * (Y.real if m > 0 else Y.imag)


def to_real(lmax, anm):
    a = np.asarray(anm).reshape(-1, 2)
    c = [np.zeros(2 * l + 1) for l in range(lmax + 1)]  # index m + l
    for l in range(lmax + 1):
        for m in range(l + 1):
            k = l * (l + 1) // 2 + m
            if m == 0:
                c[l][l] = a[k, 0]
            else:
                s = np.sqrt(2.0) * (-1) ** m
                c[l][l + m] = s * a[k, 0]
                c[l][l - m] = -s * a[k, 1]
    return c


def eval_real(lmax, c, u):
    th = np.arccos(np.clip(u[..., 2], -1, 1))
    ph = np.arctan2(u[..., 1], u[..., 0])
    r = 0.0
    for l in range(lmax + 1):
        for m in range(-l, l + 1):
            r = r + c[l][l + m] * real_sh(l, m, th, ph)
    return r


def sphere_quad(n):
    t, w = np.polynomial.legendre.leggauss(n)
    ph = 2 * np.pi * np.arange(2 * n) / (2 * n)
    ct, phg = np.meshgrid(t, ph, indexing="ij")
    st = np.sqrt(1 - ct * ct)
    u = np.stack([st * np.cos(phg), st * np.sin(phg), ct], -1)
    wg = np.repeat(w[:, None], 2 * n, 1) * (2 * np.pi / (2 * n))
    return u, wg


def T_of(A, l, n=None):
    """T(A)_{m'm} = int S_lm'(u) S_lm(A u) dOmega."""
    u, wg = sphere_quad(n or (l + 2))
    th = np.arccos(np.clip(u[..., 2], -1, 1)); ph = np.arctan2(u[..., 1], u[..., 0])
    v = u @ A.T
    thv = np.arccos(np.clip(v[..., 2], -1, 1)); phv = np.arctan2(v[..., 1], v[..., 0])
    T = np.zeros((2 * l + 1, 2 * l + 1))
    for mp in range(-l, l + 1):
        Sp = real_sh(l, mp, th, ph)
        for m in range(-l, l + 1):
            T[mp + l, m + l] = np.sum(wg * Sp * real_sh(l, m, thv, phv))
    return T


def Rx(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def Rz(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])


def Ry(t):
    c, s = np.cos(t), np.sin(t)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def zrot(l, v, cphi, sphi):
    """coefficients of f(Rz(phi) u); cphi/sphi = cos/sin(phi)"""
    o = v.copy()
    cm, sm = 1.0, 0.0
    for m in range(1, l + 1):
        cm, sm = cm * cphi - sm * sphi, cm * sphi + sm * cphi
        o[l + m] = cm * v[l + m] + sm * v[l - m]
        o[l - m] = -sm * v[l + m] + cm * v[l - m]
    return o


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    lmax = 6
    anm = shapes.random_shape(lmax, 5, amp=0.3)
    c = to_real(lmax, anm)
    u = rng.normal(size=(50, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    print("real basis vs complex:", np.abs(eval_real(lmax, c, u) - shapes.sh_radius_np(lmax, anm, u)).max())
    # random rotation M (columns = cap axes in body frame)
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    w, x, y, z = q
    M = np.array([[w*w+x*x-y*y-z*z, 2*(x*y-w*z), 2*(x*z+w*y)], [2*(x*y+w*z), w*w-x*x+y*y-z*z, 2*(y*z-w*x)],
                  [2*(x*z-w*y), 2*(y*z+w*x), w*w-x*x-y*y+z*z]])
    # ZYZ Euler: M = Rz(a) Ry(b) Rz(g)
    cb = M[2, 2]; sb = np.sqrt(max(0.0, 1 - cb * cb))
    ca, sa = M[0, 2] / sb, M[1, 2] / sb
    cg, sg = -M[2, 0] / sb, M[2, 1] / sb
    a, b, g = np.arctan2(sa, ca), np.arctan2(sb, cb), np.arctan2(sg, cg)
    print("euler recon err", np.abs(Rz(a) @ Ry(b) @ Rz(g) - M).max())
    Xp = [T_of(Rx(np.pi / 2), l) for l in range(lmax + 1)]
    Xm = [T_of(Rx(-np.pi / 2), l) for l in range(lmax + 1)]
    print("X orth:", max(np.abs(Xp[l] @ Xm[l] - np.eye(2 * l + 1)).max() for l in range(lmax + 1)),
          "Xm == Xp^T:", max(np.abs(Xm[l] - Xp[l].T).max() for l in range(lmax + 1)))
    cr = []
    for l in range(lmax + 1):
        v = zrot(l, c[l], ca, sa)            # O_{Rz(alpha)} first
        v = Xm[l] @ v                        # O_{Rx(-90)}
        v = zrot(l, v, cb, sb)               # O_{Rz(beta)}
        v = Xp[l] @ v                        # O_{Rx(+90)}
        v = zrot(l, v, cg, sg)               # O_{Rz(gamma)}
        cr.append(v)
    # g(u') = r(M u')
    up = rng.normal(size=(200, 3)); up /= np.linalg.norm(up, axis=1, keepdims=True)
    ref = shapes.sh_radius_np(lmax, anm, up @ M.T)
    got = eval_real(lmax, cr, up)
    print("rotated coefficient error:", np.abs(got - ref).max())
    # sparsity of X
    print("nonzeros in X^6:", (np.abs(Xp[6]) > 1e-12).sum(), "of", 13 * 13)
