"""ctypes loader for the CPU oracle (oracle/shpair_oracle.c).

TEST INFRASTRUCTURE ONLY — import this from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never from the product package.
PARITY UNPINNED: see the header of shpair_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    so = os.path.join(_HERE, "libshpair_oracle.so")
    srcs = [os.path.join(_HERE, n) for n in ("shpair_oracle.c", "shstep_oracle.c", "Makefile")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(p) for p in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libshpair_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libshpair_oracle.so")
        build()
        L = C.CDLL(so)
        L.sho_sh_eval.restype = C.c_double
        L.sho_sh_eval.argtypes = [C.c_int, _dp, _dp, _dp]
        L.sho_gauss_legendre.restype = None
        L.sho_gauss_legendre.argtypes = [C.c_int, _dp, _dp]
        L.sho_shape_rmax.restype = C.c_double
        L.sho_shape_rmax.argtypes = [C.c_int, _dp]
        L.sho_pair.restype = C.c_int
        L.sho_pair.argtypes = [C.c_int, _dp, C.c_double, C.c_int, _dp, C.c_double,
                               _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp]
        L.sho_compute.restype = C.c_int
        L.sho_compute.argtypes = [C.c_int, _ip, _ip, _dp, _dp, C.c_int, _dp, _dp, C.c_int,
                                  C.c_int, _dp, _dp, _ip, _ip,
                                  C.c_int, _ip, _ip, _ip, C.c_int,
                                  C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                  C.POINTER(C.c_longlong), _dp, C.c_int]
        L.sho_max_threads.restype = C.c_int
        L.sho_set_rule.restype = None
        L.sho_set_rule.argtypes = [C.c_int]
        L.sho_set_peratom.restype = None
        L.sho_set_peratom.argtypes = [_dp, _dp]
        L.sho_mass_props.restype = None
        L.sho_mass_props.argtypes = [C.c_int, _dp, _dp]
        L.sho_nve.restype = None
        L.sho_nve.argtypes = [C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int]
        L.sho_post_force.restype = None
        L.sho_post_force.argtypes = [C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, _dp, _dp, _dp, _ip, _ip,
                                     C.c_int, _dp, _dp]
        L.sho_energies.restype = None
        L.sho_energies.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int, _dp]
        L.sho_borders.restype = C.c_int
        L.sho_borders.argtypes = [C.c_int, _dp, _dp, _dp, _ip, C.c_double, _ip, _ip]
        L.sho_half_list.restype = C.c_int
        L.sho_half_list.argtypes = [C.c_int, C.c_int, _dp, _ip, _ip, _dp, C.c_double, _ip, _ip]
        _LIB = L
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def sh_eval(lmax, anm, u, grad=False):
    anm, pa = _d(anm)
    assert anm.size == (lmax + 1) * (lmax + 2)
    u, pu = _d(u)
    if grad:
        g = np.zeros(3)
        r = lib().sho_sh_eval(lmax, pa, pu, g.ctypes.data_as(_dp))
        return r, g
    return lib().sho_sh_eval(lmax, pa, pu, None)


def gauss_legendre(n):
    t = np.zeros(n)
    w = np.zeros(n)
    lib().sho_gauss_legendre(n, t.ctypes.data_as(_dp), w.ctypes.data_as(_dp))
    return t, w


def shape_rmax(lmax, anm):
    anm, pa = _d(anm)
    return lib().sho_shape_rmax(lmax, pa)


def pair(li, anmi, ri, lj, anmj, rj, xi, qi, xj, qj, nq, need_volume=True):
    """Returns (hit, out[7] = V,S_n,T_n, diag[4])."""
    anmi, pai = _d(anmi)
    anmj, paj = _d(anmj)
    xi, pxi = _d(xi)
    qi, pqi = _d(qi)
    xj, pxj = _d(xj)
    qj, pqj = _d(qj)
    out = np.zeros(7)
    diag = np.zeros(4)
    hit = lib().sho_pair(li, pai, ri, lj, paj, rj, pxi, pqi, pxj, pqj, nq, int(need_volume),
                         out.ctypes.data_as(_dp), diag.ctypes.data_as(_dp))
    return hit, out, diag


def compute(shapes, kn, expo, nq, nlocal, x, quat, type_, shtype, ilist, offs, jlist,
            newton_pair=True, eflag=False, vflag=False, force_volume=False, nthreads=1,
            want_pairs=False, want_peratom=False):
    """shapes: list of (lmax, anm, rmax). kn/expo: (ntypes+1, ntypes+1) arrays.
    Returns dict(f, torque, eng_virial, counts[, pairs])."""
    lmax, plm = _i([s[0] for s in shapes])
    offs_l = np.cumsum([0] + [np.asarray(s[1]).size for s in shapes])[:-1]
    aoff, pao = _i(offs_l)
    anm_all, paa = _d(np.concatenate([np.asarray(s[1], dtype=np.float64).ravel() for s in shapes]))
    rmax, prm = _d([s[2] for s in shapes])
    kn, pkn = _d(kn)
    expo, pex = _d(expo)
    ntypes = kn.shape[0] - 1
    x, px = _d(x)
    quat, pq = _d(quat)
    type_, pt = _i(type_)
    shtype, ps = _i(shtype)
    ilist, pil = _i(ilist)
    offs, pof = _i(offs)
    jlist, pjl = _i(jlist)
    nall = x.shape[0]
    f = np.zeros((nall, 3))
    tq = np.zeros((nall, 3))
    ev = np.zeros(7)
    counts = np.zeros(3, dtype=np.int64)
    pairs = np.zeros((max(1, jlist.size), 7)) if want_pairs else None
    eatom = np.zeros(nall) if want_peratom else None
    vatom = np.zeros((nall, 6)) if want_peratom else None
    if want_peratom:
        lib().sho_set_peratom(eatom.ctypes.data_as(_dp), vatom.ctypes.data_as(_dp))
    lib().sho_compute(len(shapes), plm, pao, paa, prm, ntypes, pkn, pex, nq,
                      nlocal, px, pq, pt, ps, ilist.size, pil, pof, pjl, int(newton_pair),
                      int(eflag), int(vflag), int(force_volume),
                      f.ctypes.data_as(_dp), tq.ctypes.data_as(_dp), ev.ctypes.data_as(_dp),
                      counts.ctypes.data_as(C.POINTER(C.c_longlong)),
                      pairs.ctypes.data_as(_dp) if want_pairs else None, nthreads)
    if want_peratom:
        lib().sho_set_peratom(None, None)
    out = dict(f=f, torque=tq, eng_virial=ev, counts=counts)
    if want_peratom:
        out["eatom"], out["vatom"] = eatom, vatom
    if want_pairs:
        out["pairs"] = pairs[: jlist.size]
    return out


def max_threads():
    return lib().sho_max_threads()


# ---- docs/SPEC.md Part II (shstep_oracle.c) ---------------------------------------------------

def mass_props(lmax, anm):
    """(V, c[3], J_c xx,yy,zz,xy,xz,yz) of a shape at unit density."""
    anm, pa = _d(anm)
    out = np.zeros(10)
    lib().sho_mass_props(lmax, pa, out.ctypes.data_as(_dp))
    return out


def nve(phase, dt, massprops, density, x, v, quat, angmom, f, torque, shtype, mask, groupbit=1):
    """In place on x, v, quat, angmom (float64 C-contiguous arrays). phase 0 = initial, 1 = final."""
    mp, pmp = _d(massprops)
    rho, prho = _d(density)
    for a in (x, v, quat, angmom):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    f, pf = _d(f)
    torque, pt = _d(torque)
    shtype, ps = _i(shtype)
    mask, pm = _i(mask)
    lib().sho_nve(phase, x.shape[0], dt, pmp, prho, x.ctypes.data_as(_dp), v.ctypes.data_as(_dp),
                  quat.ctypes.data_as(_dp), angmom.ctypes.data_as(_dp), pf, pt, ps, pm, groupbit)


def post_force(massprops, density, g, gamma_t, gamma_r, v, quat, angmom, shtype, mask, f, torque, groupbit=1):
    mp, pmp = _d(massprops)
    rho, prho = _d(density)
    g, pg = _d(g)
    v, pv = _d(v)
    quat, pq = _d(quat)
    angmom, pl = _d(angmom)
    shtype, ps = _i(shtype)
    mask, pm = _i(mask)
    for a in (f, torque):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    lib().sho_post_force(v.shape[0], pmp, prho, pg, gamma_t, gamma_r, pv, pq, pl, ps, pm, groupbit,
                         f.ctypes.data_as(_dp), torque.ctypes.data_as(_dp))


def energies(massprops, density, g, x, v, quat, angmom, shtype, mask, groupbit=1):
    """(translational KE, rotational KE, gravitational PE)."""
    mp, pmp = _d(massprops)
    rho, prho = _d(density)
    g, pg = _d(g)
    x, px = _d(x)
    v, pv = _d(v)
    quat, pq = _d(quat)
    angmom, pl = _d(angmom)
    shtype, ps = _i(shtype)
    mask, pm = _i(mask)
    out = np.zeros(3)
    lib().sho_energies(x.shape[0], pmp, prho, pg, px, pv, pq, pl, ps, pm, groupbit, out.ctypes.data_as(_dp))
    return out


def borders(x, lo, hi, periodic, cmax):
    """Wraps x in place; returns (ghost_owner[ng], ghost_shift[ng,3])."""
    assert x.dtype == np.float64 and x.flags.c_contiguous
    n = x.shape[0]
    lo, plo = _d(lo)
    hi, phi = _d(hi)
    per, pper = _i(periodic)
    own = np.zeros(26 * max(n, 1), dtype=np.int32)
    sh = np.zeros((26 * max(n, 1), 3), dtype=np.int32)
    ng = lib().sho_borders(n, x.ctypes.data_as(_dp), plo, phi, pper, cmax, own.ctypes.data_as(_ip),
                           sh.ctypes.data_as(_ip))
    return own[:ng].copy(), sh[:ng].copy()


def half_list(nlocal, x, shtype, tag, rmax, skin):
    """Brute-force SPEC §7 half list: (offsets[nlocal+1], jlist)."""
    x, px = _d(x)
    shtype, ps = _i(shtype)
    tag, pt = _i(tag)
    rmax, pr = _d(rmax)
    offs = np.zeros(nlocal + 1, dtype=np.int32)
    n = lib().sho_half_list(nlocal, x.shape[0], px, ps, pt, pr, skin, offs.ctypes.data_as(_ip), None)
    jl = np.zeros(max(n, 1), dtype=np.int32)
    lib().sho_half_list(nlocal, x.shape[0], px, ps, pt, pr, skin, offs.ctypes.data_as(_ip), jl.ctypes.data_as(_ip))
    return offs, jl[:n]


def set_rule(rule):
    """0 / "sharp": the inside test of SPEC §2.5; 1 / "weighted": covered-fraction weights of SPEC §2.8.
    Applies to the following pair() / compute() calls of this process."""
    lib().sho_set_rule(1 if rule in (1, True, "weighted") else 0)
