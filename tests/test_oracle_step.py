"""Known answers that pin the Part II oracle (oracle/shstep_oracle.c, docs/SPEC.md §5-§7) without the
reference: closed-form mass properties, exact ballistic flight, analytic torque-free precession with
second-order convergence, conservation laws, and periodic neighbour sets against scipy's cKDTree."""
import numpy as np
import pytest
from scipy.spatial import cKDTree

from shpair import shapes


def quat_to_mat(q):
    w, x, y, z = q
    return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])


def tensor(mp):
    return np.array([[mp[4], mp[7], mp[8]], [mp[7], mp[5], mp[9]], [mp[8], mp[9], mp[6]]])


def random_quats(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


# ---- §5 mass properties -------------------------------------------------------------------------

def test_sphere_mass_props(oracle):
    for R in (0.5, 1.0, 1.7):
        mp = oracle.mass_props(0, shapes.sphere(R))
        V = 4 / 3 * np.pi * R ** 3
        assert abs(mp[0] - V) < 1e-14 * V
        assert np.abs(mp[1:4]).max() < 1e-15
        assert np.allclose(mp[4:7], 0.4 * V * R * R, rtol=1e-14)
        assert np.abs(mp[7:]).max() < 1e-15


def test_mass_props_match_dense_numpy_quadrature(oracle):
    """Independent evaluation: numpy recurrence for r, a much denser product grid."""
    lmax = 6
    a = shapes.random_shape(lmax, 3, amp=0.3)
    mp = oracle.mass_props(lmax, a)
    t, w = np.polynomial.legendre.leggauss(64)
    ph = 2 * np.pi * (np.arange(128) + 0.25) / 128
    T, P = np.meshgrid(t, ph, indexing="ij")
    s = np.sqrt(1 - T * T)
    u = np.stack([s * np.cos(P), s * np.sin(P), T], axis=-1)
    r = shapes.sh_radius_np(lmax, a, u)
    dw = (w[:, None] * (2 * np.pi / 128)) * np.ones_like(P)
    V = (dw * r ** 3 / 3).sum()
    c = (dw[..., None] * (r ** 4 / 4)[..., None] * u).sum((0, 1)) / V
    JO = (dw[..., None, None] * (r ** 5 / 5)[..., None, None] * (np.eye(3) - u[..., :, None] * u[..., None, :])).sum((0, 1))
    Jc = JO - V * (c @ c * np.eye(3) - np.outer(c, c))
    assert abs(mp[0] - V) < 1e-13 * V
    assert np.abs(mp[1:4] - c).max() < 1e-13
    assert np.abs(tensor(mp) - Jc).max() < 1e-13 * np.abs(Jc).max()
    assert np.abs(c).max() > 1e-3          # the n = 1 terms do move the centroid: the offset matters


def test_offset_sphere_centroid_and_parallel_axis(oracle):
    """A sphere of radius R whose centre sits at c0 from the SH origin, projected to L = 14."""
    R, c0 = 1.0, np.array([0.05, -0.08, 0.1])

    def rad(u):
        b = u @ c0
        return b + np.sqrt(b * b + R * R - c0 @ c0)
    lmax = 14
    a = shapes.project(rad, lmax)
    mp = oracle.mass_props(lmax, a)
    V = 4 / 3 * np.pi * R ** 3
    assert abs(mp[0] - V) < 1e-8 * V
    assert np.abs(mp[1:4] - c0).max() < 1e-8
    assert np.abs(tensor(mp) - 0.4 * V * R * R * np.eye(3)).max() < 1e-7


def test_ellipsoid_mass_props_close_to_closed_form(oracle):
    ax = np.array([1.0, 0.9, 0.8])
    lmax = 12
    a = shapes.ellipsoid(*ax, lmax=lmax)
    mp = oracle.mass_props(lmax, a)
    V = 4 / 3 * np.pi * ax.prod()
    J = V / 5 * np.array([ax[1] ** 2 + ax[2] ** 2, ax[0] ** 2 + ax[2] ** 2, ax[0] ** 2 + ax[1] ** 2])
    assert abs(mp[0] - V) < 2e-4 * V        # band-limit error of the projection, not of the integrals
    assert np.abs(mp[4:7] - J).max() < 5e-4 * J.max()
    assert np.abs(mp[1:4]).max() < 1e-12 and np.abs(mp[7:]).max() < 1e-12


# ---- §6 integrator ------------------------------------------------------------------------------

def _state(rng, n, nshapes):
    return dict(x=rng.normal(size=(n, 3)), v=rng.normal(size=(n, 3)), quat=random_quats(rng, n),
                angmom=rng.normal(size=(n, 3)), f=rng.normal(size=(n, 3)), torque=rng.normal(size=(n, 3)),
                shtype=rng.integers(0, nshapes, n).astype(np.int32), mask=np.ones(n, dtype=np.int32))


def test_ballistic_flight_is_exact(oracle):
    """Constant force: velocity Verlet reproduces the parabola of the centre of mass to rounding, whatever
    the body does about it (offset centroid, spinning)."""
    rng = np.random.default_rng(1)
    lmax = 4
    a = shapes.random_shape(lmax, 11, amp=0.3)
    mp = oracle.mass_props(lmax, a)[None]
    rho = np.array([2.5])
    g = np.array([0.3, -0.2, -9.81])
    st = _state(rng, 5, 1)
    m = rho[0] * mp[0, 0]
    X0 = st["x"] + np.einsum("nij,j->ni", np.array([quat_to_mat(q) for q in st["quat"]]), mp[0, 1:4])
    v0 = st["v"].copy()
    dt, nsteps = 1e-3, 400
    for _ in range(nsteps):
        st["f"][:] = 0
        st["torque"][:] = 0
        oracle.post_force(mp, rho, g, 0.0, 0.0, st["v"], st["quat"], st["angmom"], st["shtype"], st["mask"], st["f"], st["torque"])
        oracle.nve(0, dt, mp, rho, st["x"], st["v"], st["quat"], st["angmom"], st["f"], st["torque"], st["shtype"], st["mask"])
        st["f"][:] = 0
        st["torque"][:] = 0
        oracle.post_force(mp, rho, g, 0.0, 0.0, st["v"], st["quat"], st["angmom"], st["shtype"], st["mask"], st["f"], st["torque"])
        oracle.nve(1, dt, mp, rho, st["x"], st["v"], st["quat"], st["angmom"], st["f"], st["torque"], st["shtype"], st["mask"])
    t = dt * nsteps
    X = st["x"] + np.einsum("nij,j->ni", np.array([quat_to_mat(q) for q in st["quat"]]), mp[0, 1:4])
    assert np.abs(X - (X0 + v0 * t + 0.5 * g * t * t)).max() < 1e-11
    assert np.abs(st["v"] - (v0 + g * t)).max() < 1e-11
    assert m > 0


def test_gravity_exerts_no_torque_about_the_centre_of_mass(oracle):
    rng = np.random.default_rng(2)
    lmax = 4
    a = shapes.random_shape(lmax, 12, amp=0.3)
    mp = oracle.mass_props(lmax, a)[None]
    rho = np.array([1.0])
    st = _state(rng, 4, 1)
    L0 = st["angmom"].copy()
    g = np.array([0.0, 0.0, -5.0])
    for _ in range(50):
        for ph in (0, 1):
            st["f"][:] = 0
            st["torque"][:] = 0
            oracle.post_force(mp, rho, g, 0.0, 0.0, st["v"], st["quat"], st["angmom"], st["shtype"], st["mask"], st["f"], st["torque"])
            assert np.abs(st["torque"]).max() > 1e-3       # about the SH origin there IS a torque
            oracle.nve(ph, 1e-3, mp, rho, st["x"], st["v"], st["quat"], st["angmom"], st["f"], st["torque"], st["shtype"], st["mask"])
    assert np.abs(st["angmom"] - L0).max() < 1e-13


def _free_top(oracle, mp, rho, q0, L0, dt, nsteps):
    x = np.zeros((1, 3)); v = np.zeros((1, 3)); q = q0[None].copy(); L = L0[None].copy()
    z = np.zeros((1, 3)); sh = np.zeros(1, dtype=np.int32); mk = np.ones(1, dtype=np.int32)
    for _ in range(nsteps):
        oracle.nve(0, dt, mp, rho, x, v, q, L, z, z, sh, mk)
        oracle.nve(1, dt, mp, rho, x, v, q, L, z, z, sh, mk)
    return q[0], L[0]


def test_symmetric_top_precession_second_order(oracle):
    """Axially symmetric body (m = 0 coefficients only): the symmetry axis precesses about L at |L|/J_1."""
    lmax = 4
    a = np.zeros((shapes.nterms(lmax), 2))
    a[0, 0] = np.sqrt(4 * np.pi)
    a[2 * 3 // 2, 0] = 0.35           # (n, m) = (2, 0)
    a[4 * 5 // 2, 0] = -0.1           # (4, 0)
    mp = oracle.mass_props(lmax, a.ravel())[None]
    assert abs(mp[0, 4] - mp[0, 5]) < 1e-14 and np.abs(mp[0, 7:]).max() < 1e-14 and np.abs(mp[0, 1:3]).max() < 1e-14
    # the centroid lies on the axis; choose even orders only so that it is the origin
    assert abs(mp[0, 3]) < 1e-14
    rho = np.array([1.3])
    J1 = rho[0] * mp[0, 4]
    q0 = np.array([np.cos(0.35), np.sin(0.35), 0.0, 0.0])      # axis tilted by 0.7 rad about x
    L0 = np.array([0.0, 0.0, 2.0])
    T = 1.0
    axis0 = quat_to_mat(q0)[:, 2]
    Om = np.linalg.norm(L0) / J1

    def exact(t):
        c, s = np.cos(Om * t), np.sin(Om * t)
        return np.array([c * axis0[0] - s * axis0[1], s * axis0[0] + c * axis0[1], axis0[2]])
    errs = []
    for n in (200, 400, 800):
        q, L = _free_top(oracle, mp, rho, q0, L0, T / n, n)
        assert np.abs(L - L0).max() == 0.0
        errs.append(np.abs(quat_to_mat(q)[:, 2] - exact(T)).max())
    assert errs[0] < 1e-3
    assert 3.5 < errs[0] / errs[1] < 4.5 and 3.5 < errs[1] / errs[2] < 4.5


def test_free_asymmetric_body_conserves_energy_to_second_order(oracle):
    lmax = 6
    a = shapes.random_shape(lmax, 5, amp=0.35)
    mp = oracle.mass_props(lmax, a)[None]
    rho = np.array([1.0])
    rng = np.random.default_rng(3)
    q0 = random_quats(rng, 1)[0]
    L0 = np.array([0.4, -1.0, 0.7])
    g0 = np.zeros(3)

    def ke(q, L):
        return oracle.energies(mp, rho, g0, np.zeros((1, 3)), np.zeros((1, 3)), q[None], L[None],
                               np.zeros(1, dtype=np.int32), np.ones(1, dtype=np.int32))[1]
    e0 = ke(q0, L0)
    drift = []
    for n in (100, 200):
        q, L = _free_top(oracle, mp, rho, q0, L0, 2.0 / n, n)
        assert abs(np.linalg.norm(q) - 1) < 1e-14
        drift.append(abs(ke(q, L) - e0) / e0)
    assert drift[0] < 1e-3 and drift[1] < drift[0] / 3


def test_frozen_particles_and_final_phase(oracle):
    rng = np.random.default_rng(4)
    shp = [shapes.random_shape(4, s, amp=0.3) for s in (1, 2)]
    mp = np.array([oracle.mass_props(4, a) for a in shp])
    rho = np.array([1.0, 3.0])
    st = _state(rng, 20, 2)
    st["mask"][::3] = 2                                  # other group
    ref = {k: v.copy() for k, v in st.items()}
    oracle.nve(0, 1e-2, mp, rho, st["x"], st["v"], st["quat"], st["angmom"], st["f"], st["torque"], st["shtype"], st["mask"], groupbit=1)
    fr = st["mask"] == 2
    for k in ("x", "v", "quat", "angmom"):
        assert np.array_equal(st[k][fr], ref[k][fr])
        assert not np.array_equal(st[k][~fr], ref[k][~fr])
    # final phase: x and quat untouched, kicks as in the SPEC
    st2 = {k: v.copy() for k, v in ref.items()}
    oracle.nve(1, 1e-2, mp, rho, st2["x"], st2["v"], st2["quat"], st2["angmom"], st2["f"], st2["torque"], st2["shtype"], st2["mask"], groupbit=1)
    assert np.array_equal(st2["x"], ref["x"]) and np.array_equal(st2["quat"], ref["quat"])
    i = 1
    m = rho[ref["shtype"][i]] * mp[ref["shtype"][i], 0]
    s = quat_to_mat(ref["quat"][i]) @ mp[ref["shtype"][i], 1:4]
    assert np.allclose(st2["v"][i], ref["v"][i] + 0.5e-2 / m * ref["f"][i], rtol=0, atol=1e-15)
    assert np.allclose(st2["angmom"][i], ref["angmom"][i] + 0.5e-2 * (ref["torque"][i] - np.cross(s, ref["f"][i])), rtol=0, atol=1e-15)


def test_post_force_formula(oracle):
    rng = np.random.default_rng(5)
    a = shapes.random_shape(4, 7, amp=0.3)
    mp = oracle.mass_props(4, a)[None]
    rho = np.array([2.0])
    st = _state(rng, 6, 1)
    f0, t0 = st["f"].copy(), st["torque"].copy()
    g = np.array([0.1, 0.2, -3.0])
    oracle.post_force(mp, rho, g, 0.7, 0.3, st["v"], st["quat"], st["angmom"], st["shtype"], st["mask"], st["f"], st["torque"])
    m = rho[0] * mp[0, 0]
    Iinv = np.linalg.inv(rho[0] * tensor(mp[0]))
    for i in range(6):
        R = quat_to_mat(st["quat"][i])
        s = R @ mp[0, 1:4]
        w = R @ Iinv @ R.T @ st["angmom"][i]
        Fb = m * g - 0.7 * st["v"][i]
        assert np.allclose(st["f"][i], f0[i] + Fb, atol=1e-14)
        assert np.allclose(st["torque"][i], t0[i] + np.cross(s, Fb) - 0.3 * w, atol=1e-13)


# ---- §7 borders and half list -------------------------------------------------------------------

@pytest.mark.parametrize("periodic", [(1, 1, 1), (1, 1, 0), (0, 0, 0), (0, 1, 0)])
def test_periodic_half_list_matches_ckdtree(oracle, periodic):
    rng = np.random.default_rng(6)
    n, box = 600, np.array([9.0, 8.0, 7.0])
    lo = np.array([-1.0, 2.0, 0.5])
    hi = lo + box
    x = lo + rng.uniform(0, 1, (n, 3)) * box
    x[:5] += box * np.array(periodic)            # a few outside: wrapped by borders in periodic dimensions
    rmax = np.array([0.6])
    skin = 0.2
    cmax = 2 * rmax[0] + skin
    xw = x.copy()
    own, sh = oracle.borders(xw, lo, hi, periodic, cmax)
    per = np.array(periodic, bool)
    assert np.all((xw[:, per] >= lo[per]) & (xw[:, per] < hi[per]))
    xa = np.concatenate([xw, xw[own] + sh * box])
    tag = np.concatenate([np.arange(n), own]).astype(np.int32)
    sht = np.zeros(xa.shape[0], dtype=np.int32)
    offs, jl = oracle.half_list(n, xa, sht, tag, rmax, skin)
    got = set()
    for i in range(n):
        for j in jl[offs[i]:offs[i + 1]]:
            p = (i, int(tag[j]))
            assert p[0] < p[1]
            assert p not in got                   # every physical pair exactly once
            got.add(p)
    # reference: minimum-image pairs from scipy (fully periodic tree only when all dims are periodic)
    if all(periodic):
        tree = cKDTree(xw - lo, boxsize=box)
        want = {tuple(sorted(p)) for p in tree.query_pairs(cmax)}
    else:
        d = xw[:, None, :] - xw[None, :, :]
        for k in range(3):
            if periodic[k]:
                d[..., k] -= box[k] * np.round(d[..., k] / box[k])
        r2 = (d * d).sum(-1)
        ii, jj = np.nonzero(np.triu(r2 < cmax * cmax, 1))
        want = set(zip(ii.tolist(), jj.tolist()))
    assert got == want
    assert len(got) > 100


def test_ghost_order_and_count(oracle):
    """One particle in a corner of a fully periodic box has 7 images, ordered by shift code."""
    lo, hi = np.zeros(3), np.full(3, 10.0)
    x = np.array([[0.5, 0.5, 9.7], [5.0, 5.0, 5.0]])
    own, sh = oracle.borders(x, lo, hi, (1, 1, 1), 1.0)
    assert own.tolist() == [0] * 7
    codes = ((sh[:, 2] + 1) * 9 + (sh[:, 1] + 1) * 3 + sh[:, 0] + 1).tolist()
    assert codes == sorted(codes)
    assert set(map(tuple, sh.tolist())) == {(a, b, c) for a in (0, 1) for b in (0, 1) for c in (0, -1)} - {(0, 0, 0)}
