"""Closed-form and brute-force known answers for one pair (SPEC §2)."""
import numpy as np
import pytest

from shpair import shapes

Q0 = np.array([1.0, 0.0, 0.0, 0.0])


def rot_quat(axis, ang):
    axis = np.asarray(axis, float) / np.linalg.norm(axis)
    return np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])


def qmul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return np.array([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def lens_volume(R, r, d):
    return np.pi * (R + r - d) ** 2 * (d * d + 2 * d * (R + r) - 3 * (R - r) ** 2) / (12 * d)


@pytest.mark.parametrize("R,r,d", [(1.0, 1.0, 1.9), (1.0, 0.7, 1.5), (0.8, 1.3, 1.9), (1.0, 1.0, 1.5)])
def test_spheres_with_exact_bounding_radius_give_lens_volume_and_cap_area(oracle, R, r, d):
    rng = np.random.default_rng(1)
    c = rng.normal(size=3)
    c /= np.linalg.norm(c)
    xi = rng.normal(size=3)
    xj = xi + d * c
    hit, out, diag = oracle.pair(0, shapes.sphere(R), R, 0, shapes.sphere(r), r, xi, rot_quat([1, 2, 3], 0.7),
                                 xj, rot_quat([3, 1, -2], 1.9), 24)
    assert hit == 1
    # every cap node lies inside j: the quadrature is spectrally accurate
    assert diag[0] == 2 * 24 * 24
    a2 = R * R - ((d * d + R * R - r * r) / (2 * d)) ** 2
    assert abs(out[0] - lens_volume(R, r, d)) < 1e-8 * lens_volume(R, r, d)
    assert np.abs(out[1:4] - np.pi * a2 * c).max() < 1e-12
    assert np.abs(out[4:7]).max() < 1e-13  # no torque between spheres


def test_deep_overlap_uses_the_tangent_cone_cap(oracle):
    """rho^2 - Rj^2 <= Ri^2: the cap is the tangent cone of B_j, wider than the lens rim."""
    R = r = 1.0
    d = 1.2
    _, out, diag = oracle.pair(0, shapes.sphere(R), R, 0, shapes.sphere(r), r, [0, 0, 0], Q0, [0, 0, d], Q0, 64)
    assert abs(diag[3] - np.sqrt(d * d - r * r) / d) < 1e-15
    assert 0 < diag[0] < 2 * 64 * 64
    # model limitation (SPEC §2.5): rays that cross the lens but end outside j are not counted,
    # so V undershoots the lens volume in this deep, unphysical regime
    assert 0.9 * lens_volume(R, r, d) < out[0] < lens_volume(R, r, d)
    a2 = R * R - (d / 2) ** 2
    assert abs(out[3] - np.pi * a2) < 0.02 * np.pi * a2 and np.abs(out[1:3]).max() < 1e-12


def test_spheres_default_bounding_radius_converges_with_nq(oracle):
    R, r, d = 1.0, 1.0, 1.85
    errs = []
    for nq in (8, 16, 32, 64):
        _, out, _ = oracle.pair(0, shapes.sphere(R), 1.01 * R, 0, shapes.sphere(r), 1.01 * r, [0, 0, 0], Q0,
                                [0, 0, d], Q0, nq)
        errs.append(abs(out[0] - lens_volume(R, r, d)) / lens_volume(R, r, d))
    assert errs[-1] < 2e-3 and errs[-1] < errs[0]


def test_separated_and_barely_bounding_pairs_are_exact_zero(oracle):
    a = shapes.random_shape(6, 3)
    rm = oracle.shape_rmax(6, a)
    hit, out, _ = oracle.pair(6, a, rm, 6, a, rm, [0, 0, 0], Q0, [2 * rm + 1e-9, 0, 0], Q0, 16)
    assert hit == 0 and not out.any()
    # bounding spheres overlap by a hair: a contact pair, but no node is inside
    hit, out, diag = oracle.pair(6, a, rm, 6, a, rm, [0, 0, 0], Q0, [2 * rm - 1e-6, 0, 0], Q0, 16)
    assert hit == 1 and not out.any() and np.all(np.isfinite(out))


def test_coincident_centres_contribute_nothing(oracle):
    """docs/SPEC.md 2, step 1: rho = 0 (or a separation that is not a number) has no line of centres; the pair is not a
    contact pair and contributes exact zeros (both rules) instead of 0/0."""
    a = shapes.random_shape(6, 3)
    rm = oracle.shape_rmax(6, a)
    for rule in ("sharp", "weighted"):
        oracle.set_rule(rule)
        try:
            for xj in ([0.0, 0.0, 0.0], [np.nan, 0.0, 0.0]):
                hit, out, _ = oracle.pair(6, a, rm, 6, a, rm, [0, 0, 0], Q0, xj, Q0, 12)
                assert hit == 0 and not out.any()
        finally:
            oracle.set_rule("sharp")


def test_translation_invariance(oracle):
    a, b = shapes.random_shape(6, 3, amp=0.3), shapes.random_shape(4, 4, amp=0.3)
    ra, rb = oracle.shape_rmax(6, a), oracle.shape_rmax(4, b)
    qi, qj = rot_quat([1, 1, 0], 0.4), rot_quat([0, 1, 1], 2.0)
    xi, xj = np.array([0.1, 0.2, 0.3]), np.array([1.2, 1.1, 0.9])
    _, o1, _ = oracle.pair(6, a, ra, 4, b, rb, xi, qi, xj, qj, 16)
    sh = np.array([0.5, -0.25, 2.0])  # exactly representable shift
    _, o2, _ = oracle.pair(6, a, ra, 4, b, rb, xi + sh, qi, xj + sh, qj, 16)
    assert o1[0] > 0 and np.abs(o1 - o2).max() < 1e-12 * np.abs(o1).max()


def test_body_z_rotation_equals_coefficient_phase(oracle):
    """Particle j turned by phi0 about its own z axis is the same body as j with a_nm e^{-i m phi0}."""
    lmax = 6
    a = shapes.random_shape(lmax, 8, amp=0.3)
    b = shapes.random_shape(lmax, 9, amp=0.3)
    ra, rb = oracle.shape_rmax(lmax, a), oracle.shape_rmax(lmax, b)
    phi0 = 0.83
    b2 = b.reshape(-1, 2).copy()
    for n in range(lmax + 1):
        for m in range(n + 1):
            k = n * (n + 1) // 2 + m
            z = (b2[k, 0] + 1j * b2[k, 1]) * np.exp(-1j * m * phi0)
            b2[k] = (z.real, z.imag)
    qi, qj = rot_quat([1, 0, 1], 0.3), rot_quat([1, 2, 0], 1.1)
    xi, xj = np.zeros(3), np.array([1.0, 1.2, 0.8])
    _, o1, _ = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qi, xj, qmul(qj, rot_quat([0, 0, 1], phi0)), 16)
    _, o2, _ = oracle.pair(lmax, a, ra, lmax, b2.ravel(), rb, xi, qi, xj, qj, 16)
    assert o1[0] > 0 and np.abs(o1 - o2).max() < 1e-11 * np.abs(o1).max()


def test_overlap_volume_against_monte_carlo(oracle):
    lmax = 4
    a, b = shapes.ellipsoid(1.0, 0.8, 0.6, lmax), shapes.random_shape(lmax, 5, amp=0.3)
    ra, rb = oracle.shape_rmax(lmax, a), oracle.shape_rmax(lmax, b)
    qi, qj = rot_quat([1, 2, 3], 0.9), rot_quat([-1, 0, 2], 2.2)
    xi, xj = np.zeros(3), np.array([0.9, 0.8, 0.7])
    _, out, _ = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qi, xj, qj, 64)

    def rotmat(q):
        w, x, y, z = q
        return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])
    rng = np.random.default_rng(3)
    lo = np.maximum(xi - ra, xj - rb)
    hi = np.minimum(xi + ra, xj + rb)
    n = 1_500_000
    p = rng.uniform(lo, hi, size=(n, 3))

    def inside(p, x, q, anm):
        v = (p - x) @ rotmat(q)  # R^T (p - x)
        s = np.linalg.norm(v, axis=1)
        return s < shapes.sh_radius_np(lmax, anm, v / s[:, None])
    frac = np.mean(inside(p, xi, qi, a) & inside(p, xj, qj, b))
    vmc = frac * np.prod(hi - lo)
    sigma = np.sqrt(frac * (1 - frac) / n) * np.prod(hi - lo)
    assert out[0] > 0.01
    assert abs(out[0] - vmc) < 5 * sigma + 2e-3 * vmc


def test_vector_area_is_the_gradient_of_the_overlap_volume(oracle):
    """dV/dx_i = S_n and dV/dtheta_i = T_n: ties V (ray integral) to S_n, T_n (surface integrals)."""
    lmax = 4
    a, b = shapes.random_shape(lmax, 21, amp=0.25), shapes.random_shape(lmax, 22, amp=0.25)
    ra, rb = oracle.shape_rmax(lmax, a), oracle.shape_rmax(lmax, b)
    qi, qj = rot_quat([1, 2, 3], 0.5), rot_quat([2, -1, 1], 1.4)
    xi, xj = np.zeros(3), np.array([1.1, 0.9, 0.8])
    nq = 96
    _, o, _ = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qi, xj, qj, nq)
    h = 1e-4
    for k in range(3):
        e = np.zeros(3)
        e[k] = h
        vp = oracle.pair(lmax, a, ra, lmax, b, rb, xi + e, qi, xj, qj, nq)[1][0]
        vm = oracle.pair(lmax, a, ra, lmax, b, rb, xi - e, qi, xj, qj, nq)[1][0]
        assert abs((vp - vm) / (2 * h) - o[1 + k]) < 0.03 * np.linalg.norm(o[1:4])
        # rotate i about its centre around axis k (space frame)
        axis = np.zeros(3)
        axis[k] = 1.0
        vp = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qmul(rot_quat(axis, h), qi), xj, qj, nq)[1][0]
        vm = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qmul(rot_quat(axis, -h), qi), xj, qj, nq)[1][0]
        assert abs((vp - vm) / (2 * h) - o[4 + k]) < 0.03 * np.linalg.norm(o[4:7]) + 2e-3 * np.linalg.norm(o[1:4])


def test_centre_of_i_inside_j_is_finite(oracle):
    big, small = shapes.sphere(2.0), shapes.random_shape(4, 2, amp=0.2)
    rs = oracle.shape_rmax(4, small)
    hit, out, diag = oracle.pair(4, small, rs, 0, big, 2.02, [0, 0, 0], Q0, [0.5, 0, 0], Q0, 16)
    assert hit == 1 and np.all(np.isfinite(out)) and diag[3] == -1.0
    vol = 0.0  # the whole of i lies inside j: V = volume of i = integral r^3/3
    u, wg, _, _ = shapes._sphere_grid(24)
    vol = np.sum(shapes.sh_radius_np(4, small, u) ** 3 / 3.0 * wg)
    assert abs(out[0] - vol) < 1e-3 * vol
    assert np.abs(out[1:4]).max() < 1e-3  # closed surface: vector area vanishes


def test_forces_only_mode_skips_the_volume(oracle):
    a = shapes.random_shape(6, 3, amp=0.3)
    ra = oracle.shape_rmax(6, a)
    args = (6, a, ra, 6, a, ra, [0, 0, 0], rot_quat([1, 0, 0], 0.3), [1.2, 1.0, 0.6], rot_quat([0, 1, 0], 1.0), 16)
    _, o1, d1 = oracle.pair(*args, need_volume=True)
    _, o0, d0 = oracle.pair(*args, need_volume=False)
    assert o1[0] > 0 and o0[0] == 0.0 and np.array_equal(o1[1:], o0[1:]) and d0[2] == 0 and d1[2] > 0


def numpy_pair(lmax_i, a, Ri, lmax_j, b, Rj, xi, qi, xj, qj, nq):
    """Independent restatement of SPEC §2 in numpy/scipy: Gauss nodes from numpy, r(u) from the
    vectorised setup evaluator, inner radii by scipy.optimize.brentq to 1e-15."""
    from scipy.optimize import brentq

    def rotmat(q):
        w, x, y, z = q
        return np.array([[w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y)],
                         [2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x)],
                         [2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z]])
    d = np.asarray(xj, float) - np.asarray(xi, float)
    rho = np.linalg.norm(d)
    assert Rj < rho < Ri + Rj
    if rho * rho - Rj * Rj <= Ri * Ri:
        cosa = np.sqrt(rho * rho - Rj * Rj) / rho
    else:
        cosa = (rho * rho + Ri * Ri - Rj * Rj) / (2 * rho * Ri)
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
    Rmi, Rmj = rotmat(qi), rotmat(qj)
    Ui = U @ Rmi
    ri = shapes.sh_radius_np(lmax_i, a, Ui)
    P = ri[..., None] * U - d
    Qj = P @ Rmj
    s = np.linalg.norm(Qj, axis=-1)
    rj = shapes.sh_radius_np(lmax_j, b, Qj / s[..., None])
    inside = (s < Rj) & (s < rj)
    V = 0.0
    for k, l in zip(*np.nonzero(inside)):
        u = U[k, l]

        def g(lam):
            q = (lam * u - d) @ Rmj
            sn = np.linalg.norm(q)
            return sn - shapes.sh_radius_np(lmax_j, b, q / sn)
        bp = u @ d
        lo = bp - np.sqrt(max(0.0, bp * bp - (rho * rho - Rj * Rj)))
        rin = brentq(g, lo, ri[k, l], xtol=1e-15, rtol=1e-15)
        V += OM[k, l] * (ri[k, l] ** 3 - rin ** 3) / 3.0
    return V, int(inside.sum())


def test_pair_against_independent_numpy_implementation(oracle):
    """Same nodes, independent code: classification identical, V to the accuracy SPEC §2.6 promises
    for the extrapolated inner radius (the quadrature itself is identical on both sides)."""
    lmax = 6
    a, b = shapes.random_shape(lmax, 31, amp=0.3), shapes.random_shape(lmax, 32, amp=0.3)
    ra, rb = oracle.shape_rmax(lmax, a), oracle.shape_rmax(lmax, b)
    qi, qj = rot_quat([1, 2, 3], 0.5), rot_quat([2, -1, 1], 1.4)
    xi, xj = np.zeros(3), np.array([1.15, 0.95, 0.85])
    for nq in (8, 16):
        _, o, diag = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qi, xj, qj, nq)
        V, nin = numpy_pair(lmax, a, ra, lmax, b, rb, xi, qi, xj, qj, nq)
        assert nin == diag[0] and nin > 20
        assert abs(o[0] - V) < 1e-6 * V   # deep overlap (V = 12 % of a particle); shallow contacts: ~2e-8
        assert diag[2] / diag[0] < 4.0    # evaluations per inside node


def _rotate_pair(Q, xi, qi, xj, qj):
    """The whole pair turned by the rotation quaternion Q about x_i."""
    w, x, y, z = Q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    return R, xi, qmul(Q, qi), xi + R @ (xj - xi), qmul(Q, qj)


def test_rotation_about_the_pair_axis_by_a_node_spacing_is_exact(oracle):
    """Whole-pair rotation covariance, as far as the discrete rule has it: the cap nodes sit at n_psi
    equally spaced azimuths about the pair axis, so a rigid rotation of both bodies about that axis by a multiple of
    2 pi / n_psi maps the node set onto itself.  V is then unchanged and S_n, T_n turn with the bodies to rounding.
    (For a general rotation the frame (e1, e2) of SPEC §2.3 does not turn with the pair, the nodes land elsewhere on
    the surfaces and the sums differ by the quadrature error: next test.)"""
    lmax, nq = 6, 12
    a, b = shapes.random_shape(lmax, 21, amp=0.3), shapes.random_shape(lmax, 22, amp=0.3)
    ra, rb = oracle.shape_rmax(lmax, a), oracle.shape_rmax(lmax, b)
    rng = np.random.default_rng(4)
    for trial in range(6):
        qi = rng.normal(size=4); qi /= np.linalg.norm(qi)
        qj = rng.normal(size=4); qj /= np.linalg.norm(qj)
        c = rng.normal(size=3); c /= np.linalg.norm(c)
        xi = rng.normal(size=3)
        xj = xi + rng.uniform(1.3, 1.9) * c
        _, o1, _ = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qi, xj, qj, nq)
        assert o1[0] > 0
        k = 1 + trial
        Q = rot_quat(c, 2 * np.pi * k / (2 * nq))
        R, xi2, qi2, xj2, qj2 = _rotate_pair(Q, xi, qi, xj, qj)
        assert np.abs(xj2 - xj).max() < 1e-14          # the axis is fixed
        _, o2, _ = oracle.pair(lmax, a, ra, lmax, b, rb, xi2, qi2, xj, qj2, nq)
        sc = np.abs(o1).max()
        assert abs(o2[0] - o1[0]) < 1e-12 * sc
        assert np.abs(o2[1:4] - R @ o1[1:4]).max() < 1e-12 * sc and np.abs(o2[4:7] - R @ o1[4:7]).max() < 1e-12 * sc


def test_general_rotation_is_covariant_to_quadrature_accuracy(oracle):
    lmax = 5
    a, b = shapes.random_shape(lmax, 31, amp=0.25), shapes.random_shape(lmax, 32, amp=0.25)
    ra, rb = oracle.shape_rmax(lmax, a), oracle.shape_rmax(lmax, b)
    rng = np.random.default_rng(8)
    dev = {}
    for nq in (8, 48):
        worst = 0.0
        for _ in range(5):
            qi = rng.normal(size=4); qi /= np.linalg.norm(qi)
            qj = rng.normal(size=4); qj /= np.linalg.norm(qj)
            c = rng.normal(size=3); c /= np.linalg.norm(c)
            xi = rng.normal(size=3)
            xj = xi + 1.6 * c
            Q = rng.normal(size=4); Q /= np.linalg.norm(Q)
            R, xi2, qi2, xj2, qj2 = _rotate_pair(Q, xi, qi, xj, qj)
            _, o1, _ = oracle.pair(lmax, a, ra, lmax, b, rb, xi, qi, xj, qj, nq)
            _, o2, _ = oracle.pair(lmax, a, ra, lmax, b, rb, xi2, qi2, xj2, qj2, nq)
            s = np.linalg.norm(o1[1:4])
            worst = max(worst, np.linalg.norm(o2[1:4] - R @ o1[1:4]) / s, abs(o2[0] - o1[0]) / o1[0])
        dev[nq] = worst
    assert dev[48] < 2e-2 and dev[48] < 0.5 * dev[8], dev
