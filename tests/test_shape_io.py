"""Input side of the path: the shape text format (written / read by shpair.shapes, read by the PairSH adapter)
and the least-squares fit of an SH expansion to surface points."""
import numpy as np
import pytest

from shpair import shapes


def test_shape_file_round_trip(tmp_path):
    for lmax in (0, 3, 12):
        a = shapes.random_shape(lmax, 5 + lmax, amp=0.3)
        p = tmp_path / f"s{lmax}.txt"
        shapes.write_shape_file(p, lmax, a)
        l2, b = shapes.read_shape_file(p)
        assert l2 == lmax and np.array_equal(a, b)          # repr() round-trips doubles exactly
    (tmp_path / "bad.txt").write_text("2\n3 0 1.0 0.0\n")
    with pytest.raises(ValueError):
        shapes.read_shape_file(tmp_path / "bad.txt")
    (tmp_path / "bad2.txt").write_text("2\n1 0 1.0 0.5\n")
    with pytest.raises(ValueError):
        shapes.read_shape_file(tmp_path / "bad2.txt")
    (tmp_path / "sparse.txt").write_text("# a sphere with one bump\n2\n0 0 3.5449077018110318 0\n2 1 0.1 -0.05  # rest is zero\n")
    lmax, a = shapes.read_shape_file(tmp_path / "sparse.txt")
    assert lmax == 2 and a[0] == 3.5449077018110318 and a[2 * 4] == 0.1 and a[2 * 4 + 1] == -0.05 and np.count_nonzero(a) == 3


def _full_range_text(lmax, a, header=False):
    """The same coefficients as a table over the whole range m = -n..n (a common layout of SH coefficient files)."""
    a = np.asarray(a).reshape(-1, 2)
    out = [f"{lmax}"] if header else []
    for n in range(lmax + 1):
        for m in range(-n, n + 1):
            re, im = a[n * (n + 1) // 2 + abs(m)]
            if m < 0:
                sg = -1.0 if (-m) & 1 else 1.0
                re, im = sg * re, -sg * im
            out.append(f"{n} {m} {float(re)!r} {float(im)!r}")
    return "\n".join(out) + "\n"


def test_shape_file_over_the_whole_range_of_m_and_without_a_header(tmp_path):
    """[PRIOR] the reference's shape-file format is unknown (its reader is absent from the mount); tables `n m Re Im` over
    m = -n..n without an lmax line are a common layout.  Accepted when they describe a real radius; the mirror half is
    checked, not trusted."""
    lmax = 5
    a = shapes.random_shape(lmax, 17, amp=0.3)
    for header in (False, True):
        p = tmp_path / f"full{int(header)}.txt"
        p.write_text(_full_range_text(lmax, a, header))
        l2, b = shapes.read_shape_file(p)
        assert l2 == lmax and np.array_equal(a, b)
    # only the negative half listed: it fills the positive one
    lines = [ln for ln in _full_range_text(lmax, a).split("\n") if ln and int(ln.split()[1]) <= 0]
    (tmp_path / "neg.txt").write_text("\n".join(lines) + "\n")
    l3, c = shapes.read_shape_file(tmp_path / "neg.txt")
    assert l3 == lmax and np.abs(c - a).max() == 0.0
    # a table that is not a real function is refused
    bad = _full_range_text(lmax, a).split("\n")
    k = next(i for i, ln in enumerate(bad) if ln.startswith("3 -2 "))
    n_, m_, re_, im_ = bad[k].split()
    bad[k] = f"{n_} {m_} {float(re_) + 0.01!r} {im_}"
    (tmp_path / "notreal.txt").write_text("\n".join(bad))
    with pytest.raises(ValueError, match="not a real radius"):
        shapes.read_shape_file(tmp_path / "notreal.txt")


def test_basis_matrix_reproduces_the_radius():
    lmax = 7
    a = shapes.random_shape(lmax, 9, amp=0.3)
    rng = np.random.default_rng(1)
    u = rng.normal(size=(200, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    assert np.abs(shapes.basis_matrix(lmax, u) @ a - shapes.sh_radius_np(lmax, a, u)).max() < 1e-13


def test_fit_recovers_a_band_limited_shape_from_points():
    lmax = 6
    a = shapes.random_shape(lmax, 21, amp=0.3)
    rng = np.random.default_rng(2)
    u = rng.normal(size=(3000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    centre = np.array([0.3, -1.2, 5.0])
    pts = centre + shapes.sh_radius_np(lmax, a, u)[:, None] * u
    fit, c, rms = shapes.fit_points(pts, lmax, centre=centre)
    assert rms < 1e-13 and np.abs(fit - a).max() < 1e-12
    # unknown centre: the mean of the points is close to it, the fit absorbs the offset into the n = 1 terms
    fit2, c2, rms2 = shapes.fit_points(pts, lmax + 2)
    assert np.linalg.norm(c2 - centre) < 0.1 and rms2 < 2e-3
    # noisy scan of an ellipsoid, few points, with smoothing: close to the projection of the exact shape
    ax = np.array([1.0, 0.8, 0.6])
    v = rng.normal(size=(400, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    r = 1.0 / np.sqrt(((v / ax) ** 2).sum(1))
    pts = (r * (1 + 0.01 * rng.normal(size=400)))[:, None] * v
    fit3, _, rms3 = shapes.fit_points(pts, 4, centre=np.zeros(3), ridge=1e-4)
    ref = shapes.ellipsoid(*ax, lmax=4)
    w = rng.normal(size=(500, 3)); w /= np.linalg.norm(w, axis=1, keepdims=True)
    assert np.abs(shapes.sh_radius_np(4, fit3, w) - shapes.sh_radius_np(4, ref, w)).max() < 0.02 and rms3 < 0.02
