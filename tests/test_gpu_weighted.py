"""GPU parity of the optional weighted cap rule (docs/SPEC.md §2.8): the WEIGHTED kernel variant against the
oracle in weighted mode, through the C ABI (`shpair_set_option("rule", 1)`)."""
import numpy as np
import pytest

from common import make_case, coeff_tables, oracle_compute, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9


@pytest.fixture()
def weighted(oracle):
    oracle.set_rule("weighted")
    yield oracle
    oracle.set_rule("sharp")


def make_ctx(case, nq, K, E):
    from shpair import ShPair
    sp = ShPair(0)
    sp.settings(nq)
    nt = K.shape[0] - 1
    sp.set_ntypes(nt, len(case["shapes"]))
    for s, a in enumerate(case["shapes"]):
        sp.set_shape(s, case["lmax"], a)
    for i in range(1, nt + 1):
        for j in range(1, nt + 1):
            sp.coeff(i, j, K[i, j], E[i, j])
    sp.set_neighbors_csr(case["ilist"], case["offsets"], case["jlist"])
    sp.set_option("rule", 1)
    return sp


@pytest.mark.parametrize("lmax,nq,expo", [(6, 16, 1.25), (4, 10, 1.0), (0, 8, 1.5), (1, 1, 1.25), (3, 2, 1.25), (6, 5, 1.25),
                                          (12, 32, 1.25), (9, 12, 1.0), (6, 31, 1.25)])
def test_weighted_bed_matches_oracle(weighted, lmax, nq, expo):
    """Ring lengths that divide 64 (n_q = 1, 2, 8, 16, 32) and that do not (5, 10, 12, 31: rings straddle
    slabs, so the azimuth wrap and the ring neighbour reach into the previous and the next slab)."""
    import torch
    n = 150 if nq >= 31 else 260
    case = make_case(n, lmax, 2, seed=70 + lmax + nq, rmax_fn=weighted.shape_rmax)
    K, E = coeff_tables(1, 900.0, expo)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    out = torch.zeros(case["jlist"].size, 7, dtype=torch.float64, device="cuda")
    sp.set_pair_output(out.data_ptr())
    sp.set_option("count", 1)
    f, tq, eng, vir = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    o = oracle_compute(weighted, case, nq, K, E, eflag=True, vflag=True, want_pairs=True, nthreads=8)
    assert o["counts"][2] > 30
    fs = np.abs(o["f"]).max()
    assert rel_err(f, o["f"], fs) < TOL
    assert rel_err(tq, o["torque"], max(fs, np.abs(o["torque"]).max())) < TOL
    assert abs(eng - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    assert np.abs(vir - o["eng_virial"][1:]).max() < TOL * np.abs(o["eng_virial"][1:]).max()
    pr = out.cpu().numpy()
    assert np.abs(pr - o["pairs"]).max() < 1e-10 * np.abs(o["pairs"]).max()
    st = sp.stats()
    assert [st["n_candidates"], st["n_contact"], st["n_touching"]] == o["counts"].tolist()
    # and it is a different rule: the sharp result differs at the percent level
    sp.set_option("rule", 0)
    f0, _, _, _ = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"])
    if nq >= 5:                                       # (n_q <= 2: both azimuth neighbours are the same node, w = [g < 0])
        assert rel_err(f0, o["f"], fs) > 1e-4
    sp.close()


def test_weighted_pair_soup_all_cap_branches(weighted):
    """Isolated random pairs from grazing to deep (centre of i inside j's ball, cap = full sphere)."""
    import torch
    from shpair import shapes, ShPair
    rng = np.random.default_rng(77)
    lmax, nq, npair = 5, 9, 400
    shp = [shapes.random_shape(lmax, 90 + s, amp=0.25) for s in range(2)]
    rmax = [weighted.shape_rmax(lmax, a) for a in shp]
    n = 2 * npair
    x = np.zeros((n, 3))
    sht = rng.integers(0, 2, n).astype(np.int32)
    for p in range(npair):
        x[2 * p] = [10.0 * p, 0, 0]
        dirn = rng.normal(size=3); dirn /= np.linalg.norm(dirn)
        rho = rng.uniform(0.15, 1.02) * (rmax[sht[2 * p]] + rmax[sht[2 * p + 1]])
        x[2 * p + 1] = x[2 * p] + rho * dirn
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    il = np.arange(0, n, 2, dtype=np.int32)
    of = np.arange(npair + 1, dtype=np.int32)
    jl = np.arange(1, n, 2, dtype=np.int32)
    K, E = coeff_tables(1, 500.0, 1.25)
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 2)
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
    sp.coeff(1, 1, 500.0, 1.25)
    sp.set_neighbors_csr(il, of, jl)
    sp.set_option("rule", 1)
    out = torch.zeros(npair, 7, dtype=torch.float64, device="cuda")
    sp.set_pair_output(out.data_ptr())
    ty = np.ones(n, dtype=np.int32)
    f, tq, eng, _ = sp.compute(n, x, q, ty, sht, eflag=True)
    o = weighted.compute([(lmax, a, r) for a, r in zip(shp, rmax)], K, E, nq, n, x, q, ty, sht, il, of, jl, eflag=True,
                         want_pairs=True, nthreads=8)
    pr = out.cpu().numpy()
    sc = np.abs(o["pairs"]).max(0)
    assert (np.abs(pr - o["pairs"]) / sc).max() < 1e-10
    assert rel_err(f, o["f"]) < TOL and abs(eng - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    assert (o["pairs"][:, 0] > 0).sum() > 200 and (o["pairs"][:, 0] == 0).sum() > 10
    sp.close()


def test_weighted_rule_limits_are_refused(weighted):
    from shpair.capi import ShPairError
    case = make_case(40, 13, 1, seed=72, rmax_fn=weighted.shape_rmax)
    K, E = coeff_tables(1, 900.0, 1.0)
    sp = make_ctx(case, 8, K, E)                       # lmax 13: run-time-order kernel only
    b = case["bed"]
    with pytest.raises(ShPairError) as e:
        sp.compute(40, b["x"], b["quat"], b["type"], b["shtype"])
    assert e.value.code == -6
    sp.close()
    case = make_case(40, 4, 1, seed=73, rmax_fn=weighted.shape_rmax)
    sp = make_ctx(case, 33, K, E)                      # n_q 33: a ring neighbour would be two slabs away
    with pytest.raises(ShPairError) as e:
        sp.compute(40, case["bed"]["x"], case["bed"]["quat"], case["bed"]["type"], case["bed"]["shtype"])
    assert e.value.code == -6
    with pytest.raises(ShPairError):
        sp.set_option("rule", 2)
    sp.set_option("rule", 0)                           # the sharp rule has no such limit
    sp.compute(40, case["bed"]["x"], case["bed"]["quat"], case["bed"]["type"], case["bed"]["shtype"])
    sp.close()


@pytest.mark.parametrize("lmax,nq,rows", [(6, 16, 6), (6, 12, 8), (4, 10, 9), (12, 32, 0), (6, 31, 5), (8, 24, 1)])
def test_weighted_rule_with_ring_groups(weighted, lmax, nq, rows):
    """Ring tables resident a group at a time (forced small, or the library's own 8 KB policy for rows = 0): the
    slab that is still unweighed when a group ends is carried in registers into the next group, whose tables start
    at its first ring.  Same results as with all rings resident."""
    n = 120 if nq >= 24 else 200
    case = make_case(n, lmax, 2, seed=170 + lmax + nq, rmax_fn=weighted.shape_rmax)
    K, E = coeff_tables(1, 900.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    if rows:
        sp.set_option("ring_rows", rows)           # clamped up to the minimum the window needs
    b = case["bed"]
    f, tq, eng, _ = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(weighted, case, nq, K, E, eflag=True, nthreads=8)
    fs = np.abs(o["f"]).max()
    assert o["counts"][2] > 30
    assert rel_err(f, o["f"], fs) < TOL
    assert rel_err(tq, o["torque"], max(fs, np.abs(o["torque"]).max())) < TOL
    assert abs(eng - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    sp.close()
