"""Property tests of the HIP path: seeded fuzz of isolated random pairs through every cap branch of
SPEC §2.2 against the oracle (hypothesis drives the seeds, derandomised so that the GPU box runs the same cases every
time), and the exact symmetries of the discrete rule — translation, and rotation of the whole pair about its axis
by a multiple of the azimuthal node spacing — on the HIP path itself."""
import numpy as np
import pytest
from hypothesis import given, settings, strategies as st

pytestmark = pytest.mark.gpu


def _pairs(rng, npair, rmax, nshape, lo, hi):
    n = 2 * npair
    sht = rng.integers(0, nshape, n).astype(np.int32)
    x = np.zeros((n, 3))
    dirn = rng.normal(size=(npair, 3))
    dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    ri, rj = np.array(rmax)[sht[0::2]], np.array(rmax)[sht[1::2]]
    rho = rng.uniform(lo, hi, npair) * (ri + rj)
    x[0::2, 0] = 10.0 * np.arange(npair)
    x[1::2] = x[0::2] + rho[:, None] * dirn
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    il = np.arange(0, n, 2, dtype=np.int32)
    of = np.arange(npair + 1, dtype=np.int32)
    jl = np.arange(1, n, 2, dtype=np.int32)
    branch = np.where(rho <= rj, 0, np.where(rho * rho - rj * rj <= ri * ri, 1, 2))   # SPEC §2.2: cos(alpha) cases
    return x, q, sht, il, of, jl, branch


def _hip_pairs(lmax, shp, nq, expo, x, q, sht, il, of, jl, force_volume=True):
    import torch
    from shpair import ShPair
    n, npair = x.shape[0], jl.size
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, len(shp))
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
    sp.coeff(1, 1, 700.0, expo)
    sp.set_neighbors_csr(il, of, jl)
    sp.set_option("force_volume", 1 if force_volume else 0)
    out = torch.zeros(npair, 7, dtype=torch.float64, device="cuda")
    sp.set_pair_output(out.data_ptr())
    f, tq, eng, _ = sp.compute(n, x, q, np.ones(n, np.int32), sht, eflag=True)
    pr = out.cpu().numpy()
    sp.close()
    return pr, f, tq, eng


@settings(max_examples=6, deadline=None, derandomize=True)
@given(seed=st.integers(0, 10**6), lmax=st.sampled_from([3, 5, 6, 8]), nq=st.sampled_from([7, 9, 12, 16]),
       expo=st.sampled_from([1.0, 1.25, 1.5]))
def test_fuzzed_pairs_match_the_oracle_in_every_cap_branch(oracle, seed, lmax, nq, expo):
    from shpair import shapes
    rng = np.random.default_rng(seed)
    shp = [shapes.random_shape(lmax, int(rng.integers(1, 10**6)), amp=0.25) for _ in range(3)]
    rmax = [oracle.shape_rmax(lmax, a) for a in shp]
    npair = 3000
    x, q, sht, il, of, jl, branch = _pairs(rng, npair, rmax, 3, 0.1, 1.03)
    assert all((branch == b).sum() > 100 for b in (0, 1, 2))          # a few hundred pairs per branch at least
    pr, f, tq, eng = _hip_pairs(lmax, shp, nq, expo, x, q, sht, il, of, jl)
    K, E = np.full((2, 2), 700.0), np.full((2, 2), expo)
    n = x.shape[0]
    o = oracle.compute([(lmax, a, r) for a, r in zip(shp, rmax)], K, E, nq, n, x, q, np.ones(n, np.int32), sht, il, of, jl,
                       eflag=True, force_volume=True, want_pairs=True, nthreads=oracle.max_threads())
    sc = np.abs(o["pairs"]).max(0)
    dev = np.abs(pr - o["pairs"]) / sc
    fs = np.abs(o["f"]).max()
    for b in (0, 1, 2):
        assert dev[branch == b].max() < 1e-9, (b, dev[branch == b].max())
    assert np.abs(f - o["f"]).max() < 1e-9 * fs and np.abs(tq - o["torque"]).max() < 1e-9 * max(fs, np.abs(o["torque"]).max())
    assert abs(eng - o["eng_virial"][0]) < 1e-9 * o["eng_virial"][0]
    assert (o["pairs"][:, 0] > 0).sum() > 1000


def _quat_of(axis, ang):
    axis = axis / np.linalg.norm(axis, axis=-1, keepdims=True)
    return np.concatenate([np.cos(ang / 2)[..., None], np.sin(ang / 2)[..., None] * axis], axis=-1)


def _qmul(a, b):
    w1, x1, y1, z1 = a.T
    w2, x2, y2, z2 = b.T
    return np.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                     w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], axis=1)


def _rotmat(Q):
    w, x, y, z = Q.T
    return np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1),
                     np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1),
                     np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)], 1)


@pytest.mark.parametrize("lmax,nq", [(6, 16), (4, 10), (12, 12)])
def test_rotation_about_the_pair_axis_by_node_spacings_and_translation_are_exact_on_the_gpu(lmax, nq):
    """The HIP kernel's own symmetries (cf. tests/test_oracle_pair.py): a rigid rotation of both bodies about the pair
    axis by k x 2 pi / n_psi permutes the cap nodes, so V is unchanged and S_n, T_n turn with the bodies; a common
    translation changes nothing."""
    from shpair import shapes, capi
    rng = np.random.default_rng(lmax * 100 + nq)
    shp = [shapes.random_shape(lmax, 900 + s, amp=0.25) for s in range(2)]
    rmax = [capi.shape_default_rmax(lmax, a) for a in shp]
    npair = 2000
    x, q, sht, il, of, jl, _ = _pairs(rng, npair, rmax, 2, 0.55, 0.98)
    p1, f1, t1, _ = _hip_pairs(lmax, shp, nq, 1.25, x, q, sht, il, of, jl)
    # rotate pair p about its own axis through x_i by k_p node spacings
    c = x[1::2] - x[0::2]
    k = rng.integers(1, 2 * nq, npair)
    Q = _quat_of(c, 2 * np.pi * k / (2 * nq))
    q2 = q.copy()
    q2[0::2] = _qmul(Q, q[0::2])
    q2[1::2] = _qmul(Q, q[1::2])
    p2, f2, t2, _ = _hip_pairs(lmax, shp, nq, 1.25, x, q2, sht, il, of, jl)
    R = _rotmat(Q)
    sc = np.abs(p1).max(0)
    touching = p1[:, 0] > 0
    assert touching.sum() > npair // 2
    assert np.abs(p2[:, 0] - p1[:, 0]).max() < 1e-11 * sc[0]
    assert np.abs(p2[:, 1:4] - np.einsum("pab,pb->pa", R, p1[:, 1:4])).max() < 1e-11 * sc[1:4].max()
    assert np.abs(p2[:, 4:7] - np.einsum("pab,pb->pa", R, p1[:, 4:7])).max() < 1e-11 * sc[4:7].max()
    # translation by an exactly representable shift (pairs stay 10 apart: the same isolated pairs)
    p3, f3, t3, _ = _hip_pairs(lmax, shp, nq, 1.25, x + np.array([0.5, -0.25, 2.0]), q, sht, il, of, jl)
    assert np.abs(p3 - p1).max() < 1e-11 * sc.max() and np.abs(f3 - f1).max() < 1e-11 * np.abs(f1).max()
