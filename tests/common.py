"""Shared builders for the bed-level tests (CPU and GPU)."""
import numpy as np

from shpair import shapes, bed


def make_case(n, lmax, nshapes, seed=0, amp=0.1, spacing=1.9, ntypes=1, skin=0.1, rmax_fn=None):
    """Synthetic bed + half list. rmax_fn(lmax, anm) -> bounding radius (oracle or library helper)."""
    shp = [shapes.random_shape(lmax, 1000 * seed + 17 * s + lmax, amp=amp) for s in range(nshapes)]
    rmax = [rmax_fn(lmax, a) for a in shp]
    b = bed.make_bed(n, rmax, nshapes, spacing=spacing, seed=bed.SEED0 + seed)
    if ntypes > 1:
        b["type"] = (1 + np.arange(n) % ntypes).astype(np.int32)
    il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax, skin=skin)
    return dict(lmax=lmax, shapes=shp, rmax=rmax, bed=b, ilist=il, offsets=of, jlist=jl, ntypes=ntypes, n=n)


def coeff_tables(ntypes, kn=1000.0, expo=1.0):
    """(ntypes+1)^2 tables; kn/expo scalars or symmetric functions of (i,j)."""
    K = np.zeros((ntypes + 1, ntypes + 1))
    E = np.ones((ntypes + 1, ntypes + 1))
    for i in range(1, ntypes + 1):
        for j in range(1, ntypes + 1):
            K[i, j] = kn(i, j) if callable(kn) else kn
            E[i, j] = expo(i, j) if callable(expo) else expo
    return K, E


def oracle_compute(O, case, nq, K, E, **kw):
    b = case["bed"]
    nlocal = kw.pop("nlocal", case["n"])
    return O.compute([(case["lmax"], a, r) for a, r in zip(case["shapes"], case["rmax"])], K, E, nq, nlocal,
                     b["x"], b["quat"], b["type"], b["shtype"], case["ilist"], case["offsets"], case["jlist"], **kw)


def rel_err(a, b, scale=None):
    scale = scale if scale is not None else np.abs(b).max()
    return np.abs(np.asarray(a) - np.asarray(b)).max() / scale
