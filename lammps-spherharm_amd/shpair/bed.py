"""Deterministic synthetic packed beds and half neighbour lists.

Stand-in for the reference's gravity-settled beds (no integrator is in scope):
centres on a jittered HCP lattice at `spacing` x mean radius, orientations
uniform on SO(3), LAMMPS-layout half neighbour list with cutoff
Rmax_i + Rmax_j + skin.  Setup code only; no force evaluation happens here.
"""
import numpy as np

SEED0 = 20261004


def random_quaternions(n, rng):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q


def hcp_lattice(n_target, spacing, aspect=(1.0, 1.0, 1.0)):
    """>= n_target HCP sites with nearest-neighbour distance `spacing`, trimmed to n_target."""
    dx, dy, dz = spacing, spacing * np.sqrt(3.0) / 2.0, spacing * np.sqrt(2.0 / 3.0)
    vol = n_target * dx * dy * dz
    s = (vol / (aspect[0] * aspect[1] * aspect[2])) ** (1.0 / 3.0)
    nx = max(1, int(np.ceil(aspect[0] * s / dx)))
    ny = max(1, int(np.ceil(aspect[1] * s / dy)))
    nz = max(1, int(np.ceil(aspect[2] * s / dz)))
    while nx * ny * nz < n_target:
        nz += 1
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    x = (i + 0.5 * (j % 2) + 0.5 * (k % 2)) * dx
    y = (j + (k % 2) / 3.0) * dy
    z = k * dz
    pts = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1)
    return pts[:n_target].copy()


def make_bed(n, rmax_by_shape, nshapes=1, spacing=1.9, jitter=0.04, seed=SEED0, aspect=(1.0, 1.0, 1.0)):
    """Returns dict(x, quat, type, shtype). Mean radius is 1 by construction of the shapes."""
    rng = np.random.default_rng(seed)
    x = hcp_lattice(n, spacing, aspect)
    x += rng.uniform(-jitter, jitter, size=x.shape)
    quat = random_quaternions(n, rng)
    shtype = rng.integers(0, nshapes, size=n).astype(np.int32) if nshapes > 1 else np.zeros(n, np.int32)
    type_ = np.ones(n, dtype=np.int32)
    return dict(x=np.ascontiguousarray(x), quat=np.ascontiguousarray(quat), type=type_, shtype=shtype,
                rmax=np.asarray(rmax_by_shape, dtype=np.float64))


def half_neighbor_list(x, shtype, rmax_by_shape, skin=0.1, nlocal=None, owner_rule=None, cross_rule=None):
    """LAMMPS-layout half list in CSR form: (ilist, offsets, jlist).

    Every unordered pair within Rmax_i + Rmax_j + skin appears once, stored with
    the lower index as i (newton on).  With `nlocal`, only pairs with at least
    one local atom are kept, and local-ghost pairs are filtered by ONE of
      owner_rule(i_local, j_ghost) -> bool array: keep with the local atom as i;
      cross_rule(i_local, j_ghost) -> int array: 0 drop (another rank evaluates the pair), 1 keep with the
        local atom as i, 2 keep with the GHOST as i (rows of ghost atoms then follow the owned rows; needs
        newton on: the force on the ghost travels home with the reverse exchange).
    """
    from scipy.spatial import cKDTree
    rmax = np.asarray(rmax_by_shape, dtype=np.float64)
    n = x.shape[0]
    nlocal = n if nlocal is None else nlocal
    rcut = 2.0 * rmax.max() + skin
    pairs = cKDTree(x).query_pairs(rcut, output_type="ndarray")
    a, b = pairs[:, 0], pairs[:, 1]
    d = np.linalg.norm(x[a] - x[b], axis=1)
    keep = d < rmax[shtype[a]] + rmax[shtype[b]] + skin
    a, b = a[keep], b[keep]
    # i must be local: swap pairs whose lower index is a ghost
    sw = a >= nlocal
    a, b = np.where(sw, b, a), np.where(sw, a, b)
    keep = a < nlocal
    gh = b >= nlocal
    if cross_rule is not None:
        code = np.ones(a.size, dtype=np.int64)
        sel = keep & gh
        code[sel] = cross_rule(a[sel], b[sel])
        keep &= code > 0
        flip = keep & (code == 2)
        a, b = np.where(flip, b, a), np.where(flip, a, b)
    elif owner_rule is not None:
        keep &= ~gh | owner_rule(a, b)
    a, b = a[keep], b[keep]
    order = np.lexsort((b, a))
    a, b = a[order], b[order]
    if a.size and a.max() >= nlocal:
        # rows of ghost atoms: only the non-empty ones, after the owned rows
        grow = np.unique(a[a >= nlocal])
        ilist = np.concatenate([np.arange(nlocal), grow]).astype(np.int32)
        counts = np.concatenate([np.bincount(a[a < nlocal], minlength=nlocal)[:nlocal],
                                 np.bincount(np.searchsorted(grow, a[a >= nlocal]), minlength=grow.size)])
    else:
        ilist = np.arange(nlocal, dtype=np.int32)
        counts = np.bincount(a, minlength=nlocal)[:nlocal]
    offsets = np.zeros(ilist.size + 1, dtype=np.int32)
    np.cumsum(counts, out=offsets[1:])
    return ilist, offsets, b.astype(np.int32)


def periodic_hcp(n_target, spacing, periodic=(1, 1, 1)):
    """~n_target HCP sites in a box commensurate with the lattice in the periodic directions.

    Returns (points, lo, hi).  Layers repeat every 2 in y and z, so ny and nz are even; the box spans
    exactly nx dx, ny dy, nz dz in periodic directions and leaves half a spacing of room in the others."""
    dx, dy, dz = spacing, spacing * np.sqrt(3.0) / 2.0, spacing * np.sqrt(2.0 / 3.0)
    s = (n_target * dx * dy * dz) ** (1.0 / 3.0)
    nx = max(2, int(round(s / dx)))
    ny = max(2, 2 * int(round(s / dy / 2)))
    nz = max(2, 2 * int(round(n_target / (nx * ny) / 2)))
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    x = (i + 0.5 * (j % 2) + 0.5 * (k % 2)) * dx
    y = (j + (k % 2) / 3.0) * dy
    z = k * dz
    pts = np.stack([x.ravel(), y.ravel(), z.ravel()], axis=1) + 0.25 * spacing
    ext = np.array([nx * dx, ny * dy, nz * dz])
    lo = np.zeros(3)
    per = np.array(periodic, bool)
    hi = np.where(per, ext, ext + spacing)
    pts[:, per] = np.mod(pts[:, per], ext[per])
    return pts, lo, hi
