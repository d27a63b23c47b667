"""Regenerates tests/golden/*.npz.

NOT reference vectors: /root/reference holds no source, tests or fixtures
(README.md only), so there is nothing to import or run.  These are outputs of
the CPU oracle (oracle/shpair_oracle.c) on small seeded beds, one per
BASELINE.json config, committed so that (a) the oracle is regression-pinned
and (b) the GPU box, which has no /root/reference and need not rebuild
anything, can check the HIP path against fixed numbers.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle as O  # noqa: E402
from shpair import shapes, bed  # noqa: E402

# name: (n, lmax, nq, nshapes, kn, exponent, shape generator)
CONFIGS = {
    "cfg1_L4_ellipsoid": (24, 4, 10, 1, 1000.0, 1.0, "ellipsoid"),
    "cfg2_L6_single": (30, 6, 16, 1, 1000.0, 1.25, "random"),
    "cfg3_L6_mixed4": (30, 6, 16, 4, 1000.0, 1.0, "random"),
    "cfg5_L12_high": (24, 12, 32, 1, 1000.0, 1.5, "random"),
    # the optional weighted cap rule (docs/SPEC.md §2.8) on the config-2 bed at a ring length that straddles slabs
    "cfg2w_L6_weighted": (30, 6, 12, 2, 1000.0, 1.25, "random"),
}


def build(name):
    n, lmax, nq, nshapes, kn, expo, gen = CONFIGS[name]
    seed = bed.SEED0 + sorted(CONFIGS).index(name)
    if gen == "ellipsoid":
        shp = [shapes.ellipsoid(1.0, 0.8, 0.6, lmax)]
        spacing = 1.45
    else:
        shp = [shapes.random_shape(lmax, seed + s) for s in range(nshapes)]
        spacing = 1.9
    rmax = [O.shape_rmax(lmax, a) for a in shp]
    b = bed.make_bed(n, rmax, nshapes, spacing=spacing, seed=seed)
    il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
    K = np.full((2, 2), kn)
    E = np.full((2, 2), expo)
    rule = 1 if "weighted" in name else 0
    O.set_rule(rule)
    o = O.compute([(lmax, a, r) for a, r in zip(shp, rmax)], K, E, nq, n, b["x"], b["quat"], b["type"],
                  b["shtype"], il, of, jl, eflag=True, vflag=True, want_pairs=True)
    O.set_rule(0)
    assert jl.size <= 100 and o["counts"][2] > 10, (name, jl.size, o["counts"])
    return dict(rule=rule, lmax=lmax, nq=nq, kn=kn, exponent=expo, anm=np.stack(shp), rmax=np.array(rmax), x=b["x"],
                quat=b["quat"], type=b["type"], shtype=b["shtype"], ilist=il, offsets=of, jlist=jl, f=o["f"],
                torque=o["torque"], eng_virial=o["eng_virial"], counts=o["counts"], pairs=o["pairs"])


if __name__ == "__main__":
    O.build()
    for name in CONFIGS:
        d = build(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **d)
        print(name, "pairs", d["jlist"].size, "counts", d["counts"], "E", d["eng_virial"][0])
