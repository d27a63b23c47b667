"""Fuzz: many isolated random pairs (grazing to deep, random shapes and orientations), HIP vs oracle per pair
(V, S_n, T_n) and per force, for both cap rules.  Diagnostic (GPU); prints the worst deviations."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
sys.path.insert(0, ROOT)
from shpair import ShPair, shapes  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker)

npair = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
CONFIGS = ((6, 16, 1.25), (5, 9, 1.5), (3, 12, 1.0), (8, 7, 1.25), (4, 10, 1.25), (12, 32, 1.25), (6, 16, 1.0))   # (6, 16), (4, 10), (12, 32): the specialised instances
if os.environ.get("SHP_FUZZ_MORE"):   # dense slabs (direct batches), ring groups, two waves per pair, the rules' other choices
    CONFIGS += ((6, 24, 1.25), (4, 16, 1.0), (8, 20, 1.25), (12, 16, 1.25), (2, 16, 1.5), (9, 12, 1.25), (11, 22, 1.25), (10, 5, 1.0))
for lmax, nq, expo in CONFIGS:
    rng = np.random.default_rng(1000 + lmax + nq)
    shp = [shapes.random_shape(lmax, 300 + s, amp=0.25) for s in range(3)]
    rmax = [O.shape_rmax(lmax, a) for a in shp]
    n = 2 * npair
    sht = rng.integers(0, 3, n).astype(np.int32)
    x = np.zeros((n, 3))
    dirn = rng.normal(size=(npair, 3)); dirn /= np.linalg.norm(dirn, axis=1, keepdims=True)
    rsum = np.array(rmax)[sht[0::2]] + np.array(rmax)[sht[1::2]]
    rho = rng.uniform(0.1, 1.03, npair) * rsum
    x[0::2, 0] = 10.0 * np.arange(npair)
    x[1::2] = x[0::2] + rho[:, None] * dirn
    q = rng.normal(size=(n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    il = np.arange(0, n, 2, dtype=np.int32); of = np.arange(npair + 1, dtype=np.int32); jl = np.arange(1, n, 2, dtype=np.int32)
    ty = np.ones(n, dtype=np.int32)
    K = np.full((2, 2), 700.0); E = np.full((2, 2), expo)
    for rule in (0, 1):
        sp = ShPair(0)
        sp.settings(nq); sp.set_ntypes(1, 3)
        for s, a in enumerate(shp):
            sp.set_shape(s, lmax, a)
        sp.coeff(1, 1, 700.0, expo)
        sp.set_neighbors_csr(il, of, jl)
        sp.set_option("rule", rule); sp.set_option("force_volume", 1)
        out = torch.zeros(npair, 7, dtype=torch.float64, device="cuda")
        sp.set_pair_output(out.data_ptr())
        f, tq, eng, _ = sp.compute(n, x, q, ty, sht, eflag=True)
        O.set_rule(rule)
        o = O.compute([(lmax, a, r) for a, r in zip(shp, rmax)], K, E, nq, n, x, q, ty, sht, il, of, jl, eflag=True,
                      force_volume=True, want_pairs=True, nthreads=O.max_threads())
        O.set_rule(0)
        pr = out.cpu().numpy()
        sc = np.abs(o["pairs"]).max(0)
        dev = np.abs(pr - o["pairs"]) / sc
        fs = np.abs(o["f"]).max()
        df = np.abs(f - o["f"]).max(1) / fs
        bad = int((dev.max(1) > 1e-9).sum())
        print(f"L {lmax} nq {nq:2d} m {expo} rule {rule}: {npair} pairs, touching {int((o['pairs'][:, 0] > 0).sum())}; worst per-pair dev "
              f"V {dev[:, 0].max():.1e} S {dev[:, 1:4].max():.1e} T {dev[:, 4:].max():.1e}; worst force dev {df.max():.1e}; "
              f"pairs beyond 1e-9: {bad}; energy dev {abs(eng - o['eng_virial'][0]) / o['eng_virial'][0]:.1e}", flush=True)
        sp.close()
