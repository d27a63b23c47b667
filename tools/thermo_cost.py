"""Cost of a thermo step: the headline compute with eflag = vflag = 1 (per-slot tally rows + the ordered reduce,
csrc/det_kernels.hpp) against the plain call, interleaved in one process on the bench bed.
  python tools/thermo_cost.py [--lmax 6 --nq 16 --n 100000 --rounds 10]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
import torch  # noqa: E402
from shpair import ShPair, shapes, bed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lmax", type=int, default=6)
ap.add_argument("--nq", type=int, default=16)
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--rounds", type=int, default=10)
ap.add_argument("--reps", type=int, default=4)
a = ap.parse_args()

sp = ShPair(0)
sp.settings(a.nq)
sp.set_ntypes(1, 1)
sp.set_shape(0, a.lmax, shapes.random_shape(a.lmax, bed.SEED0 + 2))
sp.coeff("*", "*", 1000.0, 1.25)
rmax = [sp.rmax(0)]
b = bed.make_bed(a.n, rmax, seed=bed.SEED0 + 2)
il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
sp.set_neighbors_csr(il, of, jl)
sp.set_option("timing", 1)
dev = torch.device("cuda:0")
x, q = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["quat"]).to(dev)
ty, sh = torch.from_numpy(b["type"]).to(dev), torch.from_numpy(b["shtype"]).to(dev)
f = torch.zeros(a.n, 3, dtype=torch.float64, device=dev)
tq = torch.zeros_like(f)
ev = torch.zeros(7, dtype=torch.float64, device=dev)
res = {"plain": [], "eflag+vflag": []}
evs = []
for r in range(a.rounds + 2):
    order = ["plain", "eflag+vflag"] if r % 2 == 0 else ["eflag+vflag", "plain"]
    for which in order:
        ks = []
        for _ in range(a.reps):
            f.zero_()
            tq.zero_()
            ev.zero_()
            on = which != "plain"
            sp.compute_device(a.n, 0, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(), tq.data_ptr(),
                              eflag=on, vflag=on, ev=ev.data_ptr())
            torch.cuda.synchronize()
            ks.append(sp.stats()["kernel_ms"])
            if on:
                evs.append(ev.cpu().numpy().copy())
        if r >= 2:
            res[which].append(float(np.mean(ks)))
p, t = np.median(res["plain"]), np.median(res["eflag+vflag"])
print(f"L={a.lmax} nq={a.nq} n={a.n} pairs={jl.size}: plain {p:.4f} ms, eflag+vflag {t:.4f} ms, thermo step costs {100 * (t / p - 1):+.2f} %")
print("tallies bitwise equal over", len(evs), "calls:", all(np.array_equal(e, evs[0]) for e in evs), " E =", evs[0][0])
