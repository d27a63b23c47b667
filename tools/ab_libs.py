"""A/B timing of several builds of libshpair on the bench bed, interleaved in one process.
usage: python tools/ab_libs.py lib1.so lib2.so ... [--lmax 6 --nq 16 --expo 1.25 --rounds 5]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
import torch  # noqa: E402
from shpair import capi, shapes, bed  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--lmax", type=int, default=6)
ap.add_argument("--nq", type=int, default=16)
ap.add_argument("--expo", type=float, default=1.25)
ap.add_argument("--n", type=int, default=100000)
ap.add_argument("--rounds", type=int, default=8)
ap.add_argument("--reps", type=int, default=4, help="launches per sample")
ap.add_argument("--rule", type=int, default=0, help="0 sharp, 1 weighted (docs/SPEC.md §2.8)")
ap.add_argument("--wpb", type=int, nargs="*", default=[0], help="waves per workgroup per lib (0 = default)")
ap.add_argument("--ring-rows", type=int, nargs="*", default=[0], help="one value per lib (0 = library default)")
ap.add_argument("--jpoly", type=int, nargs="*", default=[-1], help="one value per lib: the 'jpoly' option (-1 = leave the default)")
ap.add_argument("--det", type=int, nargs="*", default=[0], help="one value per lib: the 'deterministic' option")
ap.add_argument("--split", type=int, nargs="*", default=[-1], help="one value per lib: the 'split' option (two waves per pair; -1 = leave the default)")
a = ap.parse_args()

ctxs = []
for lib in a.libs:
    capi._LIB = None
    capi.library_path = (lambda p: (lambda: p))(os.path.join(ROOT, "lammps-spherharm_amd", "shpair", lib))
    ctxs.append(capi.ShPair(0))
shape = shapes.random_shape(a.lmax, bed.SEED0 + 2)
rmax = None
b = None
rr = (a.ring_rows * len(ctxs))[:len(ctxs)] if len(a.ring_rows) == 1 else a.ring_rows
wp = (a.wpb * len(ctxs))[:len(ctxs)] if len(a.wpb) == 1 else a.wpb
jp = (a.jpoly * len(ctxs))[:len(ctxs)] if len(a.jpoly) == 1 else a.jpoly
lp = [0] * len(ctxs)
spl = (a.split * len(ctxs))[:len(ctxs)] if len(a.split) == 1 else a.split
det = (a.det * len(ctxs))[:len(ctxs)] if len(a.det) == 1 else a.det
for sp, dv in zip(ctxs, det):
    sp.set_option("deterministic", dv)
for sp, rows, w, j, pad, sv in zip(ctxs, rr, wp, jp, lp, spl):
    if j >= 0:
        sp.set_option("jpoly", j)
    if sv >= 0:
        sp.set_option("split", sv)
    sp.set_option("ring_rows", rows)
    sp.set_option("waves_per_block", w)
    sp.settings(a.nq)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, a.lmax, shape)
    sp.coeff("*", "*", 1000.0, a.expo)
    if rmax is None:
        rmax = [sp.rmax(0)]
        b = bed.make_bed(a.n, rmax, seed=bed.SEED0 + 2)
        il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
    sp.set_neighbors_csr(il, of, jl)
    if a.rule:
        sp.set_option("rule", a.rule)
    sp.set_option("timing", 1)
dev = torch.device("cuda:0")
x = torch.from_numpy(b["x"]).to(dev)
q = torch.from_numpy(b["quat"]).to(dev)
ty = torch.from_numpy(b["type"]).to(dev)
sh = torch.from_numpy(b["shtype"]).to(dev)
f = torch.zeros(a.n, 3, dtype=torch.float64, device=dev)
tq = torch.zeros_like(f)
a.libs = [f"{lib}@{i}#{rows}w{w}j{j}p{pad}s{sv}d{dv}" for i, (lib, rows, w, j, pad, sv, dv) in enumerate(zip(a.libs, rr, wp, jp, lp, spl, det))]   # @i: the context's position (its buffers' place in memory can be worth a few per cent: a null A/B shows it)
res = {lib: [] for lib in a.libs}
fref = None
for r in range(a.rounds + 1):
    # the order of the builds alternates from round to round (a build that always runs first, or always after the same
    # neighbour, sees a different clock: measured bias of a fixed order ~2 % over 5-7 single-launch rounds), and a
    # sample is the mean of --reps launches back to back
    order = list(zip(a.libs, ctxs))
    if r % 2 == 1:
        order.reverse()
    for lib, sp in order:
        ks = []
        for _ in range(a.reps if r > 0 else 1):
            f.zero_()
            tq.zero_()
            sp.compute_device(a.n, 0, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(),
                              tq.data_ptr())
            torch.cuda.synchronize()
            ks.append(sp.stats()["kernel_ms"])
        if r > 0:
            res[lib].append(float(np.mean(ks)))
        else:  # first round: all builds must agree on the forces
            fh = f.cpu().numpy()
            if fref is None:
                fref = fh
            else:
                print(f"{lib}: max |f - f[{a.libs[0]}]| / max|f| = {np.abs(fh - fref).max() / np.abs(fref).max():.2e}")
for lib in dict.fromkeys(a.libs):
    v = np.array(res[lib])
    print(f"{lib}: median {np.median(v):.3f} ms  min {v.min():.3f}  pairs/s {jl.size / np.median(v) * 1e3:.3e}")
