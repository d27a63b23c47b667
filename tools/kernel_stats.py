"""Work counters of the pair kernel on the bench bed (diagnostic SHP_STATS build).
  make -C lammps-spherharm_amd/csrc stats && python tools/kernel_stats.py [n] [lmax] [nq]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
import torch  # noqa: E402,F401
from shpair import capi, shapes, bed  # noqa: E402

capi.library_path = lambda: os.path.join(ROOT, "lammps-spherharm_amd", "shpair", os.environ.get("SHP_STATS_LIB", "libshpair_stats.so"))
from shpair import ShPair  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 16
a = shapes.random_shape(lmax, bed.SEED0 + 2)
sp = ShPair(0)
sp.settings(nq)
sp.set_ntypes(1, 1)
sp.set_shape(0, lmax, a)
sp.coeff("*", "*", 1000.0, 1.25)
rmax = [sp.rmax(0)]
b = bed.make_bed(n, rmax, seed=bed.SEED0 + 2)
il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
sp.set_neighbors_csr(il, of, jl)
dbg = torch.zeros(16, dtype=torch.int64, device="cuda")
lib = capi.load_library()
lib.shpair_debug_set_counters.argtypes = [C.c_void_p, C.c_void_p]
lib.shpair_debug_set_counters(sp._h, dbg.data_ptr())
sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"])
d = dbg.cpu().numpy()
npairs = jl.size
Q = 2 * nq * nq
print(f"pairs {npairs}  Q {Q}  slabs/pair {d[0] / npairs:.2f}")
print(f"nodes in B_j: {d[1] / (npairs * Q):.3f} of all nodes; slabs with j-eval: {d[2] / d[0]:.3f}")
print(f"inside nodes: {d[3] / (npairs * Q):.3f} of all nodes; phase-2 batches per pair: {d[4] / npairs:.2f}; "
      f"lane fill of the batches: {d[7] / max(1, 64 * d[4]):.3f}")
print(f"root finder: wave iterations per batch {d[5] / max(1, d[4]):.2f}; lane evals per inside node "
      f"{d[6] / max(1, d[3]):.2f}; lane fill in root loop {d[6] / max(1, 64 * d[5]):.3f}")
print(f"root finder: general-case branch taken in {d[8] / max(1, d[5]):.3f} of the wave iterations; slabs that did not fit the queue: "
      f"{d[9] / npairs:.3f} per pair, of which run as direct batches {d[10] / npairs:.3f} (the others are classified twice)")
