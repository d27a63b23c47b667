"""Would two pairs per wave as half-waves (32 lanes each) pay at small n_q?  A measurement-based estimate BEFORE writing
the kernel: the diagnostic SHP_STATS build records the number of inside nodes of every list slot;
from that distribution this script counts the phase-2 batches a wave runs today (one pair per wave, batches of 64
queued nodes) and would run with two consecutive slots sharing a wave (half-batches of 32; the wave runs as many
as the FULLER of its two halves needs), and the same for the slabs of phase 1.

  make -C lammps-spherharm_amd/csrc stats && python tools/halfwave_sim.py [n] [lmax] [nq]
"""
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
lmax = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 10
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
npairs = jl.size
dbg = torch.zeros(64 + npairs, dtype=torch.int64, device="cuda")
dbg[15] = 1
lib = capi.load_library()
lib.shpair_debug_set_counters.argtypes = [C.c_void_p, C.c_void_p]
lib.shpair_debug_set_counters(sp._h, dbg.data_ptr())
sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"])
d = dbg.cpu().numpy()
cnt = d[64:].astype(np.int64)
assert cnt.sum() == d[3], (cnt.sum(), d[3])
Q = 2 * nq * nq
print(f"== half-wave estimate, n {n} L {lmax} nq {nq}: {npairs} slots, Q {Q}; inside nodes per slot: mean {cnt.mean():.1f}, "
      f"median {np.median(cnt):.0f}, 90 % {np.percentile(cnt, 90):.0f}, max {cnt.max()}; slots without any {np.mean(cnt == 0):.3f}")
print(f"   measured today: {d[4] / npairs:.3f} batches per slot at lane fill {d[7] / max(1, 64 * d[4]):.3f}, root-loop fill "
      f"{d[6] / max(1, 64 * d[5]):.3f}, {d[5] / max(1, d[4]):.2f} wave iterations per batch")
full = np.ceil(cnt / 64.0)
print(f"   model, one slot per wave: {full.sum() / npairs:.3f} batches per slot at fill {cnt.sum() / (64 * full.sum()):.3f}")
m = npairs // 2 * 2
c2 = cnt[:m].reshape(-1, 2)
h = np.ceil(c2 / 32.0)
wave = h.max(axis=1)
print(f"   model, two consecutive slots per wave (half-batches of 32): {wave.sum() / m:.3f} wave-batches per slot at fill "
      f"{c2.sum() / (64 * wave.sum()):.3f}  ->  x{wave.sum() / full[:m].sum():.3f} of today's batches "
      f"(if the halves never waited for each other: x{h.sum() / 2 / full[:m].sum():.3f})")
# sorted by count (an upper bound of what any pairing of slots could reach: equal neighbours never wait)
cs = np.sort(cnt[:m]).reshape(-1, 2)
ws = np.ceil(cs / 32.0).max(axis=1)
print(f"   ... with slots paired by equal count (upper bound of any pairing): x{ws.sum() / full[:m].sum():.3f}")
npp = nq * nq     # node PAIRS per slot (per-azimuth kernels: a lane takes (k, l) and (k, l + nq))
print(f"   phase 1: {int(np.ceil(npp / 64))} slabs of 64 node pairs per slot today, {int(np.ceil(npp / 32))} half-slabs = "
      f"{np.ceil(npp / 32) / 2:.1f} wave-slabs per slot as half-waves")
