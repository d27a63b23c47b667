"""Soak run of the device-resident loop (C++ driver) on the periodic 100k bench bed: energy and momentum
book-keeping over thousands of steps with rebuilds (diagnostic, GPU)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import ShPair, shapes, bed  # noqa: E402
from shpair.run import DeviceRun  # noqa: E402

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sp = ShPair(0)
sp.settings(16)
sp.set_ntypes(1, 1)
sp.set_shape(0, 6, shapes.random_shape(6, bed.SEED0 + 2))
sp.coeff(1, 1, 1000.0, 1.25)
pts, lo, hi = bed.periodic_hcp(100000, 1.9, (1, 1, 1))
rng = np.random.default_rng(1)
n = pts.shape[0]
run = DeviceRun(sp, pts + rng.uniform(-0.04, 0.04, pts.shape), bed.random_quaternions(n, rng), np.zeros(n, np.int32), lo, hi,
                (1, 1, 1), 0.1, dt=1e-3)
run.force(eflag=True)
e0 = run.energies()
m = sp.body(0)[0]
t0 = time.time()
for blk in range(nsteps // 500):
    run.run_native(500)
    run.force(eflag=True)
    e = run.energies()
    p = (m * run.v.sum(0)).cpu().numpy()
    print(f"step {run.steps}: contact {e[0]:.1f} ke {e[1]:.1f}+{e[2]:.1f} total {sum(e[:3]):.1f} (start {sum(e0[:3]):.1f}) |p| {np.abs(p).max():.2e} "
          f"rebuilds {run.builds} ghosts {run.nghost} finite {bool(torch.isfinite(run.x[:n]).all())} ({time.time() - t0:.0f} s)", flush=True)
sp.close()
