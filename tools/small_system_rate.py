"""Timesteps/s of the launch-bound small system (BASELINE configs[0]: the settled L = 4 bed, 1400 particles)
with the three drivers of the same loop: Python call-by-call, the library's C++ loop, and the C++ loop
replayed from captured hipGraphs (shstep_run_device)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import ShPair  # noqa: E402
from shpair.run import DeviceRun  # noqa: E402

g = np.load(os.path.join(ROOT, "tests", "golden", "settled_cfg1_L4.npz"))
n = g["x"].shape[0]
NSTEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for name, kw in (("python call-by-call", None), ("C++ loop", dict(use_graph=False)), ("C++ loop + hipGraph", dict(use_graph=True)),
                 ("C++ loop + hipGraph, check every 10", dict(use_graph=True, check_every=10))):
    sp = ShPair(0)
    sp.settings(int(g["nq"]))
    sp.set_ntypes(1, 1)
    sp.set_shape(0, int(g["lmax"]), g["anm"][0])
    sp.coeff(1, 1, float(g["kn"]), float(g["exponent"]))
    run = DeviceRun(sp, g["x"], g["quat"], np.zeros(n, np.int32), g["lo"], g["hi"], g["periodic"], float(g["skin"]), dt=1e-3,
                    gravity=g["gravity"], gamma_t=4.0, gamma_r=0.5, mask=g["mask"], groupbit=1, ghost_factor=3.0)
    (run.run(200) if kw is None else run.run_native(200, **kw))
    torch.cuda.synchronize()
    b0 = run.builds
    t0 = time.perf_counter()
    (run.run(NSTEPS) if kw is None else run.run_native(NSTEPS, **kw))
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"{name:38s}: {NSTEPS / el:9.0f} timesteps/s  ({1e6 * el / NSTEPS:6.1f} us/step, {run.builds - b0} rebuilds, "
          f"top of bed {run.x[:int(g['nmobile']), 2].max().item():.3f})", flush=True)
    sp.close()
