"""Gravity-settles BASELINE.json configs[0] on the GPU: 1000 identical L_max = 4 ellipsoid-like SH
particles rained onto a frozen floor layer in a box periodic in x and y, with viscous damping, through
the device-resident loop (shpair.run.DeviceRun).  Writes the settled state to an .npz; the expected
forces of the committed fixture are then computed by the CPU oracle (tests/golden/make_settled.py) —
this script only produces INPUTS (positions and orientations).

  python tools/settle.py gpurun_out/settled_cfg1.npz [sharp|weighted]
(the committed fixture was made with the sharp rule; `weighted` is for comparing how quietly the bed rests)
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import ShPair, shapes, bed  # noqa: E402
from shpair.run import DeviceRun  # noqa: E402

LMAX, NQ, KN, EXPO = 4, 10, 1000.0, 1.25
G = (0.0, 0.0, -10.0)


def main(out, rule="sharp"):
    rng = np.random.default_rng(bed.SEED0)
    shp = shapes.ellipsoid(1.0, 0.8, 0.6, LMAX)
    sp = ShPair(0)
    sp.settings(NQ)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, LMAX, shp)
    sp.coeff(1, 1, KN, EXPO)
    sp.set_option("rule", 1 if rule == "weighted" else 0)
    nf, fsp = 20, 1.3                       # floor: 20 x 20 frozen particles
    box = nf * fsp
    gx, gy = np.meshgrid(np.arange(nf), np.arange(nf), indexing="ij")
    floor = np.stack([(gx.ravel() + 0.5) * fsp, (gy.ravel() + 0.5) * fsp, np.zeros(nf * nf)], 1)
    nm, msp = 10, 2.6                       # mobile: 10 layers of 10 x 10
    ix, iy, iz = np.meshgrid(np.arange(nm), np.arange(nm), np.arange(10), indexing="ij")
    mob = np.stack([(ix.ravel() + 0.5) * msp, (iy.ravel() + 0.5) * msp, 2.0 + iz.ravel() * msp], 1)
    mob[:, :2] += rng.uniform(-0.3, 0.3, (mob.shape[0], 2))
    x = np.concatenate([mob, floor])
    n = x.shape[0]
    quat = bed.random_quaternions(n, rng)
    mask = np.concatenate([np.ones(mob.shape[0], np.int32), np.full(floor.shape[0], 2, np.int32)])
    lo, hi = np.array([0.0, 0.0, -2.0]), np.array([box, box, 40.0])
    run = DeviceRun(sp, x, quat, np.zeros(n, np.int32), lo, hi, (1, 1, 0), 0.3, dt=1e-3, gravity=G, gamma_t=4.0,
                    gamma_r=0.5, mask=mask, groupbit=1, ghost_factor=3.0)
    m = sp.body(0)[0]
    weight = m * abs(G[2])
    hist = []
    t0 = time.time()
    for blk in range(60):
        run.run(1000)
        run.force(eflag=True)
        pe, kt, kr, gpe = run.energies()
        fm = run.f[:mob.shape[0]].norm(dim=1).max().item()
        zmax = run.x[:mob.shape[0], 2].max().item()
        hist.append((run.steps, pe, kt, kr, gpe, fm, zmax))
        print(f"step {run.steps}: contact {pe:.4f} ke {kt:.3e}+{kr:.3e} gpe {gpe:.2f} max|F_net|/weight {fm / weight:.2e} "
              f"top {zmax:.2f} rebuilds {run.builds} ({time.time() - t0:.0f} s)", flush=True)
        if kt + kr < 1e-10 * mob.shape[0] * weight and fm < 1e-4 * weight:
            break
    torch.cuda.synchronize()
    nmob = mob.shape[0]
    np.savez_compressed(out, lmax=LMAX, nq=NQ, kn=KN, exponent=EXPO, anm=shp[None], gravity=np.array(G), lo=lo, hi=hi,
                        periodic=np.array([1, 1, 0]), x=run.x[:n].cpu().numpy(), quat=run.q[:n].cpu().numpy(),
                        mask=mask, nmobile=nmob, steps=run.steps, history=np.array(hist), mass=m)
    sp.close()


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/settled_cfg1.npz", sys.argv[2] if len(sys.argv) > 2 else "sharp")
