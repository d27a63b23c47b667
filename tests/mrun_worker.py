"""Worker of tests/test_gpu_mrun.py (launched with torch.distributed.run, backend gloo, all ranks on GPU 0):
a periodic bed stepped by shpair.mrun.MultiRankRun on a px x py x pz grid, then gathered on rank 0 and compared
with the same bed stepped by the single-rank shpair.run.DeviceRun."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from shpair import ShPair, shapes, bed  # noqa: E402
from shpair.mrun import MultiRankRun  # noqa: E402
from shpair.run import DeviceRun  # noqa: E402


def ctx(lmax, shp, nq, rule):
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, len(shp))
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
    sp.coeff(1, 1, 400.0, 1.25)
    sp.set_option("rule", rule)
    return sp


def main():
    grid = tuple(int(v) for v in sys.argv[1].split("x"))
    periodic = tuple(int(v) for v in sys.argv[2])
    nsteps = int(sys.argv[3])
    rule = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lmax, nq, skin, dt = 4, 8, 0.2, 2e-3
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    pts, lo, hi = bed.periodic_hcp(1500, 1.9, periodic)
    rng = np.random.default_rng(9)
    n = pts.shape[0]
    x = pts + rng.uniform(-0.15, 0.15, pts.shape)
    quat = bed.random_quaternions(n, rng)
    sht = rng.integers(0, 2, n).astype(np.int32)
    v0 = 2.0 * rng.normal(size=(n, 3))           # fast enough that atoms cross brick boundaries within the run
    tag = np.arange(n, dtype=np.int32)
    mask = np.where(tag % 11 == 0, 2, 1).astype(np.int32)      # every 11th particle is frozen (another group)
    v0[mask == 2] = 0.0
    # this rank's atoms (after a periodic wrap, as rebuild() would do)
    xw = x.copy()
    for d in range(3):
        if periodic[d]:
            xw[:, d] = lo[d] + np.mod(xw[:, d] - lo[d], hi[d] - lo[d])
    blen = (hi - lo) / np.array(grid)
    c = np.minimum(((xw - lo) / blen).astype(int), np.array(grid) - 1)
    owner = (c[:, 0] * grid[1] + c[:, 1]) * grid[2] + c[:, 2]
    mine = owner == rank
    sp = ctx(lmax, shp, nq, rule)
    run = MultiRankRun(sp, dist, rank, world, grid, lo, hi, periodic, skin, x[mine], quat[mine], sht[mine], tag[mine], v=v0[mine],
                       mask=mask[mine], dt=dt, gravity=(0.0, 0.0, -0.5 if not periodic[2] else 0.0), gamma_t=0.05, gamma_r=0.02, staged=True)
    n0 = run.n
    run.run(nsteps)
    torch.cuda.synchronize()
    mine_out = run.owned() + (run.builds, run.migrated, n0, run.n, run.nghost)
    parts = [None] * world if rank == 0 else None
    dist.gather_object(mine_out, parts, dst=0)
    if rank == 0:
        tg = np.concatenate([p[0] for p in parts]); o = np.argsort(tg)
        X = np.concatenate([p[1] for p in parts])[o]; V = np.concatenate([p[2] for p in parts])[o]; Q = np.concatenate([p[3] for p in parts])[o]
        assert np.array_equal(tg[o], np.arange(n)), "atoms lost or duplicated"
        sp1 = ctx(lmax, shp, nq, rule)
        ref = DeviceRun(sp1, x, quat, sht, lo, hi, periodic, skin, mask=mask, dt=dt, gravity=(0.0, 0.0, -0.5 if not periodic[2] else 0.0),
                        gamma_t=0.05, gamma_r=0.02)
        ref.v[:] = torch.from_numpy(v0).to(ref.v.device)
        ref.force()                                  # the damping term of the initial forces needs the velocities
        ref.run(nsteps)
        torch.cuda.synchronize()
        xr, vr, qr = ref.x[:n].cpu().numpy(), ref.v.cpu().numpy(), ref.q[:n].cpu().numpy()
        # compare positions modulo the periodic box
        dx = X - xr
        for d in range(3):
            if periodic[d]:
                dx[:, d] -= (hi[d] - lo[d]) * np.round(dx[:, d] / (hi[d] - lo[d]))
        print(json.dumps({"n": int(n), "dx": float(np.abs(dx).max()), "dv": float(np.abs(V - vr).max() / np.abs(vr).max()),
                          "dq": float(np.abs(np.abs((Q * qr).sum(1)) - 1).max()), "builds": [int(p[4]) for p in parts],
                          "migrated": int(sum(p[5] for p in parts)), "ref_builds": int(ref.builds),
                          "owned_start": [int(p[6]) for p in parts], "owned_end": [int(p[7]) for p in parts],
                          "ghosts": [int(p[8]) for p in parts], "vmax": float(np.abs(vr).max())}), flush=True)
        sp1.close()
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
