"""N>1 path on CPU: world_size-2 and -4 `gloo` runs of the domain decomposition and
halo exchange, with the oracle standing in for the GPU compute (tests may call the
oracle; the product path never does).  The decomposed result must equal the
single-domain result atom for atom."""
import os
import socket
import sys

import numpy as np
import pytest

from common import make_case, coeff_tables, oracle_compute

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, grid, out_dir):
    for p in (os.path.join(ROOT, "lammps-spherharm_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    from shpair.halo import Decomposition, HaloExchange
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        case = make_case(400, 4, 2, seed=21, rmax_fn=O.shape_rmax)
        K, E = coeff_tables(1, 1000.0, 1.25)
        b = case["bed"]
        dec = Decomposition(b["x"], b["shtype"], case["rmax"], grid)
        view = dec.plan(rank)
        gid, nlocal = view["gid"], view["nlocal"]
        il, of, jl = dec.neighbor_list(view, balanced=(world == 4))   # world 4 also covers rows whose i is a ghost
        halo = HaloExchange(view, torch.device("cpu"), dist)
        # owners hold the truth; ghosts start as garbage and must be filled by forward()
        x = torch.from_numpy(b["x"][gid].copy())
        q = torch.from_numpy(b["quat"][gid].copy())
        x[nlocal:] = 1e9
        q[nlocal:] = 0.0
        halo.forward(x, q)
        assert torch.equal(x, torch.from_numpy(b["x"][gid])) and torch.equal(q, torch.from_numpy(b["quat"][gid]))
        o = O.compute([(case["lmax"], a, r) for a, r in zip(case["shapes"], case["rmax"])], K, E, 8, nlocal,
                      x.numpy(), q.numpy(), b["type"][gid], b["shtype"][gid], il, of, jl, newton_pair=True,
                      eflag=True)
        f = torch.from_numpy(o["f"].copy())
        tq = torch.from_numpy(o["torque"].copy())
        halo.reverse(f, tq)
        e = torch.tensor([o["eng_virial"][0], float(o["counts"][0])], dtype=torch.float64)
        dist.all_reduce(e)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), gid=gid[:nlocal], f=f.numpy()[:nlocal],
                 tq=tq.numpy()[:nlocal], e=e.numpy(), nghost=gid.size - nlocal)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,grid", [(2, (2, 1, 1)), (4, (2, 2, 1))])
def test_decomposed_forces_equal_single_domain(oracle, tmp_path, world, grid):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, grid, str(tmp_path)), nprocs=world, join=True)
    case = make_case(400, 4, 2, seed=21, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    ref = oracle_compute(oracle, case, 8, K, E, eflag=True)
    f = np.zeros_like(ref["f"])
    tq = np.zeros_like(ref["torque"])
    seen = np.zeros(case["n"], dtype=int)
    for r in range(world):
        d = np.load(tmp_path / f"r{r}.npz")
        f[d["gid"]] = d["f"]
        tq[d["gid"]] = d["tq"]
        seen[d["gid"]] += 1
        assert d["nghost"] > 0
        assert abs(d["e"][0] - ref["eng_virial"][0]) < 1e-11 * ref["eng_virial"][0]
        assert int(d["e"][1]) == ref["counts"][0]  # every global pair evaluated exactly once
    assert np.all(seen == 1)
    fs = np.abs(ref["f"]).max()
    assert np.abs(f - ref["f"]).max() < 1e-12 * fs and np.abs(tq - ref["torque"]).max() < 1e-12 * fs


def test_proc_grid():
    from shpair.halo import proc_grid
    assert proc_grid(1) == (1, 1, 1) and proc_grid(2) == (2, 1, 1) and proc_grid(4) == (2, 2, 1)
    assert proc_grid(8) == (2, 2, 2) and proc_grid(6) == (3, 2, 1)
