"""N > 1 path on CPU (no GPU): the library's pure host planner (include/shhalo.h: shhalo_plan_geometry / owner /
ghost_mask / layout — the same geometry and layout code, and the same per-row decisions, the device path runs)
drives a decomposed evaluation of a periodic bed at world sizes 1, 2, 4, 8 (2x2x2 included), with the oracle
standing in for the pair kernel.  Transport: an in-process mailbox between rank threads, and real `gloo`
point-to-point between processes at world 2 and 8.  The decomposed forces must equal the single-domain forces atom
for atom, every pair must be evaluated exactly once, and no ghost row may be left unwritten."""
import os
import queue
import socket
import sys
import threading

import numpy as np
import pytest

from shpair import bed, mrank, shapes as shp_mod

import halo_host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _case(O, periodic, n_target=260, seed=5):
    lmax, nq, skin = 4, 8, 0.15
    sh = [shp_mod.random_shape(lmax, 300 + s, amp=0.2) for s in range(2)]
    shapes = [(lmax, a, O.shape_rmax(lmax, a)) for a in sh]
    pts, lo, hi = bed.periodic_hcp(n_target, 1.9, periodic)
    rng = np.random.default_rng(seed)
    n = pts.shape[0]
    x = pts + rng.uniform(-0.12, 0.12, pts.shape)
    # a few rows start outside the box: the planner wraps them (Domain::pbc)
    for d in range(3):
        if periodic[d]:
            x[::17, d] += (hi[d] - lo[d])
    quat = bed.random_quaternions(n, rng)
    sht = rng.integers(0, 2, n).astype(np.int32)
    ty = np.ones(n, dtype=np.int32)
    tag = np.arange(n, dtype=np.int32)
    cut = 2.0 * max(s[2] for s in shapes) + skin
    K = np.full((2, 2), 700.0)
    E = np.full((2, 2), 1.25)
    return dict(shapes=shapes, nq=nq, skin=skin, lo=lo, hi=hi, periodic=periodic, cut=cut, x=x, quat=quat, ty=ty, sht=sht,
                tag=tag, K=K, E=E, n=n)


def _rank_body(O, c, grid, rank, exchange):
    hr = halo_host.HostRank(rank, grid, c["lo"], c["hi"], c["periodic"], c["cut"], c["x"], c["quat"], c["ty"], c["sht"], c["tag"])
    remote = sorted({hr.geo.peer[k] for k in range(27) if k != 13 and hr.geo.peer[k] >= 0 and hr.geo.peer[k] != rank})
    inbox = {p: np.zeros(27, dtype=np.int64) for p in remote}
    exchange([(p, hr.count_message(p)) for p in remote], [(p, inbox[p]) for p in remote])
    hr.set_counts(inbox)
    f, tq, e, npairs = halo_host.run_rank(hr, O, c["shapes"], c["K"], c["E"], c["nq"], c["skin"], exchange)
    return dict(tag=hr.tag, f=f, tq=tq, e=e, npairs=npairs, nghost=hr.nghost, npeers=hr.lay.npeers)


def _check(c, O, parts):
    fr, tr, er, npr = halo_host.single_domain_reference(O, c["shapes"], c["K"], c["E"], c["nq"], c["skin"], c["lo"], c["hi"],
                                                        c["periodic"], c["cut"], c["x"], c["quat"], c["ty"], c["sht"], c["tag"])
    f = np.zeros_like(fr)
    tq = np.zeros_like(tr)
    seen = np.zeros(c["n"], dtype=int)
    for p in parts:
        f[p["tag"]] = p["f"]
        tq[p["tag"]] = p["tq"]
        seen[p["tag"]] += 1
    assert np.all(seen == 1), "an atom is owned by no rank or by two"
    assert sum(p["npairs"] for p in parts) == npr and npr > 0      # every pair evaluated exactly once
    assert abs(sum(p["e"] for p in parts) - er) < 1e-11 * abs(er)
    fs = np.abs(fr).max()
    assert fs > 0
    assert np.abs(f - fr).max() < 1e-12 * fs and np.abs(tq - tr).max() < 1e-12 * fs


@pytest.mark.parametrize("grid,periodic", [((1, 1, 1), (1, 1, 1)), ((2, 1, 1), (1, 0, 0)), ((2, 1, 1), (1, 1, 1)), ((2, 2, 1), (1, 1, 0)),
                                           ((2, 2, 2), (1, 1, 0)), ((2, 2, 2), (1, 1, 1)), ((3, 1, 1), (1, 1, 0)),
                                           ((4, 2, 1), (0, 1, 1))])
def test_decomposed_forces_equal_single_domain_threads(oracle, grid, periodic):
    world = int(np.prod(grid))
    c = _case(oracle, periodic, n_target=700 if max(grid) > 2 else 260)
    boxes = {(a, b): queue.Queue() for a in range(world) for b in range(world)}
    parts, errs = [None] * world, []

    def work(rank):
        def exchange(sends, recvs):
            for p, a in sends:
                boxes[(rank, p)].put(np.array(a, copy=True))
            for p, out in recvs:
                got = boxes[(p, rank)].get(timeout=120)
                assert got.shape == out.shape, (got.shape, out.shape)
                out[...] = got
        try:
            parts[rank] = _rank_body(oracle, c, grid, rank, exchange)
        except BaseException as e:  # noqa: BLE001 - reported below
            errs.append((rank, repr(e)))
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    if world == 8:
        assert all(p["npeers"] == 7 or periodic != (1, 1, 1) for p in parts)   # 2x2x2 periodic: 7 distinct peers each
    assert all(p["nghost"] > 0 for p in parts)
    _check(c, oracle, parts)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _gloo_worker(rank, world, port, grid, periodic, out_dir):
    for p in (os.path.join(ROOT, "lammps-spherharm_amd"), ROOT, os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        c = _case(O, periodic)

        def exchange(sends, recvs):
            ops, keep = [], []
            for p, out in recvs:          # same peer order on both sides; one message per peer and direction of travel
                t = torch.empty(out.shape, dtype=torch.from_numpy(np.zeros(1, out.dtype)).dtype)
                keep.append((t, out))
                ops.append(dist.P2POp(dist.irecv, t, p))
            for p, a in sends:
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(a)).clone(), p))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for t, out in keep:
                out[...] = t.numpy()
        r = _rank_body(O, c, grid, rank, exchange)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), **r)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("grid,periodic", [((2, 1, 1), (1, 1, 0)), ((2, 2, 2), (1, 1, 0))])
def test_decomposed_forces_equal_single_domain_gloo(oracle, tmp_path, grid, periodic):
    import torch.multiprocessing as mp
    world = int(np.prod(grid))
    mp.spawn(_gloo_worker, args=(world, _free_port(), grid, periodic, str(tmp_path)), nprocs=world, join=True)
    c = _case(oracle, periodic)
    parts = [dict(np.load(tmp_path / f"r{r}.npz")) for r in range(world)]
    _check(c, oracle, parts)


def _staged_worker(rank, world, port, out_dir):
    for p in (os.path.join(ROOT, "lammps-spherharm_amd"), ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    import ctypes as C
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = mrank.GlooStaged(dist)
        peers = [q for q in range(world) if q != rank]
        send = [np.full(1000 + 16 * q, 10 * rank + q, dtype=np.uint8) for q in peers]       # to q: the byte 10 rank + q
        recv = [np.zeros(1000 + 16 * rank, dtype=np.uint8) for q in peers]
        n = len(peers)
        pr = (C.c_int * n)(*peers)
        sp_ = (C.c_void_p * n)(*[a.ctypes.data for a in send])
        rp_ = (C.c_void_p * n)(*[a.ctypes.data for a in recv])
        sb = (C.c_size_t * n)(*[a.size for a in send])
        rb = (C.c_size_t * n)(*[a.size for a in recv])
        for _ in range(3):      # consecutive exchanges stay matched (forward, reverse, forward ...)
            rc = g.exchange_fn(None, n, pr, sp_, sb, n, pr, rp_, rb)
            assert rc == 0, g.last_error
            for q, a in zip(peers, recv):
                assert np.all(a == 10 * q + rank), (rank, q, a[:4])
                a[:] = 0
        v = np.array([rank + 1, 100 - rank], dtype=np.int32)
        assert g.allreduce_fn(None, v.ctypes.data, 2, 0) == 0 and list(v) == [world, 100]
        d = np.array([0.5 * (rank + 1)], dtype=np.float64)
        assert g.allreduce_fn(None, d.ctypes.data, 1, 1) == 0 and d[0] == 0.25 * world * (world + 1)
        open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_staged_transport_callbacks_over_gloo(tmp_path):
    """The caller's half of the host-staged transport (shhalo_create_staged: the library stages its packed buffers through
    page-locked host memory and hands them to two functions of the caller) as bench.py --transport staged provides it:
    mrank.GlooStaged's exchange and all-reduce functions, called through their C function pointers as the library calls
    them, between four real gloo processes.  (The library's half needs a GPU: tests/test_bench_contract.py.)"""
    import torch.multiprocessing as mp
    mp.spawn(_staged_worker, args=(4, _free_port(), str(tmp_path)), nprocs=4, join=True)
    assert all((tmp_path / f"ok{r}").exists() for r in range(4))


def test_proc_grid_and_geometry():
    assert mrank.proc_grid(1) == (1, 1, 1) and mrank.proc_grid(2) == (2, 1, 1) and mrank.proc_grid(4) == (2, 2, 1)
    assert mrank.proc_grid(8) == (2, 2, 2) and mrank.proc_grid(6) == (3, 2, 1)
    g = mrank.plan_geometry((2, 2, 2), (0, 0, 0), (10, 10, 10), (1, 1, 0), 2.0, 5)
    assert tuple(g.coord) == (1, 0, 1) and tuple(g.blo) == (5.0, 0.0, 5.0)
    peers = {g.peer[c] for c in range(27) if c != 13 and g.peer[c] >= 0}
    assert peers == {0, 1, 2, 3, 4, 6, 7}                       # 7 distinct peers: one per xGMI link of the node
    assert g.peer[13 + 9] == -1 and g.peer[13 - 9] == 4          # open in z: nothing above the top brick
    assert g.shift[14][0] == -10.0 and g.shift[12][0] == 0.0      # +x from the upper brick crosses the periodic face
    # a brick shorter than the ghost cutoff is refused
    with pytest.raises(Exception):
        mrank.plan_geometry((4, 1, 1), (0, 0, 0), (10, 10, 10), (1, 1, 1), 3.0, 0)
    # ownership: wrapped, clamped at open boundaries
    x = np.array([[10.5, 0.1, -3.0], [4.999, 9.9, 25.0]])
    xw, own = mrank.plan_owner(g, x)
    assert np.allclose(xw, [[0.5, 0.1, -3.0], [4.999, 9.9, 25.0]]) and list(own) == [0, 3]
