"""The N > 1 device path of include/shhalo.h on the one GPU of the box: the ranks are host THREADS of this process that
share an in-process hub (same plan kernels, same pack / unpack kernels, same message pattern as the RCCL transport;
only the bytes travel by device copies instead of ncclSend/ncclRecv).  One process on the GPU, whatever the rank count.

 * static bed: decomposed HIP forces == single-domain HIP forces (<= 1e-12), device plan == host planner;
 * moving atoms: migration + rebuilds through the C++ loop (shhalo_run_device) against the single-rank loop;
 * BASELINE configs[3]: 1 M particles on 2x2x2 ranks, periodic in x and y, gravity: no atom lost, momentum balance,
   sampled rows equal to the single-domain HIP forces;
 * RCCL itself with the rank as its own communicator (world size 1) through the same C ABI.
"""
import os
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _ctx(lmax, shp, nq, kn=400.0, expo=1.25):
    from shpair import ShPair
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, len(shp))
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
    sp.coeff(1, 1, kn, expo)
    return sp


def _bed(n_target, periodic, nshapes=2, jitter=0.15, seed=9):
    from shpair import bed
    pts, lo, hi = bed.periodic_hcp(n_target, 1.9, periodic)
    rng = np.random.default_rng(seed)
    n = pts.shape[0]
    x = pts + rng.uniform(-jitter, jitter, pts.shape)
    quat = bed.random_quaternions(n, rng)
    sht = rng.integers(0, nshapes, n).astype(np.int32) if nshapes > 1 else np.zeros(n, np.int32)
    return x, quat, sht, np.arange(n, dtype=np.int32), lo, hi, rng


def _run_ranks(world, body):
    """body(rank) in one thread per rank; re-raises the first failure."""
    out, errs = [None] * world, []

    def work(r):
        try:
            out[r] = body(r)
        except BaseException as e:  # noqa: BLE001
            import traceback
            errs.append((r, repr(e), traceback.format_exc()))
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs[0]
    return out


def _distribute(grid, lo, hi, periodic, cut, x):
    from shpair import mrank
    g0 = mrank.plan_geometry(grid, lo, hi, periodic, cut, 0)
    return mrank.plan_owner(g0, x)   # (wrapped x, owner)


@pytest.mark.parametrize("grid,periodic", [((1, 1, 1), (1, 1, 1)), ((2, 1, 1), (1, 0, 0)), ((2, 2, 1), (1, 1, 0)), ((2, 2, 2), (1, 1, 1))])
def test_static_forces_and_plan_match_single_domain(grid, periodic):
    import torch
    from shpair import shapes, mrank
    from shpair.run import DeviceRun
    import halo_host
    world = int(np.prod(grid))
    lmax, nq, skin = 4, 8, 0.2
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    x, quat, sht, tag, lo, hi, _ = _bed(3000, periodic)
    n = x.shape[0]
    sp0 = _ctx(lmax, shp, nq)
    cut = 2.0 * max(sp0.rmax(s) for s in range(2)) + skin
    xw, owner = _distribute(grid, lo, hi, periodic, cut, x)
    hub = mrank.Hub(world) if world > 1 else None
    ty = np.ones(n, dtype=np.int32)

    def body(rank):
        sp = _ctx(lmax, shp, nq)
        halo = mrank.Halo(sp, rank, world, grid, lo, hi, periodic, skin, hub=hub)
        mine = owner == rank
        run = mrank.RankRun(sp, halo, xw[mine], quat[mine], sht[mine], tag[mine], dt=0.0)
        st = halo.stats()
        t, _, _, _, f, tq = run.owned()
        # the device plan against the host planner, row for row
        hr = halo_host.HostRank(rank, grid, lo, hi, periodic, cut, x, quat, ty, sht, tag)
        ghosts = (run.tag[run.n:run.n + run.nghost].cpu().numpy(), run.x[run.n:run.n + run.nghost].cpu().numpy())
        res = dict(tag=t, f=f, tq=tq, nghost=run.nghost, npairs=run.npairs, stats=st, ghosts=ghosts, hr=hr, n=run.n)
        halo.close()
        sp.close()
        return res
    parts = _run_ranks(world, body)
    # host planner: counts exchanged in-process
    for r, p in enumerate(parts):
        hr = p["hr"]
        inbox = {q: parts[q]["hr"].count_message(r) for q in range(world) if q != r}
        hr.set_counts(inbox)
        assert hr.n == p["n"] and hr.nghost == p["nghost"] and p["stats"]["npeers"] == hr.lay.npeers
        assert p["stats"]["nsend_rows"] == hr.lay.nsend
    # ghost rows (tag, x) in the planner's order: rebuild them from the host send lists
    for r, p in enumerate(parts):
        hr = p["hr"]
        exp_tag = np.zeros(hr.nghost, dtype=np.int64)
        exp_x = np.zeros((hr.nghost, 3))
        for c in range(27):
            q = hr.geo.peer[c]
            if c == 13 or q < 0:
                continue
            src = parts[q]["hr"]
            rows = src.send_list[26 - c]
            sh = np.array([src.geo.shift[26 - c][d] for d in range(3)])
            a = hr.lay.recv_off[c]
            exp_tag[a:a + rows.size] = src.tag[rows]
            exp_x[a:a + rows.size] = src.x[rows] + sh
        assert np.array_equal(p["ghosts"][0], exp_tag)
        assert np.abs(p["ghosts"][1] - exp_x).max() < 1e-13
    # forces against the single-domain HIP path on the same periodic box
    ref = DeviceRun(sp0, x, quat, sht, lo, hi, periodic, skin, dt=0.0)
    torch.cuda.synchronize()
    fr, tr = ref.f[:n].cpu().numpy(), ref.tq[:n].cpu().numpy()
    f, tq, seen = np.zeros_like(fr), np.zeros_like(tr), np.zeros(n, int)
    for p in parts:
        f[p["tag"]] = p["f"]
        tq[p["tag"]] = p["tq"]
        seen[p["tag"]] += 1
    assert np.all(seen == 1)
    assert sum(p["npairs"] for p in parts) == ref.npairs
    fs = np.abs(fr).max()
    assert fs > 0 and np.abs(f - fr).max() < 1e-12 * fs and np.abs(tq - tr).max() < 1e-12 * fs
    sp0.close()
    if hub:
        hub.close()


@pytest.mark.parametrize("overlap", [0, 1, 2])
@pytest.mark.parametrize("grid,periodic", [((2, 1, 1), (1, 1, 1)), ((2, 2, 1), (1, 1, 0)), ((1, 1, 1), (1, 1, 1)), ((2, 2, 2), (1, 1, 0))])
def test_dynamic_run_matches_single_rank(grid, periodic, overlap):
    """Atoms move fast enough to change owner; the C++ loop of every rank (exchange, borders, rebuilds, forward,
    reverse) must reproduce the single-rank loop's trajectory.  overlap = 1: option "halo_overlap" — the list partitioned
    into owned-only and ghost slots, the forward exchange on a stream of its own beside the owned-only pair kernels;
    overlap = 2: the reverse exchange beside the second half of the owned-only slots as well."""
    import torch
    from shpair import shapes, mrank
    from shpair.run import DeviceRun
    world = int(np.prod(grid))
    lmax, nq, skin, dt, nsteps = 4, 8, 0.2, 2e-3, int(os.environ.get("SHPAIR_SOAK_STEPS", "120"))   # (a longer soak: the env variable)
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    x, quat, sht, tag, lo, hi, rng = _bed(1500 if world < 8 else 4000, periodic)
    n = x.shape[0]
    v0 = 2.0 * rng.normal(size=(n, 3))
    mask = np.where(tag % 11 == 0, 2, 1).astype(np.int32)   # every 11th particle is frozen (another group)
    v0[mask == 2] = 0.0
    grav = (0.0, 0.0, -0.5 if not periodic[2] else 0.0)
    sp0 = _ctx(lmax, shp, nq)
    cut = 2.0 * max(sp0.rmax(s) for s in range(2)) + skin
    xw, owner = _distribute(grid, lo, hi, periodic, cut, x)
    hub = mrank.Hub(world) if world > 1 else None

    def body(rank):
        sp = _ctx(lmax, shp, nq)
        sp.set_option("halo_overlap", overlap)
        halo = mrank.Halo(sp, rank, world, grid, lo, hi, periodic, skin, hub=hub)
        mine = owner == rank
        run = mrank.RankRun(sp, halo, xw[mine], quat[mine], sht[mine], tag[mine], v=v0[mine], mask=mask[mine], dt=dt, gravity=grav,
                            gamma_t=0.05, gamma_r=0.02)
        n0 = run.n
        run.run(nsteps)
        t, X, V, Q, _, _ = run.owned()
        st = halo.stats()
        res = dict(tag=t, x=X, v=V, q=Q, builds=run.builds, n0=n0, n1=run.n, nghost=run.nghost, stats=st)
        halo.close()
        sp.close()
        return res
    parts = _run_ranks(world, body)
    tg = np.concatenate([p["tag"] for p in parts])
    o = np.argsort(tg)
    assert np.array_equal(tg[o], np.arange(n)), "atoms lost or duplicated"
    X = np.concatenate([p["x"] for p in parts])[o]
    V = np.concatenate([p["v"] for p in parts])[o]
    Q = np.concatenate([p["q"] for p in parts])[o]
    ref = DeviceRun(sp0, x, quat, sht, lo, hi, periodic, skin, mask=mask, dt=dt, gravity=grav, gamma_t=0.05, gamma_r=0.02)
    ref.v[:] = torch.from_numpy(v0).to(ref.v.device)
    ref.force()
    ref.run(nsteps)
    torch.cuda.synchronize()
    xr, vr, qr = ref.x[:n].cpu().numpy(), ref.v.cpu().numpy(), ref.q[:n].cpu().numpy()
    dx = X - xr
    for d in range(3):
        if periodic[d]:
            dx[:, d] -= (hi[d] - lo[d]) * np.round(dx[:, d] / (hi[d] - lo[d]))
    assert np.abs(dx).max() < 1e-7 and np.abs(V - vr).max() < 1e-6 * np.abs(vr).max()
    assert np.abs(np.abs((Q * qr).sum(1)) - 1).max() < 1e-9
    builds = [p["builds"] for p in parts]
    assert min(builds) >= 3 and len(set(builds)) == 1                 # every rank rebuilt, and together
    if world > 1:
        assert sum(p["stats"]["migrated_out"] for p in parts) > 0      # atoms changed owner
        assert sum(p["stats"]["migrated_out"] for p in parts) == sum(p["stats"]["migrated_in"] for p in parts)
        assert min(p["nghost"] for p in parts) > 0
    sp0.close()
    if hub:
        hub.close()


@pytest.mark.parametrize("overlap", [0, 1, 2])
def test_a_rank_that_loses_all_its_atoms_keeps_stepping(overlap):
    """Every atom of rank 0 migrates to rank 1 in the middle of a run (a cluster flying across the brick face).  From then
    on rank 0 holds an EMPTY list: nothing of its previous list's interior / boundary partition may survive the rebuild
    (shstep_neighbor_build_device's nlocal == 0 branch; the slot ranges of shhalo_run_device are cut at n_interior) —
    with "halo_overlap" 1 and 2 a stale n_interior made rank 0 fail in mid-loop and left rank 1 waiting for it."""
    import torch
    from shpair import shapes, mrank
    from shpair.run import DeviceRun
    grid, periodic, world = (2, 1, 1), (0, 0, 0), 2
    lmax, nq, skin, dt, nsteps = 4, 8, 0.2, 2e-3, 560
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    lo, hi = np.zeros(3), np.array([40.0, 9.5, 9.5])
    rng = np.random.default_rng(31)
    # spacing 2.3: bounding spheres overlap a little (listed pairs, some of them touching lightly), the blocks stay blocks
    gy, gz = np.meshgrid(1.0 + 2.3 * np.arange(4), 1.0 + 2.3 * np.arange(4), indexing="ij")

    def block(x0):
        return np.concatenate([np.stack([np.full(16, x0 + 2.3 * k), gy.ravel(), gz.ravel()], axis=1) for k in range(2)])
    x = np.concatenate([block(15.0), block(30.0)]) + rng.uniform(-0.03, 0.03, (64, 3))
    n = x.shape[0]
    from shpair import bed
    quat = bed.random_quaternions(n, rng)
    sht = rng.integers(0, 2, n).astype(np.int32)
    tag = np.arange(n, dtype=np.int32)
    v0 = np.zeros((n, 3))
    v0[:32, 0] = 6.0      # the left cluster crosses x = 20 between t = 0.45 and t = 0.85, whole by step ~430
    sp0 = _ctx(lmax, shp, nq)
    cut = 2.0 * max(sp0.rmax(s) for s in range(2)) + skin
    xw, owner = _distribute(grid, lo, hi, periodic, cut, x)
    assert (owner == 0).sum() == 32 and (owner == 1).sum() == 32
    hub = mrank.Hub(world)

    def body(rank):
        sp = _ctx(lmax, shp, nq)
        sp.set_option("halo_overlap", overlap)
        halo = mrank.Halo(sp, rank, world, grid, lo, hi, periodic, skin, hub=hub)
        mine = owner == rank
        run = mrank.RankRun(sp, halo, xw[mine], quat[mine], sht[mine], tag[mine], v=v0[mine], dt=dt, capacity=256)
        counts = []
        for _ in range(nsteps // 40):
            run.run(40)
            counts.append(run.n)
        t, X, V, _, _, _ = run.owned()
        res = dict(tag=t, x=X, v=V, counts=counts, builds=run.builds)
        halo.close()
        sp.close()
        return res
    parts = _run_ranks(world, body)
    assert parts[0]["counts"][0] == 32 and parts[0]["counts"][-1] == 0 and parts[1]["counts"][-1] == 64
    assert parts[0]["counts"].count(0) >= 2          # rank 0 went on stepping with nothing: more than one call of the loop
    tg = np.concatenate([p["tag"] for p in parts])
    o = np.argsort(tg)
    assert np.array_equal(tg[o], np.arange(n))
    X = np.concatenate([p["x"] for p in parts])[o]
    V = np.concatenate([p["v"] for p in parts])[o]
    ref = DeviceRun(sp0, x, quat, sht, lo, hi, periodic, skin, dt=dt)
    ref.v[:] = torch.from_numpy(v0).to(ref.v.device)
    ref.force()
    ref.run(nsteps)
    torch.cuda.synchronize()
    assert np.abs(X - ref.x[:n].cpu().numpy()).max() < 1e-8 and np.abs(V - ref.v.cpu().numpy()).max() < 1e-7
    sp0.close()
    hub.close()


def test_rccl_send_recv_to_self_and_allreduce_through_the_transport():
    """The only way ncclSend / ncclRecv execute on a one-GPU box: shhalo_transport_selftest — this rank, its own
    communicator (world 1), sends a megabyte to ITSELF through RcclTransport::exchange (ncclGroupStart, ncclRecv from
    self, ncclSend to self, ncclGroupEnd: the very calls, argument types and byte counts of the halo's exchange) and
    all-reduces doubles and ints in place; every byte and value is checked.  (Between two devices: never run here.)"""
    from shpair import shapes, mrank
    lmax, nq = 4, 8
    shp = [shapes.random_shape(lmax, 400, amp=0.2)]
    sp = _ctx(lmax, shp, nq)
    lo, hi = np.zeros(3), np.array([30.0, 30.0, 30.0])
    halo = mrank.Halo(sp, 0, 1, (1, 1, 1), lo, hi, (1, 1, 1), 0.2, unique_id_bytes=mrank.unique_id())
    st = halo.stats()
    assert st["transport"] == 1 and st["nranks_transport"] == 1 and st["rccl_version"] > 20000
    for nbytes in (8, 4096, 1 << 20, (1 << 20) + 13):
        halo.transport_selftest(nbytes, sp.own_stream())
    halo.close()
    # the hub transport between rank threads: send to self goes through the same post / match code as any peer
    hub = mrank.Hub(2)

    def body(rank):
        spr = _ctx(lmax, shp, nq)
        h = mrank.Halo(spr, rank, 2, (2, 1, 1), lo, hi, (1, 1, 1), 0.2, hub=hub)
        h.transport_selftest(1 << 16, spr.own_stream())
        h.close()
        spr.close()
        return True
    assert all(_run_ranks(2, body))
    hub.close()
    sp.close()


def test_config4_one_million_particles_eight_ranks():
    """BASELINE configs[3] as a rehearsal: 1 M particles, L_max = 6, 2x2x2 bricks, periodic in x and y, gravity;
    eight rank threads on the one GPU.  A few steps of the C++ loop, then: no atom lost, the forces of a sample of
    rows equal the single-domain HIP forces, and contact forces sum to zero."""
    import torch
    from shpair import shapes, bed, mrank, ShPair
    from shpair.run import DeviceRun
    grid, periodic, world = (2, 2, 2), (1, 1, 0), 8
    lmax, nq, skin = 6, 16, 0.1
    shp = [shapes.random_shape(lmax, bed.SEED0 + 2)]
    x, quat, sht, tag, lo, hi, _ = _bed(1000000, periodic, nshapes=1, jitter=0.04, seed=bed.SEED0 + 7)
    n = x.shape[0]
    assert n > 950000
    sp0 = _ctx(lmax, shp, nq, kn=1000.0)
    cut = 2.0 * sp0.rmax(0) + skin
    xw, owner = _distribute(grid, lo, hi, periodic, cut, x)
    hub = mrank.Hub(world)

    def body(rank):
        sp = _ctx(lmax, shp, nq, kn=1000.0)
        halo = mrank.Halo(sp, rank, world, grid, lo, hi, periodic, skin, hub=hub)
        mine = owner == rank
        run = mrank.RankRun(sp, halo, xw[mine], quat[mine], sht[mine], tag[mine], dt=1e-3, gravity=(0.0, 0.0, -1.0),
                            capacity=int(1.6 * mine.sum()) + 1024)
        st0 = halo.stats()
        t0, _, _, _, f0, tq0 = run.owned()           # static forces (contacts + gravity) of the initial bed
        run.run(4)
        t1 = run.owned()[0]
        res = dict(tag0=t0, f0=f0, tq0=tq0, tag1=t1, n=run.n, nghost=run.nghost, npairs=run.npairs, stats=st0)
        halo.close()
        sp.close()
        return res
    parts = _run_ranks(world, body)
    tg = np.sort(np.concatenate([p["tag1"] for p in parts]))
    assert np.array_equal(tg, np.arange(n)), "atoms lost or duplicated"
    assert all(p["stats"]["npeers"] >= 3 for p in parts) and all(p["nghost"] > 10000 for p in parts)
    # single-domain HIP forces of the same bed
    ref = DeviceRun(sp0, x, quat, sht, lo, hi, periodic, skin, dt=1e-3, gravity=(0.0, 0.0, -1.0))
    torch.cuda.synchronize()
    fr, tr = ref.f[:n].cpu().numpy(), ref.tq[:n].cpu().numpy()
    assert sum(p["npairs"] for p in parts) == ref.npairs
    f = np.zeros_like(fr)
    tq = np.zeros_like(tr)
    for p in parts:
        f[p["tag0"]] = p["f0"]
        tq[p["tag0"]] = p["tq0"]
    fs = np.abs(fr).max()
    rows = np.random.default_rng(1).choice(n, 20000, replace=False)
    assert np.abs(f[rows] - fr[rows]).max() < 1e-12 * fs and np.abs(tq[rows] - tr[rows]).max() < 1e-12 * fs
    # contact forces cancel pairwise: what is left of the sum is gravity (m g per particle)
    m = sp0.body(0)[0]
    tot = f.sum(axis=0)
    assert abs(tot[0]) < 1e-8 * fs * np.sqrt(n) and abs(tot[1]) < 1e-8 * fs * np.sqrt(n)
    assert abs(tot[2] + m * n) < 1e-9 * m * n + 1e-8 * fs * np.sqrt(n)
    sp0.close()
    hub.close()


def test_rccl_self_communicator_through_the_c_abi():
    """RCCL itself behind shhalo_create_rccl with one rank, which is all a one-GPU box can hold (RCCL refuses two ranks
    on one device): ncclGetUniqueId, ncclCommInitRank, ncclCommCount and ncclAllReduce (thermo sums, the rebuild
    decision) are EXECUTED.  ncclSend/ncclRecv are NOT: in a periodic box every direction leads back to the rank, the
    blocks are copied by the pack kernels and RcclTransport::exchange returns at its empty-message early-out.  The
    point-to-point path over RCCL (peer order, ncclChar byte counts) has never run on this pool (one GPU per box); its
    first run is `bench.py --gpus N` with the forces verified against a single-domain compute (verify_rel_err)."""
    import torch
    from shpair import shapes, mrank
    from shpair.run import DeviceRun
    lmax, nq, skin = 4, 8, 0.2
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    periodic = (1, 1, 1)
    x, quat, sht, tag, lo, hi, _ = _bed(1500, periodic)
    n = x.shape[0]
    sp = _ctx(lmax, shp, nq)
    uid = mrank.unique_id()
    assert len(uid) == 128
    halo = mrank.Halo(sp, 0, 1, (1, 1, 1), lo, hi, periodic, skin, unique_id_bytes=uid)
    st = halo.stats()
    assert st["transport"] == 1 and st["nranks_transport"] == 1 and st["rccl_version"] > 20000
    run = mrank.RankRun(sp, halo, x, quat, sht, tag, dt=1e-3)
    run.run(20)
    e = torch.tensor([1.5, 2.5, -4.0], dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    halo.allreduce_sum(e.data_ptr(), 3, run.stream)
    sp.synchronize()
    assert e.cpu().tolist() == [1.5, 2.5, -4.0]
    sp1 = _ctx(lmax, shp, nq)
    ref = DeviceRun(sp1, x, quat, sht, lo, hi, periodic, skin, dt=1e-3)
    ref.run(20)
    torch.cuda.synchronize()
    t, X, V, Q, _, _ = run.owned()
    dx = X - ref.x[:n].cpu().numpy()
    for d in range(3):
        dx[:, d] -= (hi[d] - lo[d]) * np.round(dx[:, d] / (hi[d] - lo[d]))
    assert np.abs(dx).max() < 1e-9
    halo.close()
    sp.close()
    sp1.close()


def test_rank_without_atoms_of_its_own():
    """A brick that owns nothing (all atoms sit in the other half of an open box) still takes part: it receives ghosts,
    builds an empty list, sends nothing back — and the other rank's forces are the single-domain forces."""
    import torch
    from shpair import shapes, mrank
    from shpair.run import DeviceRun
    lmax, nq, skin = 4, 8, 0.2
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    per = (0, 0, 0)
    x, quat, sht, tag, lo, hi, _ = _bed(1200, per)
    hi = hi.copy()
    hi[0] = lo[0] + 2.0 * (x[:, 0].max() - lo[0]) + 0.5          # the atoms fill the left half only (up to the cut)
    world, grid = 2, (2, 1, 1)
    sp0 = _ctx(lmax, shp, nq)
    cut = 2.0 * max(sp0.rmax(s) for s in range(2)) + skin
    xw, owner = _distribute(grid, lo, hi, per, cut, x)
    assert (owner == 1).sum() == 0
    hub = mrank.Hub(world)

    def body(rank):
        sp = _ctx(lmax, shp, nq)
        halo = mrank.Halo(sp, rank, world, grid, lo, hi, per, skin, hub=hub)
        mine = owner == rank
        run = mrank.RankRun(sp, halo, xw[mine], quat[mine], sht[mine], tag[mine], dt=1e-3, capacity=4096)
        run.run(3)
        t, X, _, _, f, tq = run.owned()
        res = dict(tag=t, x=X, n=run.n, nghost=run.nghost, npairs=run.npairs)
        halo.close()
        sp.close()
        return res
    parts = _run_ranks(world, body)
    assert parts[1]["n"] == 0 and parts[1]["npairs"] == 0 and parts[1]["nghost"] > 0 and parts[0]["nghost"] == 0
    ref = DeviceRun(sp0, x, quat, sht, lo, hi, per, skin, dt=1e-3)
    ref.run(3)
    torch.cuda.synchronize()
    n = x.shape[0]
    assert np.array_equal(parts[0]["tag"], np.arange(n))
    assert np.abs(parts[0]["x"] - ref.x[:n].cpu().numpy()).max() < 1e-12
    sp0.close()
    hub.close()


def test_failure_modes_are_loud():
    """Bricks shorter than the ghost cutoff, a capacity that cannot take the ghosts, and an atom that left its brick
    and all neighbouring bricks between two exchanges must come back as error codes, never as silent garbage."""
    import torch
    from shpair import shapes, mrank, ShPairError
    lmax, nq, skin = 4, 8, 0.2
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    periodic = (1, 1, 1)
    x, quat, sht, tag, lo, hi, _ = _bed(1500, periodic)
    sp = _ctx(lmax, shp, nq)
    with pytest.raises(ShPairError) as e:                      # 8 bricks along x: shorter than 2 Rmax + skin
        mrank.Halo(sp, 0, 8, (8, 1, 1), lo, hi, periodic, skin, hub=mrank.Hub(8))
    assert e.value.code == -1 and "ghost cutoff" in str(e.value)
    halo = mrank.Halo(sp, 0, 1, (1, 1, 1), lo, hi, periodic, skin)
    with pytest.raises(ShPairError) as e:                      # room for the owned rows but not for their periodic images
        mrank.RankRun(sp, halo, x, quat, sht, tag, dt=1e-3, capacity=x.shape[0] + 10)
    assert e.value.code == -5 and "capacity" in str(e.value)
    halo.close()
    sp.close()
    # a lost atom: 4 x 1 x 1 ranks, periodic in x — rank 2 is not a neighbour of rank 0
    world, grid, per = 4, (4, 1, 1), (1, 1, 0)
    x, quat, sht, tag, lo, hi, _ = _bed(4000, per)
    sp0 = _ctx(lmax, shp, nq)
    cut = 2.0 * max(sp0.rmax(s) for s in range(2)) + skin
    sp0.close()
    xw, owner = _distribute(grid, lo, hi, per, cut, x)
    hub = mrank.Hub(world)

    def body(rank):
        sp = _ctx(lmax, shp, nq)
        halo = mrank.Halo(sp, rank, world, grid, lo, hi, per, skin, hub=hub)
        mine = owner == rank
        run = mrank.RankRun(sp, halo, xw[mine], quat[mine], sht[mine], tag[mine], dt=1e-3)
        n0 = run.n
        if rank == 0:                                   # teleport one atom into the brick of rank 2
            run.x[0, 0] = float(lo[0] + 0.625 * (hi[0] - lo[0]))
            torch.cuda.synchronize()
        msg = None
        try:
            halo.exchange(run.a, run.stream)            # Comm::exchange only: the count messages still pair up
            run.sync()
        except ShPairError as err:
            msg = (err.code, str(err))
        res = (msg, n0, run.n)
        halo.close()
        sp.close()
        return res
    out = _run_ranks(world, body)
    hub.close()
    # the decision is collective (one max-all-reduce of an error word before any row travels): the rank that lost the
    # atom says so, every other rank returns too — naming the failure — instead of waiting for rows that never come
    assert out[0][0] is not None and out[0][0][0] == -4 and "lost atom" in out[0][0][1]
    assert all(o[0] is not None and o[0][0] == -4 and "another rank failed" in o[0][1] and o[1] == o[2] for o in out[1:])


def test_rank_local_failure_stops_every_rank():
    """one rank's capacity runs out at a reneighbouring inside the C++ loop.  Over RCCL the peers would
    wait for ever inside ncclRecv for ghost rows that are never sent; the ranks therefore agree on an error word before
    the rows travel.  Here (hub transport, whose waits time out after 120 s) both ranks must come back within seconds:
    the short rank with SHPAIR_ENOMEM, the other with SHPAIR_ESTATE naming it.  The same for a shape index outside
    the table that arrives in a migrated row (found by the list build)."""
    import time
    import torch
    from shpair import shapes, mrank, ShPairError
    lmax, nq, skin = 4, 8, 0.2
    shp = [shapes.random_shape(lmax, 400 + s, amp=0.2) for s in range(2)]
    world, grid, per = 2, (2, 1, 1), (1, 1, 1)
    x, quat, sht, tag, lo, hi, rng = _bed(2400, per)
    sp0 = _ctx(lmax, shp, nq)
    cut = 2.0 * max(sp0.rmax(s) for s in range(2)) + skin
    sp0.close()
    xw, owner = _distribute(grid, lo, hi, per, cut, x)
    v = 0.5 * rng.normal(size=x.shape)

    for mode in ("capacity", "shape"):
        hub = mrank.Hub(world)

        def body(rank):
            sp = _ctx(lmax, shp, nq)
            halo = mrank.Halo(sp, rank, world, grid, lo, hi, per, skin, hub=hub)
            mine = owner == rank
            n = int(mine.sum())
            run = mrank.RankRun(sp, halo, xw[mine], quat[mine], sht[mine], tag[mine], v=v[mine], dt=2e-3)
            ng0 = run.nghost
            if mode == "capacity" and rank == 1:
                run.a.nmax = n + ng0 - 5          # declared capacity (the tensors are larger): short at the next reneighbouring
            if mode == "shape" and rank == 0:
                run.sh[:n] = torch.where(run.x[:n, 0] > float(run.x[:n, 0].max()) - 0.3, 7, run.sh[:n].long()).int()
                torch.cuda.synchronize()
            msg, t0 = None, time.perf_counter()
            try:
                for _ in range(60):
                    run.run(5)
            except ShPairError as err:
                msg = (err.code, str(err))
            el = time.perf_counter() - t0
            halo.close()
            sp.close()
            return msg, el
        out = _run_ranks(world, body)
        hub.close()
        assert all(o[0] is not None for o in out), (mode, out)
        assert max(o[1] for o in out) < 60.0, (mode, out)                # nobody sat in a 120 s hub timeout
        bad, other = (1, 0) if mode == "capacity" else (0, 1)
        if mode == "capacity":
            assert out[bad][0][0] == -5 and "capacity" in out[bad][0][1]
        else:
            assert out[bad][0][0] == -1 and "shape index" in out[bad][0][1]
        assert out[other][0][0] == -4 and "another rank failed" in out[other][0][1]
