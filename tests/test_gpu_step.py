"""GPU parity tests of docs/SPEC.md Part II (include/shstep.h through ctypes) against the CPU oracle:
rigid-body table, nve integrator, body forces, energies, periodic ghosts, device-built half list, and the
whole device-resident step on a periodic bed."""
import numpy as np
import pytest

from common import coeff_tables, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-13


def random_quats(rng, n):
    q = rng.normal(size=(n, 4))
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def make_ctx(shp, lmax, nq=8, kn=1000.0, expo=1.25, rho=None):
    from shpair import ShPair
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, len(shp))
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
        if rho is not None:
            sp.set_density(s, rho[s])
    sp.coeff(1, 1, kn, expo)
    return sp


def dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def state(rng, n, nshapes):
    return dict(x=rng.normal(size=(n, 3)), v=rng.normal(size=(n, 3)), quat=random_quats(rng, n),
                angmom=rng.normal(size=(n, 3)), f=rng.normal(size=(n, 3)), torque=rng.normal(size=(n, 3)),
                shtype=rng.integers(0, nshapes, n).astype(np.int32), mask=np.ones(n, dtype=np.int32))


@pytest.fixture(scope="module")
def shp2():
    from shpair import shapes
    return [shapes.random_shape(6, 31, amp=0.3), shapes.random_shape(6, 32, amp=0.2)]


def test_body_table_matches_oracle(oracle, shp2):
    from shpair import capi
    sp = make_ctx(shp2, 6, rho=[2.0, 0.7])
    for s, rho in enumerate((2.0, 0.7)):
        mp = oracle.mass_props(6, shp2[s])
        assert np.abs(capi.shape_mass_props(6, shp2[s]) - mp).max() < 1e-13
        m, com, inertia = sp.body(s)
        assert abs(m - rho * mp[0]) < 1e-13 * m
        assert np.abs(com - mp[1:4]).max() < 1e-14
        assert np.abs(inertia - rho * mp[4:]).max() < 1e-13
    sp.close()


@pytest.mark.parametrize("phase", [0, 1])
def test_nve_device_matches_oracle(oracle, shp2, phase):
    import torch
    rng = np.random.default_rng(40 + phase)
    n = 3001                                    # not a multiple of the workgroup size
    rho = np.array([2.0, 0.7])
    sp = make_ctx(shp2, 6, rho=rho)
    mp = np.array([oracle.mass_props(6, a) for a in shp2])
    st = state(rng, n, 2)
    st["mask"][::7] = 4                         # frozen group
    d = {k: dev(v) for k, v in st.items()}
    sp.nve_device(phase, n, 3e-3, *(d[k].data_ptr() for k in ("x", "v", "quat", "angmom", "f", "torque", "shtype", "mask")),
                  groupbit=1)
    torch.cuda.synchronize()
    ref = {k: v.copy() for k, v in st.items()}
    oracle.nve(phase, 3e-3, mp, rho, ref["x"], ref["v"], ref["quat"], ref["angmom"], ref["f"], ref["torque"], ref["shtype"],
               ref["mask"], groupbit=1)
    for k in ("x", "v", "quat", "angmom"):
        got = d[k].cpu().numpy()
        assert rel_err(got, ref[k]) < TOL, k
        fr = st["mask"] == 4
        assert np.array_equal(got[fr], st[k][fr])
    if phase == 1:
        assert np.array_equal(d["x"].cpu().numpy(), st["x"]) and np.array_equal(d["quat"].cpu().numpy(), st["quat"])
    sp.close()


def test_nve_host_pointer_form(oracle, shp2):
    rng = np.random.default_rng(42)
    n = 777
    rho = np.array([1.0, 1.0])
    sp = make_ctx(shp2, 6)
    mp = np.array([oracle.mass_props(6, a) for a in shp2])
    st = state(rng, n, 2)
    ref = {k: v.copy() for k, v in st.items()}
    for ph in (0, 1):
        sp.nve(ph, 1e-3, st["x"], st["v"], st["quat"], st["angmom"], st["f"], st["torque"], st["shtype"], st["mask"])
        oracle.nve(ph, 1e-3, mp, rho, ref["x"], ref["v"], ref["quat"], ref["angmom"], ref["f"], ref["torque"], ref["shtype"], ref["mask"])
    for k in ("x", "v", "quat", "angmom"):
        assert rel_err(st[k], ref[k]) < TOL, k
    # a bad shape index is refused before anything is launched
    from shpair.capi import ShPairError
    st["shtype"][5] = 9
    with pytest.raises(ShPairError):
        sp.nve(0, 1e-3, st["x"], st["v"], st["quat"], st["angmom"], st["f"], st["torque"], st["shtype"], st["mask"])
    sp.close()


def test_post_force_and_energies(oracle, shp2):
    import torch
    rng = np.random.default_rng(43)
    n = 2000
    rho = np.array([1.5, 0.9])
    sp = make_ctx(shp2, 6, rho=rho)
    mp = np.array([oracle.mass_props(6, a) for a in shp2])
    st = state(rng, n, 2)
    st["mask"][::5] = 2
    g = np.array([0.2, -0.1, -9.81])
    d = {k: dev(v) for k, v in st.items()}
    sp.post_force_device(n, g, 0.4, 0.25, d["v"].data_ptr(), d["quat"].data_ptr(), d["angmom"].data_ptr(),
                         d["shtype"].data_ptr(), d["mask"].data_ptr(), d["f"].data_ptr(), d["torque"].data_ptr())
    out = torch.zeros(3, dtype=torch.float64, device="cuda:0")
    sp.energies_device(n, g, d["x"].data_ptr(), d["v"].data_ptr(), d["quat"].data_ptr(), d["angmom"].data_ptr(),
                       d["shtype"].data_ptr(), d["mask"].data_ptr(), out.data_ptr())
    torch.cuda.synchronize()
    f, tq = st["f"].copy(), st["torque"].copy()
    oracle.post_force(mp, rho, g, 0.4, 0.25, st["v"], st["quat"], st["angmom"], st["shtype"], st["mask"], f, tq)
    assert rel_err(d["f"].cpu().numpy(), f) < TOL
    assert rel_err(d["torque"].cpu().numpy(), tq) < TOL
    e = oracle.energies(mp, rho, g, st["x"], st["v"], st["quat"], st["angmom"], st["shtype"], st["mask"])
    assert np.abs(out.cpu().numpy() - e).max() < 1e-12 * np.abs(e).max()
    sp.close()


def test_bad_shape_index_on_device_is_reported(shp2):
    import torch
    from shpair.capi import ShPairError
    rng = np.random.default_rng(44)
    n = 300
    sp = make_ctx(shp2, 6)
    st = state(rng, n, 2)
    st["shtype"][17] = 5
    d = {k: dev(v) for k, v in st.items()}
    sp.nve_device(0, n, 1e-3, *(d[k].data_ptr() for k in ("x", "v", "quat", "angmom", "f", "torque", "shtype", "mask")))
    torch.cuda.synchronize()
    assert np.array_equal(d["x"].cpu().numpy()[17], st["x"][17])        # skipped, not integrated with garbage
    sp.set_box([-5, -5, -5], [5, 5, 5], [0, 0, 0], 0.1)
    with pytest.raises(ShPairError):                                    # surfaces at the next blocking call
        sp.neighbor_build_device(n, 0, d["x"].data_ptr(), d["shtype"].data_ptr())
    sp.close()


def _periodic_case(oracle, n, periodic, seed, lmax=4, nshapes=2, skin=0.15, jitter=0.3):
    from shpair import shapes
    rng = np.random.default_rng(seed)
    shp = [shapes.random_shape(lmax, 50 + s, amp=0.2) for s in range(nshapes)]
    rmax = np.array([oracle.shape_rmax(lmax, a) for a in shp])
    m = int(round(n ** (1 / 3)))
    box = np.array([m, m, m]) * 1.9
    lo = np.array([-1.0, 0.5, 2.0])
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)
    x = lo + (g + 0.5) * 1.9 + rng.uniform(-jitter, jitter, (g.shape[0], 3))
    x[:3] -= box * np.array(periodic)             # a few outside the box
    n = x.shape[0]
    return dict(n=n, lmax=lmax, shapes=shp, rmax=rmax, lo=lo, hi=lo + box, box=box, periodic=periodic, skin=skin,
                x=x, quat=random_quats(rng, n), type=np.ones(n, dtype=np.int32),
                shtype=rng.integers(0, nshapes, n).astype(np.int32))


def _device_rows(case, nmax):
    import torch
    n = case["n"]
    x = torch.zeros(nmax, 3, dtype=torch.float64, device="cuda:0")
    q = torch.zeros(nmax, 4, dtype=torch.float64, device="cuda:0")
    ty = torch.zeros(nmax, dtype=torch.int32, device="cuda:0")
    sh = torch.zeros(nmax, dtype=torch.int32, device="cuda:0")
    x[:n] = dev(case["x"]); q[:n] = dev(case["quat"]); ty[:n] = dev(case["type"]); sh[:n] = dev(case["shtype"])
    return x, q, ty, sh


@pytest.mark.parametrize("periodic", [(1, 1, 1), (1, 1, 0), (0, 0, 0)])
def test_borders_and_half_list_match_oracle(oracle, periodic):
    import torch
    case = _periodic_case(oracle, 1000, periodic, 60)
    n = case["n"]
    sp = make_ctx(case["shapes"], case["lmax"])
    for s in range(len(case["shapes"])):
        assert abs(sp.rmax(s) - case["rmax"][s]) < 1e-14
    sp.set_box(case["lo"], case["hi"], periodic, case["skin"])
    nmax = 4 * n
    x, q, ty, sh = _device_rows(case, nmax)
    ng = sp.borders_device(n, nmax, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    cmax = 2 * case["rmax"].max() + case["skin"]
    xw = case["x"].copy()
    own, shift = oracle.borders(xw, case["lo"], case["hi"], periodic, cmax)
    assert ng == own.size
    assert (ng > 0) == any(periodic)
    xa = np.concatenate([xw, xw[own] + shift * case["box"]])
    assert np.abs(x[: n + ng].cpu().numpy() - xa).max() < 1e-13
    assert np.array_equal(q[n: n + ng].cpu().numpy(), case["quat"][own])
    assert np.array_equal(sh[n: n + ng].cpu().numpy(), case["shtype"][own])
    assert np.array_equal(ty[n: n + ng].cpu().numpy(), case["type"][own])
    # half list: identical rows (both sides order a row by j)
    npairs = sp.neighbor_build_device(n, ng, x.data_ptr(), sh.data_ptr())
    offs, jl = sp.copy_neighbors(n, npairs)
    tag = np.concatenate([np.arange(n), own]).astype(np.int32)
    sha = np.concatenate([case["shtype"], case["shtype"][own]])
    o_offs, o_jl = oracle.half_list(n, x[: n + ng].cpu().numpy(), sha, tag, case["rmax"], case["skin"])
    assert npairs == o_jl.size and npairs > 2 * n
    assert np.array_equal(offs, o_offs)
    assert np.array_equal(jl, o_jl)
    # forward: move the owners, ghosts follow; reverse: ghost forces fold into owners
    x[:n] += 0.01
    q[:n] = dev(random_quats(np.random.default_rng(1), n))
    sp.forward_device(x.data_ptr(), q.data_ptr())
    f = dev(np.random.default_rng(2).normal(size=(n + ng, 3)))
    t = dev(np.random.default_rng(3).normal(size=(n + ng, 3)))
    f0, t0 = f.cpu().numpy().copy(), t.cpu().numpy().copy()
    sp.reverse_device(f.data_ptr(), t.data_ptr())
    torch.cuda.synchronize()
    xs = x.cpu().numpy()
    if ng:
        assert np.abs(xs[n: n + ng] - (xs[own] + shift * case["box"])).max() < 1e-13
        assert np.array_equal(q[n: n + ng].cpu().numpy(), q[:n].cpu().numpy()[own])
    fe, te = f0[:n].copy(), t0[:n].copy()
    np.add.at(fe, own, f0[n:])
    np.add.at(te, own, t0[n:])
    assert np.abs(f[:n].cpu().numpy() - fe).max() < 1e-13
    assert np.abs(t[:n].cpu().numpy() - te).max() < 1e-13
    sp.close()


def test_neighbor_check_and_capacity(oracle):
    from shpair.capi import ShPairError
    case = _periodic_case(oracle, 512, (1, 1, 1), 61)
    n = case["n"]
    sp = make_ctx(case["shapes"], case["lmax"])
    sp.set_box(case["lo"], case["hi"], (1, 1, 1), case["skin"])
    x, q, ty, sh = _device_rows(case, n + 10)
    with pytest.raises(ShPairError) as e:                 # not enough rows for the ghosts
        sp.borders_device(n, n + 10, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    assert e.value.code == -5
    x, q, ty, sh = _device_rows(case, 4 * n)
    assert sp.neighbor_check_device(n, x.data_ptr())      # no list yet
    ng = sp.borders_device(n, 4 * n, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    sp.neighbor_build_device(n, ng, x.data_ptr(), sh.data_ptr())
    assert not sp.neighbor_check_device(n, x.data_ptr())
    x[100, 1] += 0.49 * case["skin"]
    assert not sp.neighbor_check_device(n, x.data_ptr())
    x[100, 1] += 0.02 * case["skin"]
    assert sp.neighbor_check_device(n, x.data_ptr())
    # a box edge shorter than twice the ghost cutoff is refused
    sp.set_box([0, 0, 0], [3.0, 50, 50], (1, 0, 0), 0.1)
    with pytest.raises(ShPairError):
        sp.borders_device(n, 4 * n, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    sp.close()


def test_device_resident_step_matches_oracle_pipeline(oracle):
    """Periodic bed: ghosts + list built on the device, pair forces, reverse, post_force, both integrator
    phases — against the same sequence made of oracle pieces."""
    import torch
    case = _periodic_case(oracle, 729, (1, 1, 1), 62, lmax=4, nshapes=2)
    n, nq = case["n"], 8
    rho = np.array([1.2, 0.8])
    sp = make_ctx(case["shapes"], case["lmax"], nq=nq, rho=rho)
    sp.set_box(case["lo"], case["hi"], (1, 1, 1), case["skin"])
    nmax = 4 * n
    x, q, ty, sh = _device_rows(case, nmax)
    rng = np.random.default_rng(7)
    v0, L0 = 0.1 * rng.normal(size=(n, 3)), 0.05 * rng.normal(size=(n, 3))
    v, L = dev(v0), dev(L0)
    mask = torch.ones(n, dtype=torch.int32, device="cuda:0")
    ng = sp.borders_device(n, nmax, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    npairs = sp.neighbor_build_device(n, ng, x.data_ptr(), sh.data_ptr())
    f = torch.zeros(nmax, 3, dtype=torch.float64, device="cuda:0")
    tq = torch.zeros_like(f)
    ev = torch.zeros(7, dtype=torch.float64, device="cuda:0")
    g = np.array([0.0, 0.0, -1.0])
    dt = 2e-3

    def force():
        f.zero_(); tq.zero_()
        sp.forward_device(x.data_ptr(), q.data_ptr())
        sp.compute_device(n, ng, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(), tq.data_ptr(),
                          eflag=True, ev=ev.data_ptr())
        sp.reverse_device(f.data_ptr(), tq.data_ptr())
        sp.post_force_device(n, g, 0.05, 0.02, v.data_ptr(), q.data_ptr(), L.data_ptr(), sh.data_ptr(), mask.data_ptr(),
                             f.data_ptr(), tq.data_ptr())
    force()
    for _ in range(3):
        sp.nve_device(0, n, dt, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), f.data_ptr(), tq.data_ptr(),
                      sh.data_ptr(), mask.data_ptr())
        force()
        sp.nve_device(1, n, dt, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), f.data_ptr(), tq.data_ptr(),
                      sh.data_ptr(), mask.data_ptr())
    torch.cuda.synchronize()

    # the same with oracle pieces
    mp = np.array([oracle.mass_props(case["lmax"], a) for a in case["shapes"]])
    cmax = 2 * case["rmax"].max() + case["skin"]
    xo = case["x"].copy()
    own, shift = oracle.borders(xo, case["lo"], case["hi"], (1, 1, 1), cmax)
    qo, vo, Lo = case["quat"].copy(), v0.copy(), L0.copy()
    sha = np.concatenate([case["shtype"], case["shtype"][own]])
    tya = np.ones(n + own.size, dtype=np.int32)
    tag = np.concatenate([np.arange(n), own]).astype(np.int32)
    offs, jl = oracle.half_list(n, np.concatenate([xo, xo[own] + shift * case["box"]]), sha, tag, case["rmax"], case["skin"])
    assert jl.size == npairs
    K, E = coeff_tables(1, 1000.0, 1.25)
    mk = np.ones(n, dtype=np.int32)
    eng = [0.0]

    def oforce():
        xa = np.concatenate([xo, xo[own] + shift * case["box"]])
        qa = np.concatenate([qo, qo[own]])
        o = oracle.compute([(case["lmax"], a, r) for a, r in zip(case["shapes"], case["rmax"])], K, E, nq, n, xa, qa, tya,
                           sha, np.arange(n, dtype=np.int32), offs, jl, eflag=True, nthreads=8)
        fo, to = o["f"][:n].copy(), o["torque"][:n].copy()
        np.add.at(fo, own, o["f"][n:])
        np.add.at(to, own, o["torque"][n:])
        eng[0] += o["eng_virial"][0]
        oracle.post_force(mp, rho, g, 0.05, 0.02, vo, qo, Lo, case["shtype"], mk, fo, to)
        return fo, to
    fo, to = oforce()
    for _ in range(3):
        oracle.nve(0, dt, mp, rho, xo, vo, qo, Lo, fo, to, case["shtype"], mk)
        fo, to = oforce()
        oracle.nve(1, dt, mp, rho, xo, vo, qo, Lo, fo, to, case["shtype"], mk)
    assert np.abs(fo).max() > 1.0
    assert rel_err(f[:n].cpu().numpy(), fo) < 1e-9
    assert rel_err(tq[:n].cpu().numpy(), to, max(np.abs(fo).max(), np.abs(to).max())) < 1e-9
    assert rel_err(x[:n].cpu().numpy(), xo) < 1e-12
    assert rel_err(v.cpu().numpy(), vo) < 1e-10
    assert rel_err(q[:n].cpu().numpy(), qo) < 1e-12
    assert rel_err(L.cpu().numpy(), Lo) < 1e-10
    assert abs(ev[0].item() - eng[0]) < 1e-9 * eng[0]
    sp.close()


def _nve_run(sp, case, dt, nsteps):
    """Device-resident NVE loop; returns (pe0, pe1, ke1, momentum, momentum scale, rebuilds)."""
    import torch
    n = case["n"]
    nmax = 4 * n
    x, q, ty, sh = _device_rows(case, nmax)
    v = torch.zeros(n, 3, dtype=torch.float64, device="cuda:0")
    L = torch.zeros_like(v)
    mask = torch.ones(n, dtype=torch.int32, device="cuda:0")
    f = torch.zeros(nmax, 3, dtype=torch.float64, device="cuda:0")
    tq = torch.zeros_like(f)
    ev = torch.zeros(7, dtype=torch.float64, device="cuda:0")
    ke = torch.zeros(3, dtype=torch.float64, device="cuda:0")
    g0 = np.zeros(3)
    state = dict(ng=0, builds=0)

    def rebuild():
        state["ng"] = sp.borders_device(n, nmax, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
        sp.neighbor_build_device(n, state["ng"], x.data_ptr(), sh.data_ptr())
        state["builds"] += 1

    def force():
        f.zero_(); tq.zero_(); ev.zero_()
        sp.forward_device(x.data_ptr(), q.data_ptr())
        sp.compute_device(n, state["ng"], x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(), tq.data_ptr(),
                          eflag=True, ev=ev.data_ptr())
        sp.reverse_device(f.data_ptr(), tq.data_ptr())

    def total_energy():
        ke.zero_()
        sp.energies_device(n, g0, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), sh.data_ptr(), mask.data_ptr(), ke.data_ptr())
        torch.cuda.synchronize()
        return ev[0].item(), ke[0].item() + ke[1].item()
    rebuild()
    force()
    pe0, ke0 = total_energy()
    assert ke0 == 0.0 and pe0 > 0
    for step in range(nsteps):
        sp.nve_device(0, n, dt, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), f.data_ptr(), tq.data_ptr(), sh.data_ptr(), mask.data_ptr())
        if sp.neighbor_check_device(n, x.data_ptr()):
            rebuild()
        force()
        sp.nve_device(1, n, dt, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), f.data_ptr(), tq.data_ptr(), sh.data_ptr(), mask.data_ptr())
    pe1, ke1 = total_energy()
    m = sp.body(0)[0]
    return pe0, pe1, ke1, m * v.sum(0).cpu().numpy(), m * v.abs().sum().item(), state["builds"]


def test_periodic_nve_conserves_momentum_and_energy(oracle):
    """No gravity, no damping, fully periodic bed relaxing from rest.  Linear momentum is conserved to
    rounding (F_j = -F_i per pair, whatever the quadrature error).  Total energy: the force is the exact
    gradient of the overlap volume but the reported energy uses the ray volume of SPEC §2.5, which
    undershoots in deep overlaps, so the sum is conserved only to ~1e-3 on a shallow-contact bed — measured
    (tools/energy_drift.py) independent of dt and of n_q; the integrator itself shows in the dt comparison."""
    case = _periodic_case(oracle, 512, (1, 1, 1), 63, lmax=4, nshapes=1, skin=0.3, jitter=0.1)
    sp = make_ctx(case["shapes"], case["lmax"], nq=12, kn=200.0, expo=1.5)
    sp.set_box(case["lo"], case["hi"], (1, 1, 1), case["skin"])
    pe0, pe1, ke1, p, pscale, builds = _nve_run(sp, case, 2e-3, 150)
    assert builds >= 2                             # the list was rebuilt on the way
    assert ke1 > 0.2 * pe0                         # the bed did relax: energy moved from contact to motion
    assert np.abs(p).max() < 1e-11 * pscale
    assert abs((pe1 + ke1) - pe0) < 5e-3 * pe0
    pe0b, pe1b, ke1b, _, _, _ = _nve_run(sp, case, 1e-3, 300)
    assert abs(pe0b - pe0) < 1e-12 * pe0                  # atomics: summation order differs run to run
    assert abs((pe1b + ke1b) - (pe1 + ke1)) < 3e-4 * pe0     # halving dt changes the total by far less than that
    sp.close()


def test_settled_bed_fixture_through_the_device_pipeline():
    """BASELINE configs[0] (gravity-settled L = 4 bed, tests/golden/settled_cfg1_L4.npz): ghosts and half list
    built on the device, pair forces, reverse — against the oracle's committed numbers."""
    import os
    import torch
    from shpair import ShPair
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "settled_cfg1_L4.npz"))
    lmax, nq, n = int(g["lmax"]), int(g["nq"]), g["x"].shape[0]
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, lmax, g["anm"][0])
    sp.coeff(1, 1, float(g["kn"]), float(g["exponent"]))
    assert abs(sp.rmax(0) - g["rmax"][0]) < 1e-14
    sp.set_box(g["lo"], g["hi"], g["periodic"], float(g["skin"]))
    case = dict(n=n, x=g["x"], quat=g["quat"], type=np.ones(n, np.int32), shtype=np.zeros(n, np.int32))
    nmax = 3 * n
    x, q, ty, sh = _device_rows(case, nmax)
    ng = sp.borders_device(n, nmax, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    assert ng == g["ghost_owner"].size
    npairs = sp.neighbor_build_device(n, ng, x.data_ptr(), sh.data_ptr())
    offs, jl = sp.copy_neighbors(n, npairs)
    assert np.array_equal(offs, g["offsets"]) and np.array_equal(jl, g["jlist"])
    f = torch.zeros(nmax, 3, dtype=torch.float64, device="cuda:0")
    tq = torch.zeros_like(f)
    ev = torch.zeros(7, dtype=torch.float64, device="cuda:0")
    sp.set_option("count", 1)
    sp.compute_device(n, ng, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(), tq.data_ptr(),
                      eflag=True, ev=ev.data_ptr())
    sp.reverse_device(f.data_ptr(), tq.data_ptr())
    torch.cuda.synchronize()
    st = sp.stats()
    assert [st["n_candidates"], st["n_contact"], st["n_touching"]] == g["counts"].tolist()
    fs = np.abs(g["f"]).max()
    assert np.abs(f[:n].cpu().numpy() - g["f"]).max() < 1e-9 * fs
    assert np.abs(tq[:n].cpu().numpy() - g["torque"]).max() < 1e-9 * max(fs, np.abs(g["torque"]).max())
    assert abs(ev[0].item() - float(g["energy"])) < 1e-9 * float(g["energy"])
    sp.close()


def test_half_list_with_explicit_tags_and_atoms_outside_an_open_box(oracle):
    """Tags decide which particle of a pair is the integrated one (not the row index); in non-periodic
    directions particles beyond the box are clamped into the boundary cells and still find their pairs."""
    import torch
    case = _periodic_case(oracle, 512, (1, 0, 0), 64, nshapes=2)
    n = case["n"]
    rng = np.random.default_rng(5)
    case["x"][rng.choice(n, 40, replace=False), 1] += rng.uniform(3.0, 9.0, 40)      # beyond hi in the open y direction
    case["x"][rng.choice(n, 40, replace=False), 2] -= rng.uniform(3.0, 9.0, 40)      # below lo in z
    tag0 = rng.permutation(n).astype(np.int32) + 1000
    sp = make_ctx(case["shapes"], case["lmax"])
    sp.set_box(case["lo"], case["hi"], case["periodic"], case["skin"])
    nmax = 3 * n
    x, q, ty, sh = _device_rows(case, nmax)
    tag = torch.zeros(nmax, dtype=torch.int32, device="cuda:0")
    tag[:n] = dev(tag0)
    ng = sp.borders_device(n, nmax, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), tag=tag.data_ptr())
    cmax = 2 * case["rmax"].max() + case["skin"]
    xw = case["x"].copy()
    own, shift = oracle.borders(xw, case["lo"], case["hi"], case["periodic"], cmax)
    assert ng == own.size and ng > 0
    tags = np.concatenate([tag0, tag0[own]])
    assert np.array_equal(tag[: n + ng].cpu().numpy(), tags)
    npairs = sp.neighbor_build_device(n, ng, x.data_ptr(), sh.data_ptr(), tag=tag.data_ptr())
    offs, jl = sp.copy_neighbors(n, npairs)
    sha = np.concatenate([case["shtype"], case["shtype"][own]])
    o_offs, o_jl = oracle.half_list(n, x[: n + ng].cpu().numpy(), sha, tags, case["rmax"], case["skin"])
    assert np.array_equal(offs, o_offs) and np.array_equal(jl, o_jl) and npairs > n
    sp.close()


def test_step_entry_points_with_no_particles_and_far_wraps(oracle):
    import torch
    case = _periodic_case(oracle, 216, (1, 1, 1), 65)
    n = case["n"]
    sp = make_ctx(case["shapes"], case["lmax"])
    sp.set_box(case["lo"], case["hi"], (1, 1, 1), case["skin"])
    # empty system: every entry point is a no-op that succeeds
    z = torch.zeros(8, 4, dtype=torch.float64, device="cuda:0")
    zi = torch.zeros(8, dtype=torch.int32, device="cuda:0")
    assert sp.borders_device(0, 8, z.data_ptr(), z.data_ptr(), zi.data_ptr(), zi.data_ptr()) == 0
    assert sp.neighbor_build_device(0, 0, z.data_ptr(), zi.data_ptr()) == 0
    assert not sp.neighbor_check_device(0, z.data_ptr())
    sp.forward_device(z.data_ptr(), z.data_ptr())
    sp.reverse_device(z.data_ptr(), z.data_ptr())
    sp.nve_device(0, 0, 1e-3, *([z.data_ptr()] * 6), zi.data_ptr(), zi.data_ptr())
    sp.post_force_device(0, [0, 0, -1], 0.0, 0.0, z.data_ptr(), z.data_ptr(), z.data_ptr(), zi.data_ptr(), zi.data_ptr(),
                         z.data_ptr(), z.data_ptr())
    sp.compute_device(0, 0, z.data_ptr(), z.data_ptr(), zi.data_ptr(), zi.data_ptr(), z.data_ptr(), z.data_ptr())
    torch.cuda.synchronize()
    # particles several box lengths away are wrapped back in one go; the images agree with the oracle's
    far = case["x"].copy()
    far[:50] += case["box"] * np.array([3, -2, 5])
    case2 = dict(case, x=far)
    x, q, ty, sh = _device_rows(case2, 4 * n)
    ng = sp.borders_device(n, 4 * n, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    xw = far.copy()
    own, shift = oracle.borders(xw, case["lo"], case["hi"], (1, 1, 1), 2 * case["rmax"].max() + case["skin"])
    assert ng == own.size
    assert np.abs(x[:n].cpu().numpy() - xw).max() < 1e-12
    sp.close()


@pytest.mark.parametrize("use_graph,rule", [(False, 0), (True, 0), (True, 1)])
def test_native_run_loop_matches_the_python_driver(oracle, use_graph, rule):
    """shstep_run_device (C++ loop in the library, optionally replayed from hipGraphs) takes the same steps as
    the call-by-call driver: same rebuild decisions, same trajectory (forces are atomically summed, so
    equality is to rounding, not bitwise)."""
    from shpair.run import DeviceRun
    case = _periodic_case(oracle, 512, (1, 1, 0), 66, lmax=4, nshapes=2, skin=0.2, jitter=0.15)
    runs = []
    for native in (False, True):
        sp = make_ctx(case["shapes"], case["lmax"], nq=8, kn=300.0, expo=1.25, rho=[1.0, 1.4])
        sp.set_option("rule", rule)                      # 1: the weighted cap rule inside the captured graphs
        r = DeviceRun(sp, case["x"], case["quat"], case["shtype"], case["lo"], case["hi"], case["periodic"], case["skin"],
                      dt=2e-3, gravity=(0.0, 0.0, -2.0), gamma_t=0.3, gamma_r=0.1)
        if native:
            r.run_native(120, use_graph=use_graph)
        else:
            r.run(120)
        r.force(eflag=True)
        runs.append((r, sp, r.energies()))
    (a, spa, ea), (b, spb, eb) = runs
    n = case["n"]
    assert a.builds == b.builds and a.builds >= 3 and a.nghost == b.nghost
    assert np.abs(a.x[:n].cpu().numpy() - b.x[:n].cpu().numpy()).max() < 1e-10
    assert np.abs(a.v.cpu().numpy() - b.v.cpu().numpy()).max() < 1e-9
    assert np.abs(a.q[:n].cpu().numpy() - b.q[:n].cpu().numpy()).max() < 1e-10
    assert abs(ea[0] - eb[0]) < 1e-9 * abs(ea[0]) and abs(ea[1] - eb[1]) < 1e-8 * abs(ea[1])
    spa.close()
    spb.close()


def test_native_run_loop_argument_checks(oracle):
    from shpair.capi import ShPairError, StepArrays
    case = _periodic_case(oracle, 216, (1, 1, 1), 67)
    sp = make_ctx(case["shapes"], case["lmax"])
    a = StepArrays()
    a.nlocal, a.nmax, a.check_every, a.dt = 10, 20, 1, 1e-3
    with pytest.raises(ShPairError):                  # null arrays
        sp.run_device(a, 5, 0)
    a.check_every = 0
    with pytest.raises(ShPairError):
        sp.run_device(a, 5, 0)
    sp.close()


def test_random_boxes_ghosts_and_lists_match_oracle(oracle):
    """Property test over random boxes (any mix of periodic / open directions, edges down to the minimum
    2 c_max, particles outside open faces, one to a few hundred particles, dense and dilute): the
    device-built ghosts and half list equal the oracle's, entry for entry."""
    import torch
    from shpair import shapes
    rng = np.random.default_rng(2026)
    lmax = 3
    shp = [shapes.random_shape(lmax, 70, amp=0.15), shapes.random_shape(lmax, 71, amp=0.3)]
    rmax = np.array([oracle.shape_rmax(lmax, a) for a in shp])
    sp = make_ctx(shp, lmax)
    checked_pairs = 0
    for trial in range(40):
        periodic = tuple(int(v) for v in rng.integers(0, 2, 3))
        skin = float(rng.choice([0.0, 0.05, 0.4]))
        cmax = 2 * rmax.max() + skin
        edge = np.where(np.array(periodic, bool), rng.uniform(2.0 * cmax + 1e-9, 5 * cmax, 3), rng.uniform(0.3, 6 * cmax, 3))
        lo = rng.uniform(-5, 5, 3)
        hi = lo + edge
        n = int(rng.choice([1, 2, 7, 60, 300]))
        x = lo + rng.uniform(-0.2, 1.2, (n, 3)) * edge      # some outside: wrapped if periodic, clamped into edge cells if not
        if trial % 5 == 0:
            x[: n // 2] = x[0] + rng.normal(0, 0.3, (n // 2, 3))   # a dense clump: long rows
        case = dict(n=n, x=x, quat=random_quats(rng, n), type=np.ones(n, np.int32),
                    shtype=rng.integers(0, 2, n).astype(np.int32))
        sp.set_box(lo, hi, periodic, skin)
        nmax = 27 * n + 8
        xd, qd, tyd, shd = _device_rows(case, nmax)
        ng = sp.borders_device(n, nmax, xd.data_ptr(), qd.data_ptr(), tyd.data_ptr(), shd.data_ptr())
        xw = x.copy()
        own, shift = oracle.borders(xw, lo, hi, periodic, cmax)
        assert ng == own.size, (trial, periodic)
        xa = xd[: n + ng].cpu().numpy()
        assert np.abs(xa - np.concatenate([xw, xw[own] + shift * edge])).max() < 1e-12
        npairs = sp.neighbor_build_device(n, ng, xd.data_ptr(), shd.data_ptr())
        offs, jl = sp.copy_neighbors(n, npairs)
        tag = np.concatenate([np.arange(n), own]).astype(np.int32)
        sha = np.concatenate([case["shtype"], case["shtype"][own]])
        o_offs, o_jl = oracle.half_list(n, xa, sha, tag, rmax, skin)
        assert np.array_equal(offs, o_offs) and np.array_equal(jl, o_jl), (trial, periodic, n)
        checked_pairs += npairs
    assert checked_pairs > 5000
    torch.cuda.synchronize()
    sp.close()
