"""The C++ host path: PairSH (lammps-spherharm_amd/lammps/pair_sh.cpp) driven through
settings -> coeff -> init_style -> init_one -> compute by a minimal C++ host built
against the stub headers (this image has no LAMMPS tree).  CPU: the adapter compiles
and, without a GPU, refuses to run.  GPU: forces/torques/energy/virial equal the oracle."""
import os
import subprocess

import numpy as np
import pytest

from common import make_case, coeff_tables, oracle_compute

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LMP = os.path.join(ROOT, "lammps-spherharm_amd", "lammps")
HOST = os.path.join(LMP, "build", "lammps_host")
# one host per LAMMPS API generation the stub models (lammps/stub/lammps_stub.h SHPAIR_STUB_GEN, lammps/sh_lammps_compat.h):
# the adapter compiled with -DSHPAIR_LAMMPS_VERSION of a release inside the generation, against a stub that offers only
# that generation's calls.  [PRIOR] dates; gen 4 = current LAMMPS, no define.
GEN_HOSTS = {4: HOST, **{g: os.path.join(LMP, "build", f"lammps_host_gen{g}") for g in (3, 2, 1, 0)}}
GEN_VERSION = {4: 20230802, 3: 20210929, 2: 20201029, 1: 20200303, 0: 20181212}
HOST_OLD = GEN_HOSTS[1]


def build_host():
    subprocess.check_call(["make", "-C", LMP], stdout=subprocess.DEVNULL)
    assert all(os.path.exists(h) for h in GEN_HOSTS.values())


def write_inputs(tmp_path, case, nlocal, newton, eflag):
    b = case["bed"]
    n = case["n"]
    bedf = tmp_path / "bed.txt"
    with open(bedf, "w") as fp:
        fp.write(f"{nlocal} {n - nlocal} 1 {int(newton)} {int(eflag)}\n")
        for i in range(n):
            fp.write(" ".join(repr(float(v)) for v in (*b["x"][i], *b["quat"][i])) +
                     f" {b['type'][i]} {b['shtype'][i]}\n")
        of, jl = case["offsets"], case["jlist"]
        rows = [ii for ii in range(len(case["ilist"]))]
        fp.write(f"{len(rows)}\n")
        for ii in rows:
            js = jl[of[ii]:of[ii + 1]]
            fp.write(f"{case['ilist'][ii]} {len(js)} " + " ".join(str(int(j)) for j in js) + "\n")
    from shpair import shapes as shp_mod
    shapes = []
    for s, a in enumerate(case["shapes"]):
        p = tmp_path / f"shape{s}.txt"
        shp_mod.write_shape_file(p, case["lmax"], a)
        # the adapter's reader takes comments and blank lines too
        txt = open(p).read().split("\n")
        open(p, "w").write("# shape %d of the test bed\n\n" % s + txt[0] + "   # lmax\n" + "\n".join(txt[1:]))
        if s % 2 == 1:
            # ... and tables over the whole range m = -n..n without an lmax line (PairSH::load_shapes, [PRIOR] layout)
            from test_shape_io import _full_range_text
            open(p, "w").write("# whole range of m, no header\n" + _full_range_text(case["lmax"], a))
        shapes.append(str(p))
    return str(bedf), shapes


def test_adapter_compiles_and_has_no_cpu_fallback(tmp_path, oracle, gpu_available):
    build_host()
    if gpu_available:
        pytest.skip("GPU present: the host run is covered by the -m gpu test")
    case = make_case(20, 4, 1, seed=30, rmax_fn=oracle.shape_rmax)
    bedf, shapes = write_inputs(tmp_path, case, 20, True, False)
    r = subprocess.run([HOST, bedf, str(tmp_path / "out.txt"), "8", "1000.0", "1.0", *shapes],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "cannot open HIP device" in r.stderr and "no CPU fallback" in r.stderr


def test_every_api_generation_compiles_only_with_its_own_switches():
    """The boundary is the product: pair_sh.cpp / fix_nve_sh.cpp against each of the five stub generations, with the
    switches derived from the version of each — only the diagonal may compile (the stub offers one generation's calls
    and nothing else, as a real tree does), so a wrong SHPAIR_LAMMPS_VERSION is a compile error, not a silent mismatch."""
    inc = ["-Istub", "-I../../include"]
    for g in GEN_VERSION:
        for gv, ver in GEN_VERSION.items():
            ok = all(subprocess.run(["g++", "-std=c++17", "-fsyntax-only", *inc, f"-DSHPAIR_STUB_GEN={g}", f"-DSHPAIR_LAMMPS_VERSION={ver}",
                                     src], cwd=LMP, capture_output=True).returncode == 0 for src in ("pair_sh.cpp", "fix_nve_sh.cpp"))
            assert ok == (g == gv), (g, ver, ok)
    # no define at all = current LAMMPS; one switch flipped by hand is independent of the others
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", *inc, "pair_sh.cpp"], cwd=LMP, capture_output=True)
    assert r.returncode == 0
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", *inc, "-DSHPAIR_STUB_GEN=3", "-DSHPAIR_LMP_NEIGH_REQUEST=1",
                        "-DSHPAIR_LMP_FORWARD_COMM_PAIR=1", "pair_sh.cpp"], cwd=LMP, capture_output=True)
    assert r.returncode == 0


@pytest.mark.gpu
@pytest.mark.parametrize("newton,expo,gen", [(True, 1.25, 4), (False, 1.0, 4), (True, 1.25, 3), (True, 1.25, 2), (True, 1.25, 1), (True, 1.25, 0)],
                         ids=["newton", "newton_off", "gen3_2021", "gen2_2020", "gen1_2019", "gen0_2018"])
def test_pairsh_adapter_matches_oracle(tmp_path, oracle, newton, expo, gen):
    """Every API generation: utils::bounds / Force::bounds, add_request / request, ev_init / ev_setup — and, in the
    generations without 2-d custom arrays (2, 1, 0), the shape index as a custom INTEGER VECTOR (fix property/atom
    i_shtype, two-argument find_custom) with the quaternions from the atom style."""
    build_host()
    host = GEN_HOSTS[gen]
    case = make_case(260, 6, 2, seed=31, rmax_fn=oracle.shape_rmax)
    nlocal = 260 if newton else 130
    if not newton:
        case = dict(case)
        case["ilist"] = case["ilist"][:nlocal]
        case["jlist"] = case["jlist"][:case["offsets"][nlocal]]
        case["offsets"] = case["offsets"][:nlocal + 1]
    bedf, shapes = write_inputs(tmp_path, case, nlocal, newton, True)
    out = tmp_path / "out.txt"
    env = dict(os.environ)
    if gen <= 2 or gen == 3:
        env["LAMMPS_HOST_SHTYPE_CUSTOM"] = "1"
    r = subprocess.run([host, bedf, str(out), "12", "750.0", repr(expo), *shapes], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    assert f"stub generation {gen}" in r.stderr
    lines = open(out).read().split("\n")
    cut, e1, e2 = (float(v) for v in lines[0].split()[:3])
    assert int(lines[0].split()[3]) == 0      # quat from the atom style: the pair style does not forward it itself
    vir = np.array([float(v) for v in lines[1].split()])
    ft = np.array([[float(v) for v in ln.split()] for ln in lines[2:] if ln.strip()])
    K, E = coeff_tables(1, 750.0, expo)
    o = oracle_compute(oracle, case, 12, K, E, nlocal=nlocal, newton_pair=newton, eflag=True, vflag=True, want_peratom=True)
    fs = np.abs(o["f"]).max()
    assert abs(cut - 2 * max(case["rmax"])) < 1e-14
    assert np.abs(ft[:, :3] - o["f"]).max() < 1e-9 * fs
    assert np.abs(ft[:, 3:] - o["torque"]).max() < 1e-9 * max(fs, np.abs(o["torque"]).max())
    assert abs(e1 - o["eng_virial"][0]) < 1e-9 * o["eng_virial"][0] and abs(e2 - e1) < 1e-11 * e1
    assert np.abs(vir - o["eng_virial"][1:]).max() < 1e-9 * np.abs(o["eng_virial"][1:]).max()
    # per-atom tallies through Pair::eatom / vatom (eflag, vflag bit 2)
    pa = np.loadtxt(str(out) + ".peratom")
    assert np.abs(pa[:, 0] - o["eatom"]).max() < 1e-9 * o["eatom"].max()
    assert np.abs(pa[:, 1:] - o["vatom"]).max() < 1e-9 * np.abs(o["vatom"]).max()


@pytest.mark.gpu
def test_pairsh_forwards_custom_quaternions_to_ghosts(tmp_path, oracle):
    """Orientations kept by `fix property/atom d2_quat 4 ghost yes` reach ghosts only at reneighbourings in stock
    LAMMPS, while the integrator turns the owners every step.  PairSH forwards them itself (comm_forward = 4,
    pack/unpack_forward_comm, Comm::forward_comm(this) at the top of compute): the bed handed to the host has STALE
    ghost orientations, the forces must be those of the owners' current ones."""
    build_host()
    case = make_case(260, 6, 2, seed=35, rmax_fn=oracle.shape_rmax)
    n, nlocal = case["n"], 200
    b = case["bed"]
    rng = np.random.default_rng(3)
    owners = rng.integers(0, nlocal, n - nlocal).astype(np.int32)
    truth = dict(case)
    tb = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in b.items()}
    tb["quat"][nlocal:] = tb["quat"][owners]           # what the ghosts' orientations must be
    tb["shtype"][nlocal:] = tb["shtype"][owners]
    truth["bed"] = tb
    # rows of the half list: owned i only (newton on: ghost forces are computed too)
    truth["ilist"] = case["ilist"][:nlocal]
    truth["jlist"] = case["jlist"][:case["offsets"][nlocal]]
    truth["offsets"] = case["offsets"][:nlocal + 1]
    stale = dict(truth)
    sb = {k: (v.copy() if hasattr(v, "copy") else v) for k, v in tb.items()}
    sb["quat"][nlocal:] = np.array([1.0, 0.0, 0.0, 0.0])   # as after borders(), before the owners were turned
    stale["bed"] = sb
    bedf, shapes = write_inputs(tmp_path, stale, nlocal, True, False)
    gof = tmp_path / "ghost_owners.txt"
    np.savetxt(gof, owners, fmt="%d")
    out = tmp_path / "out.txt"
    env = dict(os.environ, LAMMPS_HOST_GHOST_OWNERS=str(gof))
    r = subprocess.run([HOST, bedf, str(out), "12", "750.0", "1.25", *shapes], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().split("\n")
    assert int(lines[0].split()[3]) == 2                 # one forward per compute() call
    ft = np.array([[float(v) for v in ln.split()] for ln in lines[2:] if ln.strip()])
    K, E = coeff_tables(1, 750.0, 1.25)
    o = oracle_compute(oracle, truth, 12, K, E, nlocal=nlocal, newton_pair=True)
    o_stale = oracle_compute(oracle, stale, 12, K, E, nlocal=nlocal, newton_pair=True)
    fs = np.abs(o["f"]).max()
    assert np.abs(o_stale["f"] - o["f"]).max() > 1e-3 * fs       # the stale orientations would have been visibly wrong
    assert np.abs(ft[:, :3] - o["f"]).max() < 1e-9 * fs
    assert np.abs(ft[:, 3:] - o["torque"]).max() < 1e-9 * max(fs, np.abs(o["torque"]).max())


@pytest.mark.gpu
def test_a_tree_without_custom_arrays_says_where_the_quaternions_must_come_from(tmp_path, oracle):
    """Generations 2, 1, 0 (two-argument find_custom): there is no `fix property/atom d2_quat 4`; the adapter accepts
    atom->extract("quat") only and init_style says so."""
    build_host()
    assert "predates 2-d custom per-atom arrays" in open(os.path.join(LMP, "sh_lammps_compat.h")).read()
    case = make_case(40, 4, 1, seed=37, rmax_fn=oracle.shape_rmax)
    bedf, shapes = write_inputs(tmp_path, case, case["n"], True, False)
    gof = tmp_path / "ghost_owners.txt"
    open(gof, "w").write("")
    r = subprocess.run([GEN_HOSTS[2], bedf, str(tmp_path / "out.txt"), "8", "750.0", "1.25", *shapes], capture_output=True, text=True,
                       env=dict(os.environ, LAMMPS_HOST_GHOST_OWNERS=str(gof)), timeout=300)
    assert r.returncode == 6 and "no 2-d custom per-atom arrays" in r.stderr     # the scaffold cannot even offer them


@pytest.mark.gpu
@pytest.mark.parametrize("host", [HOST, GEN_HOSTS[3]], ids=["gen4_forward_comm", "gen3_forward_comm_pair"])
def test_pairsh_forward_comm_is_collective(tmp_path, oracle, host):
    """A rank WITHOUT ghosts (its atoms may still be ghosts of a neighbour, which waits for their orientations) must
    enter Comm::forward_comm like every other rank when the quaternions live in a custom property: one call per
    compute() whatever nghost is (the call used to be guarded by atom->nghost > 0)."""
    build_host()
    case = make_case(120, 4, 1, seed=36, rmax_fn=oracle.shape_rmax)
    n = case["n"]
    bedf, shapes = write_inputs(tmp_path, case, n, True, False)     # nghost = 0
    gof = tmp_path / "ghost_owners.txt"
    open(gof, "w").write("")
    out = tmp_path / "out.txt"
    env = dict(os.environ, LAMMPS_HOST_GHOST_OWNERS=str(gof))
    r = subprocess.run([host, bedf, str(out), "8", "750.0", "1.25", *shapes], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    lines = open(out).read().split("\n")
    assert int(lines[0].split()[3]) == 2
    ft = np.array([[float(v) for v in ln.split()] for ln in lines[2:] if ln.strip()])
    K, E = coeff_tables(1, 750.0, 1.25)
    o = oracle_compute(oracle, case, 8, K, E, nlocal=n, newton_pair=True)
    assert np.abs(ft[:, :3] - o["f"]).max() < 1e-9 * np.abs(o["f"]).max()


@pytest.mark.gpu
def test_fix_nve_sh_adapter_matches_oracle(tmp_path, oracle):
    """FixNVESH + PairSH stepped by the C++ host in Verlet::run order from rest, against the same steps
    made of oracle pieces (sho_compute + sho_nve)."""
    build_host()
    case = make_case(200, 4, 2, seed=33, rmax_fn=oracle.shape_rmax)
    n = case["n"]
    bedf, shapes = write_inputs(tmp_path, case, n, True, False)
    out = tmp_path / "out.txt"
    nsteps, dt, rho = 4, 2e-3, 1.5
    env = dict(os.environ, LAMMPS_HOST_NSTEPS=str(nsteps), LAMMPS_HOST_DT=repr(dt))
    r = subprocess.run([HOST, bedf, str(out), "8", "500.0", "1.25", *shapes], capture_output=True, text=True, env=env,
                       timeout=300)
    assert r.returncode == 0, r.stderr
    traj = np.loadtxt(str(out) + ".traj")
    b = case["bed"]
    K, E = coeff_tables(1, 500.0, 1.25)
    mp = np.array([oracle.mass_props(case["lmax"], a) for a in case["shapes"]])
    # `density 1.5` names one value: shape 0 gets it, shape 1 keeps the default 1
    dens = np.array([rho, 1.0])
    x, q = b["x"].copy(), b["quat"].copy()
    v, L = np.zeros((n, 3)), np.zeros((n, 3))
    mask = np.ones(n, dtype=np.int32)
    sh_list = [(case["lmax"], a, rm) for a, rm in zip(case["shapes"], case["rmax"])]

    def force():
        o = oracle.compute(sh_list, K, E, 8, n, x, q, b["type"], b["shtype"], case["ilist"], case["offsets"], case["jlist"],
                           nthreads=8)
        return o["f"], o["torque"]
    f, t = force()
    for _ in range(nsteps):
        oracle.nve(0, dt, mp, dens, x, v, q, L, f, t, b["shtype"], mask)
        f, t = force()
        oracle.nve(1, dt, mp, dens, x, v, q, L, f, t, b["shtype"], mask)
    assert np.abs(v).max() > 1e-3
    assert np.abs(traj[:, 0:3] - x).max() < 1e-12
    assert np.abs(traj[:, 3:6] - v).max() < 1e-10 * np.abs(v).max()
    assert np.abs(traj[:, 6:10] - q).max() < 1e-12
    assert np.abs(traj[:, 10:13] - L).max() < 1e-10 * np.abs(L).max()
