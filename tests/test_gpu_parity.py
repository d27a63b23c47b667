"""GPU parity tests proper: the HIP path, called through the C ABI
(include/shpair.h via ctypes), against the CPU oracle on identical seeded
inputs and against the committed golden vectors.

Tolerance (docs/SPEC.md §4): |dF| <= 1e-9 max|F|, same for torque scaled by
max(|F|,|tau|); the task's bar is 1e-6.  Both sides are FP64 and differ only
in operation order / FMA contraction; measured differences are ~1e-15.
"""
import ctypes
import glob
import os

import numpy as np
import pytest

from common import make_case, coeff_tables, oracle_compute, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-9
HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "cfg*.npz")))


def make_ctx(case, nq, K, E, rmax=None):
    from shpair import ShPair
    sp = ShPair(0)
    sp.settings(nq)
    nt = K.shape[0] - 1
    sp.set_ntypes(nt, len(case["shapes"]))
    for s, a in enumerate(case["shapes"]):
        sp.set_shape(s, case["lmax"], a, 0.0 if rmax is None else rmax[s])
    for i in range(1, nt + 1):
        for j in range(1, nt + 1):
            sp.coeff(i, j, K[i, j], E[i, j])
    sp.set_neighbors_csr(case["ilist"], case["offsets"], case["jlist"])
    return sp


def check(f, tq, o):
    fs = np.abs(o["f"]).max()
    ts = max(fs, np.abs(o["torque"]).max())
    assert fs > 0
    assert rel_err(f, o["f"], fs) < TOL
    assert rel_err(tq, o["torque"], ts) < TOL


@pytest.mark.parametrize("lmax,nq,nshapes,expo,fv", [
    (0, 8, 1, 1.0, 1), (1, 6, 1, 1.0, 0), (2, 8, 2, 1.5, 0), (3, 7, 1, 1.0, 1), (4, 10, 1, 1.0, 0),
    (5, 9, 3, 1.25, 0), (6, 16, 1, 1.0, 0), (6, 16, 4, 1.0, 1), (6, 16, 1, 1.25, 0), (7, 12, 1, 1.0, 1),
    (8, 16, 2, 2.0, 0), (9, 11, 1, 1.0, 1), (10, 20, 1, 1.5, 0), (11, 13, 1, 1.0, 1), (12, 32, 1, 1.0, 0),
    (12, 32, 1, 1.25, 0), (13, 8, 1, 1.0, 1), (16, 8, 2, 1.25, 0), (20, 6, 1, 1.0, 1)])
def test_forces_and_torques_match_oracle(oracle, lmax, nq, nshapes, expo, fv):
    """Every compiled order 0..12, the run-time-order kernel (13..20), odd nq (ragged last slab)."""
    case = make_case(220 if lmax <= 12 else 120, lmax, nshapes, seed=lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, expo)
    sp = make_ctx(case, nq, K, E)
    sp.set_option("force_volume", fv)
    sp.set_option("count", 1)
    b = case["bed"]
    f, tq, eng, vir = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    st = sp.stats()
    o = oracle_compute(oracle, case, nq, K, E, eflag=True, vflag=True, force_volume=bool(fv))
    check(f, tq, o)
    assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
    assert np.abs(vir - o["eng_virial"][1:]).max() < TOL * np.abs(o["eng_virial"][1:]).max()
    assert (st["n_candidates"], st["n_contact"], st["n_touching"]) == tuple(o["counts"])
    for s in range(nshapes):
        assert abs(sp.rmax(s) - case["rmax"][s]) < 1e-14
    sp.close()


@pytest.mark.parametrize("lmax,nq,jpoly", [(L, nq, jp) for L, nq in ((0, 5), (1, 8), (2, 7), (3, 16), (4, 9), (5, 12), (6, 8),
                                                                     (6, 20), (7, 10), (8, 16), (9, 6), (10, 11),
                                                                     (11, 8), (12, 16), (2, 1), (3, 2), (4, 3), (6, 40), (5, 64), (12, 40)) for jp in (0, 1)])
def test_both_kernel_families_match_oracle(oracle, lmax, nq, jpoly):
    """Every compiled order through BOTH kernel families, forced with the "jpoly" option (left alone the library
    picks one per (lmax, nq)): the body-frame Horner evaluation of the neighbour's radius, and the per-azimuth
    polynomials in the pair's common frame (rotation kernel + node pairs; odd and even n_q, n_q that do and do not
    divide 64, ragged last slabs, one to three rings, 64 rings, ring groups), with the volume path and two shapes."""
    case = make_case(200, lmax, 2, seed=100 + lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    sp.set_option("jpoly", jpoly)
    sp.set_option("count", 1)
    b = case["bed"]
    f, tq, eng, vir = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    assert sp.kernel_info()["family"] == jpoly
    st = sp.stats()
    o = oracle_compute(oracle, case, nq, K, E, eflag=True, vflag=True)
    check(f, tq, o)
    assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
    assert np.abs(vir - o["eng_virial"][1:]).max() < TOL * np.abs(o["eng_virial"][1:]).max()
    assert (st["n_candidates"], st["n_contact"], st["n_touching"]) == tuple(o["counts"])
    sp.close()


@pytest.mark.parametrize("lmax,nq,rows,expo", [(7, 8, 0, 1.25), (7, 16, 4, 1.25), (8, 12, 0, 1.0), (9, 16, 0, 1.25), (9, 20, 8, 1.25),
                                                (10, 24, 0, 1.25), (11, 16, 8, 1.5), (12, 16, 0, 1.25), (12, 32, 0, 1.25),
                                                (12, 32, 4, 1.0), (12, 64, 0, 1.25), (8, 10, 0, 1.25)])
def test_two_waves_per_pair_match_oracle(oracle, lmax, nq, rows, expo):
    """The two-waves-per-pair form of the per-azimuth-polynomial kernels (option "split": the pair's tables shared by a
    128-lane workgroup, each wave half of the azimuths, a queue per wave, workgroup barriers at every table hand-over)
    forced for every order it is compiled for: n_q / 2 that does and does not divide 64, all rings resident and ring
    groups (rows: the "ring_rows" option), forces-only and volume laws, against the oracle and against the one-wave
    kernels of the same library."""
    case = make_case(160, lmax, 2, seed=300 + lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, expo)
    b = case["bed"]
    out = {}
    for split in (1, 0):
        sp = make_ctx(case, nq, K, E)
        sp.set_option("jpoly", 1)
        sp.set_option("split", split)
        if rows:
            sp.set_option("ring_rows", rows)
        sp.set_option("count", 1)
        f, tq, eng, vir = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
        ki = sp.kernel_info()
        assert ki["family"] == 1 and ki["waves_per_pair"] == (2 if split else 1) and ki["scratch_bytes"] == 0
        st = sp.stats()
        out[split] = (f, tq, eng, (st["n_candidates"], st["n_contact"], st["n_touching"]))
        sp.close()
    o = oracle_compute(oracle, case, nq, K, E, eflag=True, vflag=True)
    for split in (1, 0):
        f, tq, eng, counts = out[split]
        check(f, tq, o)
        assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
        assert counts == tuple(o["counts"])
    fs = np.abs(o["f"]).max()
    assert np.abs(out[1][0] - out[0][0]).max() < 1e-12 * fs


@pytest.mark.parametrize("lmax,nq,expo,nshapes", [(6, 16, 1.25, 1), (6, 16, 1.0, 4), (4, 10, 1.25, 2), (4, 10, 1.0, 1), (12, 32, 1.25, 1),
                                                   (12, 32, 1.0, 2)])
def test_specialised_instances_match_oracle_and_the_general_kernels(oracle, lmax, nq, expo, nshapes):
    """The BASELINE shapes run instances in which n_q, the resident ring rows and the queue capacity are compile-time
    constants (pair_kernel.hpp PairSpec; option "spec", default 1): the same arithmetic with fewer index instructions.
    Both force laws (NEEDV true / false instances), one and several shapes; against the oracle, against the general
    kernels of the same library (option "spec" 0), per pair — and the library must say which one ran."""
    import torch
    case = make_case(260, lmax, nshapes, seed=700 + lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, expo)
    b = case["bed"]
    out = {}
    npairs = case["jlist"].size
    for spec in (1, 0):
        sp = make_ctx(case, nq, K, E)
        sp.set_option("spec", spec)
        sp.set_option("count", 1)
        pairs = torch.zeros(max(npairs, 1), 7, dtype=torch.float64, device="cuda:0")
        sp.set_pair_output(pairs.data_ptr())
        f, tq, eng, vir = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
        ki = sp.kernel_info()
        assert ki["specialised"] == spec and ki["family"] == 1 and ki["scratch_bytes"] == 0, ki
        assert ki["waves_per_pair"] == (2 if lmax == 12 else 1)
        st = sp.stats()
        out[spec] = (f, tq, eng, (st["n_candidates"], st["n_contact"], st["n_touching"]), pairs.cpu().numpy())
        sp.close()
    o = oracle_compute(oracle, case, nq, K, E, eflag=True, vflag=True, want_pairs=True)
    for spec in (1, 0):
        f, tq, eng, counts, pr = out[spec]
        check(f, tq, o)
        assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0]) and counts == tuple(o["counts"])
    # same instructions on the data path: per pair the two agree to the last bits of the sums' order, and with the oracle
    scale = np.abs(o["pairs"]).max(axis=0) + 1e-300
    assert (np.abs(out[1][4] - out[0][4]) / scale).max() < 1e-13
    assert (np.abs(out[1][4] - o["pairs"]) / scale).max() < TOL
    # a launch that is NOT the order's BASELINE shape keeps the general kernel, whatever the option says
    sp = make_ctx(case, nq + 2, K, E)
    sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    assert sp.kernel_info()["specialised"] == 0
    sp.close()


def test_two_waves_per_pair_random_configurations_agree_with_one_wave(oracle):
    """A wider net for the two-wave kernels' barriers, queues and ring groups: pseudo-random (order, even n_q, resident
    rows, exponent) — including n_q / 2 that do not divide 64, single-slab caps and many short ring groups — each compared
    with the one-wave kernels of the same library on the same bed (1e-12) and, for every third case, with the oracle."""
    rng = np.random.default_rng(20261004)
    for case_no in range(18):
        lmax = int(rng.integers(7, 13))
        nq = int(2 * rng.integers(4, 21))                      # 8 .. 40, even
        per_slab = (64 + nq // 2 - 1) // (nq // 2)
        rows = int(rng.choice([0, 0, 2 * per_slab, 3 * per_slab, nq]))
        expo = float(rng.choice([1.0, 1.25, 1.5]))
        case = make_case(90, lmax, 2, seed=500 + case_no, rmax_fn=oracle.shape_rmax)
        K, E = coeff_tables(1, 700.0, expo)
        b = case["bed"]
        res = {}
        for split in (1, 0):
            sp = make_ctx(case, nq, K, E)
            sp.set_option("jpoly", 1)
            sp.set_option("split", split)
            if rows:
                sp.set_option("ring_rows", min(rows, nq))
            f, tq, eng, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
            assert sp.kernel_info()["waves_per_pair"] == (2 if split else 1), (lmax, nq)
            res[split] = (f, tq, eng)
            sp.close()
        fs = np.abs(res[0][0]).max()
        assert fs > 0, (lmax, nq, rows)
        assert np.abs(res[1][0] - res[0][0]).max() < 1e-12 * fs, (lmax, nq, rows, expo)
        assert np.abs(res[1][1] - res[0][1]).max() < 1e-12 * max(fs, np.abs(res[0][1]).max()), (lmax, nq, rows, expo)
        assert abs(res[1][2] - res[0][2]) < 1e-11 * abs(res[0][2]), (lmax, nq, rows, expo)
        if case_no % 3 == 0:
            check(res[1][0], res[1][1], oracle_compute(oracle, case, nq, K, E))


def test_split_option_is_ignored_where_it_does_not_apply(oracle):
    """Odd n_q, orders below 7 and the body-frame family have no two-wave form: the option must fall back silently."""
    for lmax, nq, jp in ((8, 9, 1), (6, 16, 1), (9, 16, 0)):
        case = make_case(60, lmax, 1, seed=11, rmax_fn=oracle.shape_rmax)
        K, E = coeff_tables(1, 1000.0, 1.25)
        sp = make_ctx(case, nq, K, E)
        sp.set_option("jpoly", jp)
        sp.set_option("split", 1)
        b = case["bed"]
        f, tq, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
        assert sp.kernel_info()["waves_per_pair"] == 1
        check(f, tq, oracle_compute(oracle, case, nq, K, E))
        sp.close()


def test_mixed_types_and_exponents(oracle):
    case = make_case(300, 6, 3, seed=40, ntypes=3, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(3, kn=lambda i, j: 300.0 * (i + j), expo=lambda i, j: 1.0 + 0.25 * abs(i - j))
    sp = make_ctx(case, 12, K, E)
    b = case["bed"]
    f, tq, eng, vir = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    o = oracle_compute(oracle, case, 12, K, E, eflag=True, vflag=True)
    check(f, tq, o)
    assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
    sp.close()


def test_per_pair_integrals_match_oracle(oracle):
    import torch
    case = make_case(200, 6, 2, seed=7, amp=0.25, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.0)
    sp = make_ctx(case, 16, K, E)
    sp.set_option("force_volume", 1)
    out = torch.zeros(case["jlist"].size, 7, dtype=torch.float64, device="cuda")
    sp.set_pair_output(out.data_ptr())
    b = case["bed"]
    sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    o = oracle_compute(oracle, case, 16, K, E, force_volume=True, want_pairs=True)
    got = out.cpu().numpy()
    assert np.abs(got - o["pairs"]).max() < TOL * np.abs(o["pairs"]).max()
    assert (got[:, 0] > 0).sum() == o["counts"][2]
    sp.set_pair_output(None)
    sp.close()


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_vectors(path):
    """Committed vectors only: needs neither the oracle nor /root/reference."""
    from shpair import ShPair
    g = np.load(path)
    lmax, nq = int(g["lmax"]), int(g["nq"])
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, g["anm"].shape[0])
    for s, a in enumerate(g["anm"]):
        sp.set_shape(s, lmax, a)
        assert abs(sp.rmax(s) - g["rmax"][s]) < 1e-14
    sp.coeff("*", "*", float(g["kn"]), float(g["exponent"]))
    sp.set_neighbors_csr(g["ilist"], g["offsets"], g["jlist"])
    sp.set_option("count", 1)
    sp.set_option("rule", int(g["rule"]) if "rule" in g else 0)
    f, tq, eng, vir = sp.compute(g["x"].shape[0], g["x"], g["quat"], g["type"], g["shtype"], eflag=True, vflag=True)
    fs = np.abs(g["f"]).max()
    assert np.abs(f - g["f"]).max() < TOL * fs
    assert np.abs(tq - g["torque"]).max() < TOL * max(fs, np.abs(g["torque"]).max())
    assert abs(eng - g["eng_virial"][0]) < TOL * g["eng_virial"][0]
    assert np.abs(vir - g["eng_virial"][1:]).max() < TOL * np.abs(g["eng_virial"][1:]).max()
    st = sp.stats()
    assert (st["n_candidates"], st["n_contact"], st["n_touching"]) == tuple(g["counts"])
    sp.close()


def test_lammps_layout_list_equals_csr_and_neighmask(oracle):
    case = make_case(150, 4, 1, seed=9, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1)
    sp = make_ctx(case, 8, K, E)
    b = case["bed"]
    f0, t0, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    # LAMMPS layout: reversed ilist order, per-atom arrays, special bits set on some j
    of, jl = case["offsets"], case["jlist"].copy()
    jl[::2] |= np.int32(1 << 30)
    first = [jl[of[i]:of[i + 1]] for i in range(case["n"])]
    numneigh = np.diff(of).astype(np.int32)
    sp.set_neighbors(case["ilist"][::-1].copy(), numneigh, first)
    f1, t1, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    fs = np.abs(f0).max()
    assert np.abs(f1 - f0).max() < 1e-12 * fs and np.abs(t1 - t0).max() < 1e-12 * fs
    sp.close()


def test_newton_off_with_ghosts(oracle):
    case = make_case(240, 6, 2, seed=11, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    n, nlocal = case["n"], 120
    sub = dict(case)
    sub["ilist"] = case["ilist"][:nlocal]
    sub["offsets"] = case["offsets"][:nlocal + 1]
    sub["jlist"] = case["jlist"][:case["offsets"][nlocal]]
    sp = make_ctx(sub, 10, K, E)
    b = case["bed"]
    for newton in (True, False):
        f = np.zeros((n, 3))
        tq = np.zeros((n, 3))
        _, _, eng, vir = sp.compute(nlocal, b["x"], b["quat"], b["type"], b["shtype"], newton_pair=newton,
                                    eflag=True, vflag=True, f=f, torque=tq)
        o = oracle_compute(oracle, sub, 10, K, E, nlocal=nlocal, newton_pair=newton, eflag=True, vflag=True)
        check(f, tq, o)
        assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
        assert np.abs(vir - o["eng_virial"][1:]).max() < TOL * np.abs(o["eng_virial"][1:]).max()
        if not newton:
            assert not f[nlocal:].any() and not tq[nlocal:].any()
    sp.close()


def test_compute_adds_into_f_and_torque(oracle):
    case = make_case(100, 4, 1, seed=12, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1)
    sp = make_ctx(case, 8, K, E)
    b = case["bed"]
    f0, t0, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    f = np.full((case["n"], 3), 5.0)
    tq = np.full((case["n"], 3), -2.0)
    sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], f=f, torque=tq)
    assert np.allclose(f - 5.0, f0, rtol=0, atol=1e-9 * np.abs(f0).max())
    assert np.allclose(tq + 2.0, t0, rtol=0, atol=1e-9 * np.abs(f0).max())
    sp.close()


def test_device_pointer_entry_point_and_stream(oracle):
    import torch
    case = make_case(300, 6, 1, seed=13, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.5)
    sp = make_ctx(case, 16, K, E)
    b = case["bed"]
    dev = torch.device("cuda:0")
    x = torch.from_numpy(b["x"]).to(dev)
    q = torch.from_numpy(b["quat"]).to(dev)
    ty = torch.from_numpy(b["type"]).to(dev)
    sh = torch.from_numpy(b["shtype"]).to(dev)
    f = torch.zeros(case["n"], 3, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(f)
    ev = torch.zeros(7, dtype=torch.float64, device=dev)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        sp.compute_device(case["n"], 0, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(),
                          tq.data_ptr(), eflag=True, vflag=True, ev=ev.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    o = oracle_compute(oracle, case, 16, K, E, eflag=True, vflag=True)
    check(f.cpu().numpy(), tq.cpu().numpy(), o)
    assert np.abs(ev.cpu().numpy() - o["eng_virial"]).max() < TOL * np.abs(o["eng_virial"]).max()
    # second call on the context's own stream accumulates on top
    sp.compute_device(case["n"], 0, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(),
                      tq.data_ptr(), stream=sp.own_stream())
    sp.synchronize()
    # and a third on the null stream (stream=None)
    sp.compute_device(case["n"], 0, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(),
                      tq.data_ptr())
    torch.cuda.synchronize()
    f.mul_(2.0 / 3.0)
    assert rel_err(f.cpu().numpy(), 2 * o["f"]) < TOL
    sp.close()


def test_device_resident_neighbor_list(oracle):
    """CSR list handed over as device pointers and expanded on the device == host upload."""
    import torch
    case = make_case(400, 6, 2, seed=16, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, 12, K, E)
    b = case["bed"]
    f0, t0, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    dev = torch.device("cuda:0")
    jl = case["jlist"].copy()
    jl[::4] |= np.int32(1 << 30)
    il_d = torch.from_numpy(case["ilist"]).to(dev)
    of_d = torch.from_numpy(case["offsets"]).to(dev)
    jl_d = torch.from_numpy(jl).to(dev)
    sp.set_neighbors_csr([0], [0, 0], [])          # forget the host-uploaded list
    sp.set_neighbors_device(case["n"], il_d.data_ptr(), of_d.data_ptr(), jl_d.data_ptr(), jl.size, case["n"] - 1)
    torch.cuda.synchronize()
    f1, t1, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    fs = np.abs(f0).max()
    assert fs > 0 and np.abs(f1 - f0).max() < 1e-12 * fs and np.abs(t1 - t0).max() < 1e-12 * fs
    sp.close()


def test_empty_ragged_and_separated_inputs(oracle):
    from shpair import ShPair, shapes
    case = make_case(60, 4, 1, seed=14, spacing=4.0, rmax_fn=oracle.shape_rmax)  # no pairs at all
    K, E = coeff_tables(1)
    sp = make_ctx(case, 8, K, E)
    b = case["bed"]
    f, tq, eng, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    assert not f.any() and not tq.any() and eng == 0.0
    # listed pairs whose bounding spheres do not overlap (stale list, skin region)
    il = np.arange(60, dtype=np.int32)
    of = np.zeros(61, dtype=np.int32)
    of[1:] = np.minimum(np.arange(1, 61), 3).cumsum()
    jl = np.concatenate([np.arange(i + 1, i + 1 + min(i + 1, 3)) % 60 for i in range(60)]).astype(np.int32)
    sp.set_neighbors_csr(il, of, jl)
    sp.set_option("count", 1)
    f, tq, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    assert not f.any() and sp.stats()["n_contact"] == 0 and sp.stats()["n_candidates"] == jl.size
    # zero atoms / zero-length list
    sp.set_neighbors_csr(np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    f, tq, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    assert not f.any()
    sp.close()


def test_error_codes(oracle):
    from shpair import ShPair, ShPairError, shapes
    sp = ShPair(0)
    with pytest.raises(ShPairError) as e:
        sp.settings(0)
    assert e.value.code == -1
    with pytest.raises(ShPairError) as e:
        sp.settings(1000)
    assert e.value.code == -6
    with pytest.raises(ShPairError) as e:
        sp.set_shape(0, 4, shapes.sphere(1.0, 4))
    assert e.value.code == -4  # set_ntypes first
    sp.set_ntypes(1, 2)
    with pytest.raises(ShPairError) as e:
        sp.set_shape(5, 4, shapes.sphere(1.0, 4))
    assert e.value.code == -1
    with pytest.raises(ShPairError) as e:
        sp.set_shape(0, 21, np.zeros(22 * 23))
    assert e.value.code == -6
    bad = shapes.sphere(1.0, 2)
    bad[3] = np.nan
    with pytest.raises(ShPairError) as e:
        sp.set_shape(0, 2, bad)
    assert e.value.code == -1
    sp.set_shape(0, 2, shapes.sphere(1.0, 2))
    with pytest.raises(ShPairError) as e:
        sp.coeff(1, 3, 1.0, 1.0)
    assert e.value.code == -1
    with pytest.raises(ShPairError) as e:
        sp.coeff(1, 1, 1.0, 0.5)
    assert e.value.code == -1
    x = np.zeros((2, 3))
    x[1, 0] = 1.5
    q = np.tile([1.0, 0, 0, 0], (2, 1))
    one = np.ones(2, np.int32)
    zero = np.zeros(2, np.int32)
    with pytest.raises(ShPairError) as e:  # no neighbour list yet
        sp.compute(2, x, q, one, zero)
    assert e.value.code == -4
    sp.set_neighbors_csr([0, 1], [0, 1, 1], [1])
    with pytest.raises(ShPairError) as e:  # shape 1 never set
        sp.compute(2, x, q, one, zero)
    assert e.value.code == -4 and "shape 1" in str(e.value)
    sp.set_shape(1, 0, shapes.sphere(1.0))
    with pytest.raises(ShPairError) as e:  # pair_coeff never set
        sp.compute(2, x, q, one, zero)
    assert e.value.code == -4 and "pair_coeff" in str(e.value)
    sp.coeff("*", "*", 100.0, 1.0)
    f, tq, _, _ = sp.compute(2, x, q, one, zero)
    assert f[0, 0] < 0 < f[1, 0] and abs(f[0, 0] + f[1, 0]) < 1e-12
    # out-of-range indices must be refused on the host, never reach the kernel
    with pytest.raises(ShPairError) as e:
        sp.compute(2, x, q, np.array([1, 2], np.int32), zero)
    assert e.value.code == -1 and "type" in str(e.value)
    with pytest.raises(ShPairError) as e:
        sp.compute(2, x, q, one, np.array([0, 2], np.int32))
    assert e.value.code == -1 and "shape index" in str(e.value)
    sp.set_neighbors_csr([0, 1], [0, 1, 1], [5])
    with pytest.raises(ShPairError) as e:
        sp.compute(2, x, q, one, zero)
    assert e.value.code == -1 and "stale list" in str(e.value)
    with pytest.raises(ShPairError) as e:
        sp.set_neighbors_csr([0, 1], [0, 1, 1], [-3 & 0x1FFFFFFF | 0])  # masked: a huge positive index
        sp.compute(2, x, q, one, zero)
    assert e.value.code == -1
    with pytest.raises(ShPairError) as e:
        ShPair(99)
    assert e.value.code == -2
    sp.close()


def test_reconfiguration_between_computes(oracle):
    """pair_coeff / shapes / nq changed after a compute are picked up (tables re-uploaded)."""
    case = make_case(150, 4, 1, seed=15, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.0)
    sp = make_ctx(case, 8, K, E)
    b = case["bed"]
    sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    sp.coeff(1, 1, 250.0, 1.5)
    sp.settings(12)
    K2, E2 = coeff_tables(1, 250.0, 1.5)
    f, tq, _, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"])
    check(f, tq, oracle_compute(oracle, case, 12, K2, E2))
    sp.close()


@pytest.mark.parametrize("cfg,lmax,nq,nshapes,nrows", [("config3", 6, 16, 4, 1200), ("config5", 12, 32, 1, 250)])
def test_full_size_other_baseline_configs(oracle, cfg, lmax, nq, nshapes, nrows):
    """BASELINE configs 3 (100k particles, 4 mixed L=6 shapes) and 5 (100k, L=12, nq=32) at full size:
    conservation laws on the whole bed, and a random subset of rows against the oracle."""
    case = make_case(100000, lmax, nshapes, seed=3 if nshapes > 1 else 5, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    n = case["n"]
    f, tq, eng, vir = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    fs = np.abs(f).max()
    assert fs > 0 and np.all(np.isfinite(f)) and np.all(np.isfinite(tq)) and eng > 0
    assert np.abs(f.sum(axis=0)).max() < 1e-10 * fs * np.sqrt(n)
    ang = (tq + np.cross(b["x"], f)).sum(axis=0)
    assert np.abs(ang).max() < 1e-9 * fs * np.sqrt(n) * np.abs(b["x"]).max()
    of, jl = case["offsets"], case["jlist"]
    rng = np.random.default_rng(1)
    rows = np.sort(rng.choice(n, nrows, replace=False))
    sof = np.zeros(rows.size + 1, np.int32)
    sof[1:] = (of[rows + 1] - of[rows]).cumsum()
    sjl = np.concatenate([jl[of[r]:of[r + 1]] for r in rows]).astype(np.int32)
    sub = dict(case)
    sub["ilist"], sub["offsets"], sub["jlist"] = rows.astype(np.int32), sof, sjl
    sp.set_neighbors_csr(sub["ilist"], sof, sjl)
    fg, tg, eg, _ = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(oracle, sub, nq, K, E, eflag=True, nthreads=0)
    check(fg, tg, o)
    assert abs(eg - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    sp.close()


def test_full_size_bed_conservation_and_linearity(oracle):
    """BASELINE config 2 at full size (100k particles, L=6, nq=16): size-independent properties.
    Newton's third law, angular-momentum balance, pair-list linearity, run-to-run agreement, and a
    spot check of 2000 random atoms' neighbourhoods against the oracle."""
    from shpair import ShPair
    case = make_case(100000, 6, 1, seed=2, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, 16, K, E)
    b = case["bed"]
    n = case["n"]
    f, tq, eng, vir = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    fs = np.abs(f).max()
    assert fs > 0 and np.all(np.isfinite(f)) and np.all(np.isfinite(tq))
    assert np.abs(f.sum(axis=0)).max() < 1e-10 * fs * np.sqrt(n)
    ang = (tq + np.cross(b["x"], f)).sum(axis=0)
    assert np.abs(ang).max() < 1e-9 * fs * np.sqrt(n) * np.abs(b["x"]).max()
    W = np.einsum("ia,ib->ab", b["x"], f)
    assert np.abs(vir - [W[0, 0], W[1, 1], W[2, 2], W[0, 1], W[0, 2], W[1, 2]]).max() < 1e-8 * np.abs(vir).max()
    # run-to-run: atomics reorder the sums, nothing else
    f2, tq2, eng2, _ = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    assert np.abs(f2 - f).max() < 1e-12 * fs and abs(eng2 - eng) < 1e-11 * eng
    # linearity in the pair list: two halves of the list sum to the whole
    of, jl, il = case["offsets"], case["jlist"], case["ilist"]
    h = n // 2
    facc = np.zeros_like(f)
    tacc = np.zeros_like(tq)
    for lo, hi in ((0, h), (h, n)):
        sp.set_neighbors_csr(il[lo:hi], of[lo:hi + 1] - of[lo], jl[of[lo]:of[hi]])
        sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], f=facc, torque=tacc)
    assert np.abs(facc - f).max() < 1e-12 * fs and np.abs(tacc - tq).max() < 1e-12 * fs
    # spot check against the oracle: all pairs of 1500 random rows
    rng = np.random.default_rng(0)
    rows = np.sort(rng.choice(n, 1500, replace=False))
    cnt = (of[rows + 1] - of[rows]).astype(np.int32)
    sof = np.zeros(rows.size + 1, np.int32)
    sof[1:] = cnt.cumsum()
    sjl = np.concatenate([jl[of[r]:of[r + 1]] for r in rows]).astype(np.int32)
    sub = dict(case)
    sub["ilist"], sub["offsets"], sub["jlist"] = rows.astype(np.int32), sof, sjl
    sp.set_neighbors_csr(sub["ilist"], sof, sjl)
    fg, tg, eg, _ = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(oracle, sub, 16, K, E, eflag=True, nthreads=0)
    check(fg, tg, o)
    assert abs(eg - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    sp.close()


def test_degenerate_orientations_of_the_cap_frame(oracle):
    """The cap-frame rotation reads Euler angles off M = [b1 b2 bc]; sin(beta) = 0 (neighbour exactly
    along +-z of i's body frame) and its neighbourhood are separate code paths.  Lattice-like inputs
    (identity quaternions, neighbours along the axes) hit them exactly."""
    from shpair import ShPair, shapes
    lmax, nq = 6, 12
    a = shapes.random_shape(lmax, 77, amp=0.3)
    rmax = oracle.shape_rmax(lmax, a)
    dirs = [(0, 0, 1), (0, 0, -1), (1, 0, 0), (0, -1, 0), (1e-9, 0, 1), (0, 1e-7, -1), (1e-5, 1e-5, 1),
            (3e-4, -2e-4, -1), (1e-3, 0, 1), (0.6, 0.0, 0.8), (-0.6, 0.0, -0.8), (2e-8, -1e-8, 1),
            (1e-12, 3e-12, -1), (1e-160, 0, 1)]
    quats = [(1, 0, 0, 0), (0, 0, 0, 1), (0, 1, 0, 0), (np.sqrt(0.5), 0, 0, np.sqrt(0.5))]
    x, q = [], []
    il, of, jl = [], [0], []
    for dvec in dirs:
        for qi in quats:
            d = np.array(dvec, float)
            d *= 1.8 / np.linalg.norm(d)
            base = np.array([10.0 * len(x), 0.0, 0.0])
            il.append(len(x))
            jl.append(len(x) + 1)
            of.append(len(jl))
            x += [base, base + d]
            q += [qi, (0.5, 0.5, -0.5, 0.5)]
            il.append(len(x) - 1)     # the second atom has an empty row
            of.append(len(jl))
    x = np.array(x)
    q = np.array(q, float)
    n = len(x)
    ty = np.ones(n, np.int32)
    sh = np.zeros(n, np.int32)
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, lmax, a)
    sp.coeff(1, 1, 1000.0, 1.5)
    sp.set_neighbors_csr(il, of, jl)
    f, tq, eng, _ = sp.compute(n, x, q, ty, sh, eflag=True)
    K, E = coeff_tables(1, 1000.0, 1.5)
    o = oracle.compute([(lmax, a, rmax)], K, E, nq, n, x, q, ty, sh, il, of, jl, eflag=True)
    assert o["counts"][2] >= len(dirs) * len(quats) - 4
    fs = np.abs(o["f"]).max()
    assert np.abs(f - o["f"]).max() < TOL * fs
    assert np.abs(tq - o["torque"]).max() < TOL * max(fs, np.abs(o["torque"]).max())
    # per pair, not only per bed: every orientation class must be right
    fo = o["f"][0::2]
    hit = np.abs(fo).max(axis=1) > 0
    per_pair = np.abs(f[0::2] - fo)[hit].max(axis=1) / np.abs(fo)[hit].max(axis=1)
    assert per_pair.max() < 1e-11 and not f[0::2][~hit].any()
    assert abs(eng - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    sp.close()


@pytest.mark.parametrize("lmax,nq", [(2, 1), (3, 2), (6, 64), (4, 128), (12, 64), (20, 24)])
def test_extreme_quadrature_orders(oracle, lmax, nq):
    """nq = 1 (two nodes per pair), ragged slabs, and table sizes that change the LDS budget / block shape."""
    case = make_case(60, lmax, 1, seed=50 + lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    f, tq, eng, vir = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    o = oracle_compute(oracle, case, nq, K, E, eflag=True, vflag=True)
    if np.abs(o["f"]).max() > 0:
        check(f, tq, o)
        assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
    else:
        assert not f.any()
    sp.close()


@pytest.mark.parametrize("lmax,nq,rows", [(6, 16, 3), (6, 16, 5), (4, 10, 5), (12, 32, 2), (12, 32, 9), (8, 7, 6),
                                          # the ring-table build maps lanes to (ring, order class) with 8 / 4 / 2 / 1 lanes
                                          # per ring by the group's row count, in passes of 64 lanes:
                                          (6, 16, 8), (5, 24, 9), (3, 40, 17), (3, 40, 33), (1, 96, 70), (2, 128, 128),
                                          (0, 100, 100)])
def test_ring_groups_do_not_change_the_result(oracle, lmax, nq, rows):
    """Large (lmax, nq) process the cap in groups of LDS-resident rings; forcing small groups on
    small problems must reproduce the oracle (and the single-group result) exactly as well.  The later cases force
    group sizes on either side of every lane-mapping threshold of cap_frame_rings (<= 8, <= 16, <= 32 rows, more than 64
    rows = two passes), including last groups that are smaller than the others."""
    case = make_case(120, lmax, 2, seed=60 + lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    f0, t0, e0, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    sp.set_option("ring_rows", rows)
    f1, t1, e1, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(oracle, case, nq, K, E, eflag=True)
    check(f1, t1, o)
    fs = np.abs(f0).max()
    assert np.abs(f1 - f0).max() < 1e-12 * fs and abs(e1 - e0) < 1e-12 * abs(e0)
    sp.close()


@pytest.mark.parametrize("lmax,nq,spacing", [(6, 16, 1.6), (6, 24, 1.7), (8, 20, 1.7), (12, 32, 1.75), (5, 16, 1.6), (2, 16, 1.6), (4, 20, 1.6),
                                                (9, 12, 1.6), (0, 16, 1.5)])
def test_dense_slabs_and_the_node_queue(oracle, lmax, nq, spacing):
    """Deeply overlapping pairs: the central slabs of a cap are full of inside nodes and do not always fit the node
    queue on top of a leftover (the slab is then classified a second time).  With the queue at its 128 entries and with
    the entries the LDS granule leaves (option queue_slack) the result is the oracle's; the layout stays within the same
    number of granules."""
    case = make_case(90, lmax, 2, seed=70 + lmax, spacing=spacing, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    o = oracle_compute(oracle, case, nq, K, E, eflag=True)
    res = {}
    for slack in (1, 0):
        sp.set_option("queue_slack", slack)
        f, t, e, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
        check(f, t, o)
        assert abs(e - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
        res[slack] = (f, e, sp.kernel_info())
    fs = np.abs(res[0][0]).max()
    assert np.abs(res[1][0] - res[0][0]).max() < 1e-12 * fs
    k1, k0 = res[1][2], res[0][2]
    assert k0["queue_entries"] == 128 <= k1["queue_entries"] <= 192
    assert k1["family"] == 1 and k1["lds_bytes_per_wave"] >= k0["lds_bytes_per_wave"] and k1["waves_per_cu_lds"] == k0["waves_per_cu_lds"]
    sp.close()


@pytest.mark.parametrize("lmax,nq,wpb,jpoly", [(6, 16, 2, -1), (6, 16, 4, -1), (4, 10, 4, -1), (5, 9, 3, 0), (9, 12, 2, -1), (14, 8, 2, -1)])
def test_several_waves_per_workgroup(oracle, lmax, nq, wpb, jpoly):
    """The tuning option waves_per_block: several one-wave pairs per workgroup, each with its own slice of the
    workgroup's LDS (the default is one; A/B in DESIGN 4.4) — same results."""
    case = make_case(100, lmax, 2, seed=80 + lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    f0, t0, e0, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    sp.set_option("waves_per_block", wpb)
    if jpoly >= 0:
        sp.set_option("jpoly", jpoly)
    f1, t1, e1, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(oracle, case, nq, K, E, eflag=True)
    check(f1, t1, o)
    assert abs(e1 - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
    assert np.abs(f1 - f0).max() < 1e-11 * np.abs(f0).max()
    sp.close()


@pytest.mark.parametrize("lmax,nq", [(6, 16), (3, 7), (12, 10)])
def test_loop_kernel_forced_for_a_compiled_order_and_the_timing_option(oracle, lmax, nq):
    """Option variant = 1: the run-time-order loop kernel also for the orders that have compiled kernels; option
    timing = 1: the pair kernels' time of the last call in the statistics."""
    case = make_case(100, lmax, 2, seed=90 + lmax, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    sp.set_option("variant", 1)
    sp.set_option("timing", 1)
    f1, t1, e1, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(oracle, case, nq, K, E, eflag=True)
    check(f1, t1, o)
    assert abs(e1 - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
    k = sp.kernel_info()
    assert k["compiled_order"] == 0 and k["lmax"] == lmax
    st = sp.stats()
    assert 0.0 < st["kernel_ms"] <= st["total_ms"] < 1e4
    sp.set_option("variant", 0)
    sp.set_option("timing", 0)
    f0, t0, e0, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    assert sp.kernel_info()["compiled_order"] == 1
    assert np.abs(f1 - f0).max() < 1e-11 * np.abs(f0).max()
    sp.close()


@pytest.mark.parametrize("lmax,nq,family,waves", [(11, 16, 1, 1), (12, 16, 1, 1), (9, 20, 1, 1), (11, 20, 1, 1), (12, 20, 1, 2), (11, 22, 1, 2),
                                                   (9, 26, 1, 2), (11, 24, 1, 1), (12, 24, 1, 2), (9, 10, 1, 1), (12, 10, 1, 1), (8, 4, 1, 1),
                                                   (10, 5, 0, 1), (12, 4, 0, 1), (10, 6, 1, 1), (8, 32, 1, 2), (7, 32, 1, 1)])
def test_the_library_s_own_choice_of_kernel_at_the_boundaries_of_its_rules(oracle, lmax, nq, family, waves):
    """With every option left alone: the kernel family, one or two waves per pair and the ring groups the measured rules
    pick (shpair_api.hip: use_jpoly_at, use_split, the ring-row rule) on either side of their boundaries — against the
    oracle, and the choice itself as the rules of round 4 make it."""
    case = make_case(80, lmax, 2, seed=110 + lmax + nq, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, nq, K, E)
    b = case["bed"]
    f, tq, eng, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(oracle, case, nq, K, E, eflag=True)
    check(f, tq, o)
    assert abs(eng - o["eng_virial"][0]) < TOL * abs(o["eng_virial"][0])
    k = sp.kernel_info()
    assert (k["family"], k["waves_per_pair"]) == (family, waves) and k["scratch_bytes"] == 0
    sp.close()


def test_non_finite_inputs_terminate(oracle):
    """inf coordinates and zero quaternions must not hang the kernel (every loop is bounded) and must not disturb pairs
    they are not part of; a NaN coordinate makes its pairs' separation not a number, which docs/SPEC.md 2 step 1 treats
    like coincident centres: those pairs are skipped and reported, the others are computed as before."""
    import torch
    from shpair import ShPairError
    case = make_case(200, 6, 1, seed=70, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, 16, K, E)
    b = case["bed"]
    n = case["n"]
    f0, t0, _, _ = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"])
    of, jl, il = case["offsets"], case["jlist"], case["ilist"]

    def clean_rows(bad):
        touched = set(bad)
        for ii, i in enumerate(il):
            for j in jl[of[ii]:of[ii + 1]]:
                if i in bad or int(j) in bad:
                    touched.update((int(i), int(j)))
        return np.array([a for a in range(n) if a not in touched])
    x = b["x"].copy()
    q = b["quat"].copy()
    x[50, 1] = np.inf
    q[100] = 0.0
    f, tq, _, _ = sp.compute(n, x, q, b["type"], b["shtype"])
    clean = clean_rows({50, 100})
    assert np.abs(f[clean] - f0[clean]).max() < 1e-12 * np.abs(f0).max()
    # NaN: reported; through the device-pointer entry point the forces of the clean rows can be looked at as well
    x[3] = np.nan
    dev = torch.device("cuda:0")
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in (("x", x), ("q", q), ("ty", b["type"]), ("sh", b["shtype"]))}
    fd = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    td = torch.zeros_like(fd)
    sp.compute_device(n, 0, t["x"].data_ptr(), t["q"].data_ptr(), t["ty"].data_ptr(), t["sh"].data_ptr(), fd.data_ptr(), td.data_ptr())
    torch.cuda.synchronize()
    with pytest.raises(ShPairError) as e:
        sp.synchronize()
    assert "coincident centres" in str(e.value)
    clean = clean_rows({3, 50, 100})
    assert np.abs(fd.cpu().numpy()[clean] - f0[clean]).max() < 1e-12 * np.abs(f0).max()
    sp.close()


@pytest.mark.parametrize("lmax,nq", [(3, 8), (6, 16), (9, 12)])
def test_random_pair_soup_all_cap_branches(oracle, lmax, nq):
    """Isolated random pairs from deep overlap (centre of i inside j, full-sphere cap, tangent-cone
    cap) to grazing: per-pair V, S_n, T_n and forces against the oracle."""
    import torch
    from shpair import ShPair, shapes
    rng = np.random.default_rng(1000 + lmax)
    shp = [shapes.random_shape(lmax, 300 + s, amp=0.35) for s in range(3)]
    rmax = [oracle.shape_rmax(lmax, a) for a in shp]
    npair = 240
    x = np.zeros((2 * npair, 3))
    q = rng.normal(size=(2 * npair, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    sh = rng.integers(0, 3, size=2 * npair).astype(np.int32)
    sep = np.concatenate([rng.uniform(0.15, 0.9, 60), rng.uniform(0.9, 1.6, 60), rng.uniform(1.6, 2.6, 120)])
    for p in range(npair):
        d = rng.normal(size=3)
        d *= sep[p] / np.linalg.norm(d)
        x[2 * p] = (20.0 * p, 0.0, 0.0)
        x[2 * p + 1] = x[2 * p] + d
    il = np.arange(2 * npair, dtype=np.int32)
    of = np.zeros(2 * npair + 1, np.int32)
    of[1:] = np.repeat(np.arange(1, npair + 1), 2)
    of[1::2] = np.arange(1, npair + 1)
    of[2::2] = np.arange(1, npair + 1)
    jl = (2 * np.arange(npair) + 1).astype(np.int32)
    ty = np.ones(2 * npair, np.int32)
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 3)
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
    sp.coeff(1, 1, 1000.0, 1.5)
    sp.set_neighbors_csr(il, of, jl)
    out = torch.zeros(npair, 7, dtype=torch.float64, device="cuda")
    sp.set_pair_output(out.data_ptr())
    f, tq, eng, _ = sp.compute(2 * npair, x, q, ty, sh, eflag=True)
    K, E = coeff_tables(1, 1000.0, 1.5)
    o = oracle.compute([(lmax, a, r) for a, r in zip(shp, rmax)], K, E, nq, 2 * npair, x, q, ty, sh, il, of, jl,
                       eflag=True, want_pairs=True)
    got = out.cpu().numpy()
    ref = o["pairs"]
    rho = sep
    rj = np.array(rmax)[sh[1::2]]
    assert (rho < rj).sum() > 40 and (got[:, 0] > 0).sum() > 150          # the deep branches are exercised
    scale = np.maximum(np.abs(ref).max(axis=1), 1e-30)
    err = np.abs(got - ref).max(axis=1) / scale
    assert err[ref[:, 0] > 0].max() < 1e-9, (np.argmax(err), err.max())
    assert not got[ref[:, 0] == 0, 0].any()
    fs = np.abs(o["f"]).max()
    assert np.abs(f - o["f"]).max() < TOL * fs
    assert abs(eng - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    sp.set_pair_output(None)
    sp.close()


@pytest.mark.parametrize("lmax", [8, 12])
def test_flat_spectrum_shapes(oracle, lmax):
    """Shapes whose high orders are as strong as the low ones: the worst case for the monomial
    (Horner) form of particle j and for the rounding of the cap-frame rotation of particle i."""
    from shpair import shapes
    rng = np.random.default_rng(90 + lmax)
    shp = []
    for s in range(2):
        a = np.zeros((shapes.nterms(lmax), 2))
        a[0, 0] = np.sqrt(4 * np.pi)
        amp = 0.25 / np.sqrt(shapes.nterms(lmax))
        for n in range(1, lmax + 1):
            for m in range(n + 1):
                a[n * (n + 1) // 2 + m, 0] = rng.normal(0, amp) * np.sqrt(4 * np.pi)
                if m:
                    a[n * (n + 1) // 2 + m, 1] = rng.normal(0, amp) * np.sqrt(4 * np.pi)
        shp.append(a.ravel())
    case = make_case(150, lmax, 2, seed=91, rmax_fn=oracle.shape_rmax)
    case["shapes"] = shp
    case["rmax"] = [oracle.shape_rmax(lmax, a) for a in shp]
    from shpair import bed
    case["ilist"], case["offsets"], case["jlist"] = bed.half_neighbor_list(case["bed"]["x"], case["bed"]["shtype"],
                                                                            case["rmax"])
    K, E = coeff_tables(1, 1000.0, 1.25)
    sp = make_ctx(case, 16, K, E)
    b = case["bed"]
    f, tq, eng, _ = sp.compute(case["n"], b["x"], b["quat"], b["type"], b["shtype"], eflag=True)
    o = oracle_compute(oracle, case, 16, K, E, eflag=True)
    assert o["counts"][2] > 100
    check(f, tq, o)
    assert rel_err(f, o["f"]) < 1e-11      # far inside TOL even in this worst case
    assert abs(eng - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
    sp.close()


@pytest.mark.parametrize("newton", [True, False])
def test_peratom_energy_and_virial(oracle, newton):
    """Pair::eatom / vatom (ev_tally_xyz halves), device and host forms, against the oracle."""
    import torch
    case = make_case(300, 6, 2, seed=45, rmax_fn=oracle.shape_rmax)
    nlocal = 300 if newton else 170
    if not newton:
        case = dict(case)
        case["ilist"] = case["ilist"][:nlocal]
        case["jlist"] = case["jlist"][:case["offsets"][nlocal]]
        case["offsets"] = case["offsets"][:nlocal + 1]
    K, E = coeff_tables(1, 900.0, 1.25)
    sp = make_ctx(case, 12, K, E)
    b = case["bed"]
    n = case["n"]
    o = oracle_compute(oracle, case, 12, K, E, nlocal=nlocal, newton_pair=newton, eflag=True, vflag=True, want_peratom=True)
    es, vs = o["eatom"].max(), np.abs(o["vatom"]).max()
    # host form
    ea, va = np.zeros(n), np.zeros((n, 6))
    sp.set_peratom_host(ea, va)
    f, tq, eng, vir = sp.compute(nlocal, b["x"], b["quat"], b["type"], b["shtype"], newton_pair=newton, eflag=True, vflag=True)
    check(f, tq, o)
    assert np.abs(ea - o["eatom"]).max() < TOL * es and np.abs(va - o["vatom"]).max() < TOL * vs
    if newton:
        assert abs(ea.sum() - eng) < 1e-12 * eng and np.abs(va.sum(0) - vir).max() < 1e-11 * np.abs(vir).max()
    else:
        assert ea[nlocal:].max() == 0.0           # ghosts are not tallied with newton off
    sp.set_peratom_host(None, None)
    # device form, without eflag: the per-atom energy alone switches the volume path on
    dev = torch.device("cuda:0")
    x, q = torch.from_numpy(b["x"]).to(dev), torch.from_numpy(b["quat"]).to(dev)
    ty, sh = torch.from_numpy(b["type"]).to(dev), torch.from_numpy(b["shtype"]).to(dev)
    fd = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    td = torch.zeros_like(fd)
    ed = torch.zeros(n, dtype=torch.float64, device=dev)
    vd = torch.zeros(n, 6, dtype=torch.float64, device=dev)
    sp.set_peratom_output(ed.data_ptr(), vd.data_ptr())
    for _ in range(2):                            # ADD semantics: two calls, twice the tallies
        sp.compute_device(nlocal, n - nlocal, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), fd.data_ptr(),
                          td.data_ptr(), newton_pair=newton)
    torch.cuda.synchronize()
    assert np.abs(ed.cpu().numpy() - 2 * o["eatom"]).max() < TOL * es
    assert np.abs(vd.cpu().numpy() - 2 * o["vatom"]).max() < TOL * vs
    sp.set_peratom_output(None, None)
    sp.close()


def test_two_waves_per_pair_with_tallies_newton_off_and_deterministic_mode(oracle):
    """The two-wave kernels' epilogue (wave 0 adds the other half's sums and finishes the pair) through every output it
    feeds: global and per-atom energy / virial, newton off with ghosts, and the per-slot stores of the deterministic mode
    (bitwise equal across two computes), against the oracle."""
    case = make_case(240, 9, 2, seed=61, rmax_fn=oracle.shape_rmax)
    nlocal = 150
    case = dict(case)
    case["ilist"] = case["ilist"][:nlocal]
    case["jlist"] = case["jlist"][:case["offsets"][nlocal]]
    case["offsets"] = case["offsets"][:nlocal + 1]
    K, E = coeff_tables(1, 800.0, 1.25)
    b = case["bed"]
    n = case["n"]
    o = oracle_compute(oracle, case, 16, K, E, nlocal=nlocal, newton_pair=False, eflag=True, vflag=True, want_peratom=True)
    es, vs = o["eatom"].max(), np.abs(o["vatom"]).max()
    runs = []
    for det in (0, 1, 1):
        sp = make_ctx(case, 16, K, E)
        sp.set_option("jpoly", 1)
        sp.set_option("split", 1)
        sp.set_option("deterministic", det)
        ea, va = np.zeros(n), np.zeros((n, 6))
        sp.set_peratom_host(ea, va)
        f, tq, eng, vir = sp.compute(nlocal, b["x"], b["quat"], b["type"], b["shtype"], newton_pair=False, eflag=True, vflag=True)
        assert sp.kernel_info()["waves_per_pair"] == 2
        check(f, tq, o)
        assert abs(eng - o["eng_virial"][0]) < TOL * o["eng_virial"][0]
        assert np.abs(vir - o["eng_virial"][1:]).max() < TOL * np.abs(o["eng_virial"][1:]).max()
        assert np.abs(ea - o["eatom"]).max() < TOL * es and np.abs(va - o["vatom"]).max() < TOL * vs
        assert ea[nlocal:].max() == 0.0
        runs.append((f.copy(), tq.copy()))
        sp.close()
    assert np.array_equal(runs[1][0], runs[2][0]) and np.array_equal(runs[1][1], runs[2][1])


def test_kernel_info_reports_the_launched_footprint(oracle):
    case = make_case(60, 6, 1, seed=46, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 900.0, 1.25)
    sp = make_ctx(case, 16, K, E)
    from shpair.capi import ShPairError
    with pytest.raises(ShPairError):
        sp.kernel_info()                         # nothing launched yet
    b = case["bed"]
    sp.compute(60, b["x"], b["quat"], b["type"], b["shtype"])
    k = sp.kernel_info()
    # L = 6, n_q = 16 runs the per-azimuth-polynomial kernel by the library's own rule: 96 registers (5 waves per
    # SIMD), 8.1 KB of LDS per wave + the queue entries that fill its 7th allocation granule of 1 280 B (18 waves per CU; one granule fewer would be 20)
    assert k["lmax"] == 6 and k["compiled_order"] == 1 and 80 < k["vgprs"] <= 96 and k["scratch_bytes"] == 0
    assert k["family"] == 1 and 7680 < k["lds_bytes_per_wave"] <= 8960 and k["ring_rows"] == 16
    assert k["waves_per_simd_vgpr"] == 5 and k["waves_per_cu"] == 18 and k["waves_per_cu_lds"] == 18
    assert 128 < k["queue_entries"] <= 192 and k["queue_entries"] % 4 == 0     # the rest of the 7th granule: 18 bytes per entry
    sp.set_option("jpoly", 0)       # the body-frame kernel of the same order
    sp.compute(60, b["x"], b["quat"], b["type"], b["shtype"])
    k = sp.kernel_info()
    assert k["lmax"] == 6 and k["compiled_order"] == 1 and 64 <= k["vgprs"] <= 80 and k["scratch_bytes"] == 0
    assert k["family"] == 0 and 4096 < k["lds_bytes_per_wave"] <= 8192 and k["ring_rows"] == 16
    assert k["waves_per_simd_vgpr"] == 6 and k["waves_per_cu"] == 21        # LDS (7.2 KB per wave = 6 granules of 1 280 B) is the limit
    assert k["queue_entries"] == 128
    sp.set_option("rule", 1)
    sp.compute(60, b["x"], b["quat"], b["type"], b["shtype"])
    kw = sp.kernel_info()
    assert kw["lds_bytes_per_wave"] > k["lds_bytes_per_wave"] and kw["waves_per_cu"] in (16, 17, 18, 19, 20)
    sp.close()


def test_runtime_order_kernel_with_peratom_tallies_and_info(oracle):
    """L = 14: the loop kernel (no compiled order) — per-atom tallies, kernel info, and the weighted rule refused."""
    from shpair.capi import ShPairError
    case = make_case(80, 14, 1, seed=47, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 900.0, 1.25)
    sp = make_ctx(case, 10, K, E)
    b = case["bed"]
    n = case["n"]
    ea, va = np.zeros(n), np.zeros((n, 6))
    sp.set_peratom_host(ea, va)
    f, tq, eng, vir = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=True, vflag=True)
    o = oracle_compute(oracle, case, 10, K, E, eflag=True, vflag=True, want_peratom=True, nthreads=8)
    check(f, tq, o)
    assert np.abs(ea - o["eatom"]).max() < TOL * o["eatom"].max() and abs(ea.sum() - eng) < 1e-12 * eng
    assert np.abs(va - o["vatom"]).max() < TOL * np.abs(o["vatom"]).max()
    k = sp.kernel_info()
    assert k["lmax"] == 14 and k["compiled_order"] == 0 and k["waves_per_cu"] >= 4
    sp.set_peratom_host(None, None)
    sp.set_option("rule", 1)
    with pytest.raises(ShPairError):
        sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"])
    sp.close()
