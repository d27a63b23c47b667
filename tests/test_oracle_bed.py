"""Bed-level invariants of the oracle (SPEC §2.7, §3)."""
import numpy as np
import pytest

from common import make_case, coeff_tables, oracle_compute, rel_err


@pytest.fixture(scope="module")
def case(oracle):
    return make_case(250, 4, 2, seed=1, amp=0.25, ntypes=2, rmax_fn=oracle.shape_rmax)


def test_momentum_and_angular_momentum_balance(oracle, case):
    K, E = coeff_tables(2, kn=lambda i, j: 500.0 * (i + j), expo=lambda i, j: 1.0 + 0.25 * (i != j))
    o = oracle_compute(oracle, case, 10, K, E)
    f, tq, x = o["f"], o["torque"], case["bed"]["x"]
    assert o["counts"][2] > 100
    fs = np.abs(f).max()
    assert np.abs(f.sum(axis=0)).max() < 1e-12 * fs * len(f)
    ang = (tq + np.cross(x, f)).sum(axis=0)
    assert np.abs(ang).max() < 1e-11 * fs * len(f)


def test_energy_and_virial_tally(oracle, case):
    K, E = coeff_tables(2, kn=800.0, expo=1.5)
    o = oracle_compute(oracle, case, 10, K, E, eflag=True, vflag=True, want_pairs=True)
    V = o["pairs"][:, 0]
    assert abs(o["eng_virial"][0] - (800.0 * V[V > 0] ** 1.5).sum()) < 1e-12 * o["eng_virial"][0]
    # virial = sum over pairs of (x_i - x_j) (x) F_i = -sum_atoms x (x) f  (newton on, no ghosts)
    x, f = case["bed"]["x"], o["f"]
    W = np.einsum("ia,ib->ab", x, f)
    v = o["eng_virial"][1:]
    ref = np.array([W[0, 0], W[1, 1], W[2, 2], W[0, 1], W[0, 2], W[1, 2]])
    assert np.abs(v - ref).max() < 1e-10 * np.abs(ref).max()


def test_threads_agree_with_serial(oracle, case):
    K, E = coeff_tables(2, kn=1000.0, expo=1.25)
    o1 = oracle_compute(oracle, case, 8, K, E, eflag=True, vflag=True, nthreads=1)
    o4 = oracle_compute(oracle, case, 8, K, E, eflag=True, vflag=True, nthreads=4)
    assert rel_err(o4["f"], o1["f"]) < 1e-13 and rel_err(o4["torque"], o1["torque"], np.abs(o1["f"]).max()) < 1e-13
    assert np.array_equal(o1["counts"], o4["counts"])
    assert rel_err(o4["eng_virial"], o1["eng_virial"]) < 1e-12


def test_newton_off_with_ghosts_matches_newton_on_for_locals(oracle, case):
    """Treat the upper half of the atoms as ghosts owned elsewhere."""
    K, E = coeff_tables(2, kn=1000.0, expo=1.0)
    n = case["n"]
    nlocal = n // 2
    il, of, jl = case["ilist"], case["offsets"], case["jlist"]
    # keep rows of local i only (half list stores i < j, so every local-ghost pair has a local i)
    sub = dict(case)
    sub["ilist"], sub["offsets"], sub["jlist"] = il[:nlocal], of[:nlocal + 1], jl[:of[nlocal]]
    on = oracle_compute(oracle, sub, 8, K, E, nlocal=nlocal, newton_pair=True, eflag=True)
    off = oracle_compute(oracle, sub, 8, K, E, nlocal=nlocal, newton_pair=False, eflag=True)
    assert np.abs(on["f"][:nlocal] - off["f"][:nlocal]).max() == 0.0
    assert not off["f"][nlocal:].any() and not off["torque"][nlocal:].any() and on["f"][nlocal:].any()
    assert off["eng_virial"][0] < on["eng_virial"][0]


def test_linear_exponent_forces_do_not_need_the_volume(oracle, case):
    K, E = coeff_tables(2, kn=1000.0, expo=1.0)
    a = oracle_compute(oracle, case, 8, K, E, force_volume=False)
    b = oracle_compute(oracle, case, 8, K, E, force_volume=True)
    assert rel_err(a["f"], b["f"]) < 1e-15 and a["counts"][2] == b["counts"][2]


def test_neighmask_bits_are_stripped(oracle, case):
    K, E = coeff_tables(2)
    a = oracle_compute(oracle, case, 6, K, E)
    tagged = dict(case)
    jl = case["jlist"].copy()
    jl[::3] |= np.int32(1 << 30)  # LAMMPS special-bond bits live above NEIGHMASK
    jl[1::3] |= np.int32(1 << 29)
    tagged["jlist"] = jl
    b = oracle_compute(oracle, tagged, 6, K, E)
    assert np.array_equal(a["f"], b["f"])


def test_empty_and_ragged_lists(oracle):
    c = make_case(40, 4, 1, seed=3, spacing=4.0, rmax_fn=oracle.shape_rmax)  # nothing within reach
    K, E = coeff_tables(1)
    assert c["jlist"].size == 0
    o = oracle_compute(oracle, c, 6, K, E)
    assert not o["f"].any() and tuple(o["counts"]) == (0, 0, 0)


def test_peratom_tallies_sum_to_the_global_ones(oracle):
    """ev_tally_xyz: with newton on, the per-atom halves add up to the global energy and virial."""
    case = make_case(120, 4, 2, seed=44, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 800.0, 1.5)
    o = oracle_compute(oracle, case, 8, K, E, eflag=True, vflag=True, want_peratom=True)
    assert o["counts"][2] > 50
    assert abs(o["eatom"].sum() - o["eng_virial"][0]) < 1e-12 * o["eng_virial"][0]
    assert np.abs(o["vatom"].sum(0) - o["eng_virial"][1:]).max() < 1e-12 * np.abs(o["eng_virial"][1:]).max()
    assert (o["eatom"] >= 0).all() and (o["eatom"] > 0).sum() > 60
    # forces unchanged by asking for the tallies
    o2 = oracle_compute(oracle, case, 8, K, E)
    assert np.array_equal(o["f"], o2["f"])
