"""The tuned CPU baseline of bench.py (bench/cpu_tuned.c: tabulated recurrence constants, SIMD over cap nodes, OpenMP
over rows) against the plain oracle on the same beds: it is only allowed to be FASTER, not different."""
import importlib.util
import os
import time

import numpy as np
import pytest

from common import make_case, coeff_tables, oracle_compute

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tuned():
    spec = importlib.util.spec_from_file_location("cpu_tuned", os.path.join(ROOT, "bench", "cpu_tuned.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("lmax,nshapes,nq,expo,newton", [(6, 1, 16, 1.25, True), (4, 2, 10, 1.0, True), (12, 1, 9, 1.5, False),
                                                         (0, 1, 8, 1.25, True)])
def test_tuned_cpu_baseline_equals_the_oracle(oracle, lmax, nshapes, nq, expo, newton):
    T = _tuned()
    case = make_case(160, lmax, nshapes, seed=40 + lmax, rmax_fn=oracle.shape_rmax)
    n = case["n"]
    nlocal = n if newton else 100
    if not newton:
        case = dict(case)
        case["ilist"] = case["ilist"][:nlocal]
        case["jlist"] = case["jlist"][:case["offsets"][nlocal]]
        case["offsets"] = case["offsets"][:nlocal + 1]
    K, E = coeff_tables(1, 900.0, expo)
    o = oracle_compute(oracle, case, nq, K, E, nlocal=nlocal, newton_pair=newton, eflag=True)
    b = case["bed"]
    t = T.compute([(lmax, a, r) for a, r in zip(case["shapes"], case["rmax"])], K, E, nq, nlocal, b["x"], b["quat"], b["type"],
                  b["shtype"], case["ilist"], case["offsets"], case["jlist"], newton_pair=newton, eflag=True, nthreads=4)
    assert list(t["counts"]) == list(o["counts"]) and o["counts"][2] > 50
    fs = np.abs(o["f"]).max()
    assert np.abs(t["f"] - o["f"]).max() < 1e-12 * fs
    assert np.abs(t["torque"] - o["torque"]).max() < 1e-12 * max(fs, np.abs(o["torque"]).max())
    assert abs(t["energy"] - o["eng_virial"][0]) < 1e-12 * o["eng_virial"][0]


def test_tuned_is_faster_than_the_plain_oracle(oracle):
    T = _tuned()
    case = make_case(400, 6, 1, seed=3, rmax_fn=oracle.shape_rmax)
    K, E = coeff_tables(1, 900.0, 1.25)
    b = case["bed"]
    sh = [(6, a, r) for a, r in zip(case["shapes"], case["rmax"])]
    t0 = time.perf_counter()
    oracle_compute(oracle, case, 16, K, E)
    t1 = time.perf_counter()
    T.compute(sh, K, E, 16, case["n"], b["x"], b["quat"], b["type"], b["shtype"], case["ilist"], case["offsets"], case["jlist"],
              nthreads=1)
    t2 = time.perf_counter()
    assert (t2 - t1) < 0.6 * (t1 - t0), (t1 - t0, t2 - t1)
