"""The oracle reproduces the committed golden vectors (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "cfg*.npz")))
SETTLED = os.path.join(HERE, "golden", "settled_cfg1_L4.npz")


def test_fixtures_exist():
    assert len(GOLDEN) == 5 and os.path.exists(SETTLED)


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    lmax, nq = int(g["lmax"]), int(g["nq"])
    K = np.full((2, 2), float(g["kn"]))
    E = np.full((2, 2), float(g["exponent"]))
    oracle.set_rule(int(g["rule"]) if "rule" in g else 0)
    try:
        o = oracle.compute([(lmax, a, r) for a, r in zip(g["anm"], g["rmax"])], K, E, nq, g["x"].shape[0], g["x"],
                           g["quat"], g["type"], g["shtype"], g["ilist"], g["offsets"], g["jlist"], eflag=True,
                           vflag=True, want_pairs=True)
    finally:
        oracle.set_rule(0)
    fs = np.abs(g["f"]).max()
    assert np.abs(o["f"] - g["f"]).max() < 1e-12 * fs
    assert np.abs(o["torque"] - g["torque"]).max() < 1e-12 * fs
    assert np.abs(o["pairs"] - g["pairs"]).max() < 1e-12 * np.abs(g["pairs"]).max()
    assert np.abs(o["eng_virial"] - g["eng_virial"]).max() < 1e-11 * np.abs(g["eng_virial"]).max()
    assert np.array_equal(o["counts"], g["counts"])
    for s, (a, r) in enumerate(zip(g["anm"], g["rmax"])):
        assert abs(oracle.shape_rmax(lmax, a) - r) < 1e-14


def test_oracle_reproduces_settled_bed(oracle):
    """BASELINE configs[0]: the gravity-settled L = 4 bed (tests/golden/make_settled.py): ghosts, half list,
    forces, torques and energy from the oracle pieces equal the committed numbers, and the bed is a bed."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_settled", os.path.join(HERE, "golden", "make_settled.py"))
    ms = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ms)
    g = np.load(SETTLED)
    e = ms.expected(g)
    assert np.array_equal(e["ghost_owner"], g["ghost_owner"]) and np.array_equal(e["ghost_shift"], g["ghost_shift"])
    assert np.array_equal(e["offsets"], g["offsets"]) and np.array_equal(e["jlist"], g["jlist"])
    fs = np.abs(g["f"]).max()
    assert np.abs(e["f"] - g["f"]).max() < 1e-12 * fs
    assert np.abs(e["torque"] - g["torque"]).max() < 1e-12 * fs
    assert abs(e["energy"] - g["energy"]) < 1e-12 * g["energy"]
    # physical sanity of the inputs: 1000 mobile particles resting on the floor, weight carried by contacts
    nm = int(g["nmobile"])
    assert nm == 1000 and g["x"].shape[0] == 1400
    assert g["x"][:nm, 2].min() > 0.3 and g["x"][:nm, 2].max() < 6.0
    w = float(g["mass"]) * abs(g["gravity"][2])
    fz = g["f"][:nm, 2].sum()
    assert abs(fz - nm * w) < 0.02 * nm * w          # the floor carries the bed
