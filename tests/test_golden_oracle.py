"""The oracle reproduces the committed golden vectors (tests/golden/make_golden.py)."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))


def test_fixtures_exist():
    assert len(GOLDEN) == 4


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    lmax, nq = int(g["lmax"]), int(g["nq"])
    K = np.full((2, 2), float(g["kn"]))
    E = np.full((2, 2), float(g["exponent"]))
    o = oracle.compute([(lmax, a, r) for a, r in zip(g["anm"], g["rmax"])], K, E, nq, g["x"].shape[0], g["x"],
                       g["quat"], g["type"], g["shtype"], g["ilist"], g["offsets"], g["jlist"], eflag=True,
                       vflag=True, want_pairs=True)
    fs = np.abs(g["f"]).max()
    assert np.abs(o["f"] - g["f"]).max() < 1e-12 * fs
    assert np.abs(o["torque"] - g["torque"]).max() < 1e-12 * fs
    assert np.abs(o["pairs"] - g["pairs"]).max() < 1e-12 * np.abs(g["pairs"]).max()
    assert np.abs(o["eng_virial"] - g["eng_virial"]).max() < 1e-11 * np.abs(g["eng_virial"]).max()
    assert np.array_equal(o["counts"], g["counts"])
    for s, (a, r) in enumerate(zip(g["anm"], g["rmax"])):
        assert abs(oracle.shape_rmax(lmax, a) - r) < 1e-14
