"""The optional weighted cap rule (docs/SPEC.md §2.8) in the oracle: equal to an independent numpy statement
of the rule, more accurate than the sharp rule at equal n_q, and continuous in the particle positions."""
import importlib.util
import os

import numpy as np
import pytest

from shpair import shapes, bed

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture()
def weighted(oracle):
    oracle.set_rule("weighted")
    yield oracle
    oracle.set_rule("sharp")


def _proto():
    spec = importlib.util.spec_from_file_location("weighted_rule_proto", os.path.join(ROOT, "tools", "proto", "weighted_rule_proto.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _contacts(rng, n, lo=1.75, hi=1.95):
    for _ in range(n):
        qi = rng.normal(size=4); qi /= np.linalg.norm(qi)
        qj = rng.normal(size=4); qj /= np.linalg.norm(qj)
        d = rng.normal(size=3); d *= rng.uniform(lo, hi) / np.linalg.norm(d)
        yield qi, qj, d


def test_weighted_rule_equals_independent_numpy_statement(weighted):
    P = _proto()
    lmax = 5
    a = shapes.random_shape(lmax, 81, amp=0.25)
    R = weighted.shape_rmax(lmax, a)
    rng = np.random.default_rng(1)
    n = 0
    for qi, qj, d in _contacts(rng, 12):
        for nq in (1, 2, 5, 12):
            hit, o, diag = weighted.pair(lmax, a, R, lmax, a, R, np.zeros(3), qi, d, qj, nq, need_volume=False)
            ref = P.rule(lmax, a, R, qi, qj, d, nq, True)
            assert hit
            assert np.abs(o[1:4] - ref).max() < 1e-12 * max(1.0, np.abs(ref).max())
            n += diag[0] > 0
    assert n > 20


def test_weighted_rule_is_more_accurate_at_equal_nq(oracle):
    lmax = 6
    a = shapes.random_shape(lmax, bed.SEED0 + 2)
    R = oracle.shape_rmax(lmax, a)
    rng = np.random.default_rng(2)
    es, ew = [], []
    for qi, qj, d in _contacts(rng, 16):
        oracle.set_rule("weighted")
        _, ref, _ = oracle.pair(lmax, a, R, lmax, a, R, np.zeros(3), qi, d, qj, 96, need_volume=False)
        _, ow, _ = oracle.pair(lmax, a, R, lmax, a, R, np.zeros(3), qi, d, qj, 12, need_volume=False)
        oracle.set_rule("sharp")
        _, osh, _ = oracle.pair(lmax, a, R, lmax, a, R, np.zeros(3), qi, d, qj, 12, need_volume=False)
        if np.linalg.norm(ref[1:4]) < 2e-2:
            continue
        es.append(np.linalg.norm(osh[1:4] - ref[1:4]) / np.linalg.norm(ref[1:4]))
        ew.append(np.linalg.norm(ow[1:4] - ref[1:4]) / np.linalg.norm(ref[1:4]))
    assert len(es) >= 10
    assert np.median(ew) < 0.4 * np.median(es)
    assert np.median(ew) < 6e-3


def test_weighted_force_is_continuous_in_the_separation(oracle):
    """Approach along a line in steps of 1e-4: the sharp rule jumps by whole node weights, the weighted rule
    moves by O(step)."""
    lmax = 4
    a = shapes.random_shape(lmax, 83, amp=0.2)
    R = oracle.shape_rmax(lmax, a)
    rng = np.random.default_rng(3)
    for qi, qj, d0 in _contacts(rng, 50, 1.8, 1.8):       # a pair that does touch along the whole path
        _, o, _ = oracle.pair(lmax, a, R, lmax, a, R, np.zeros(3), qi, d0 * (1.86 / 1.8), qj, 10)
        if o[0] > 5e-3:
            break
    dirn = d0 / np.linalg.norm(d0)
    jumps = {}
    for rule in ("sharp", "weighted"):
        oracle.set_rule(rule)
        S = []
        for k in range(400):
            _, o, _ = oracle.pair(lmax, a, R, lmax, a, R, np.zeros(3), qi, dirn * (1.86 - 1e-4 * k), qj, 10, need_volume=True)
            S.append([o[0], np.linalg.norm(o[1:4])])
        S = np.array(S)
        assert S.min() > 0
        jumps[rule] = np.abs(np.diff(S, axis=0)).max(0) / S.max(0)
    oracle.set_rule("sharp")
    assert jumps["sharp"][1] > 5e-3                        # a node flips: a visible step in |S_n|
    assert jumps["weighted"][1] < 0.1 * jumps["sharp"][1]
    # V is continuous under either rule (a node that flips has zero depth): both move by O(step)
    assert jumps["weighted"][0] < 2e-3 and jumps["sharp"][0] < 2e-3


def test_weighted_rule_limits(weighted):
    """Separated pairs give exactly zero; two spheres give the cap area more accurately than the sharp rule."""
    a0 = shapes.sphere(1.0)
    hit, o, _ = weighted.pair(0, a0, 1.0, 0, a0, 1.0, np.zeros(3), [1, 0, 0, 0], [2.5, 0, 0], [1, 0, 0, 0], 8)
    assert hit == 0 and not o.any()
    d = 1.7
    h = 1.0 - d / 2.0
    exact = np.pi * (2 * 1.0 * h - h * h)          # |S_n| = pi a^2, a^2 = 2Rh - h^2
    errs = {}
    for rule in ("sharp", "weighted"):
        weighted.set_rule(rule)
        _, o, _ = weighted.pair(0, a0, 1.05, 0, a0, 1.05, np.zeros(3), [1, 0, 0, 0], [d, 0, 0], [1, 0, 0, 0], 10)
        errs[rule] = abs(o[1] - exact) / exact
    assert errs["weighted"] < 0.3 * errs["sharp"]
