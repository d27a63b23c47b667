"""The headers are plain C: examples/c_api_min.c (C99, -pedantic) includes both, drives the pair path through
the C ABI without LAMMPS or Python.  Without a GPU it must refuse (exit 77: no CPU fallback); with one it must
produce a force pair that obeys Newton's third law and the torque balance (exit 0)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def _build():
    subprocess.check_call(["make", "-C", EX], stdout=subprocess.DEVNULL)
    return os.path.join(EX, "c_api_min")


def test_c_example_compiles_and_refuses_without_a_gpu(gpu_available):
    exe = _build()
    if gpu_available:
        pytest.skip("GPU present: covered by the -m gpu test")
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 77 and "no CPU fallback" in r.stdout


@pytest.mark.gpu
def test_c_example_runs_on_the_gpu():
    exe = _build()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gfx950" in r.stdout and "F_i" in r.stdout


def test_c_halo_planner_example_runs_without_a_gpu():
    """include/shhalo.h is plain C99 and its host planner needs no device: 7 distinct peers at 2 x 2 x 2, one message
    per peer and direction of travel."""
    _build()
    r = subprocess.run([os.path.join(EX, "c_halo_plan")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "distinct peers: 7" in r.stdout and "7 messages each way" in r.stdout
