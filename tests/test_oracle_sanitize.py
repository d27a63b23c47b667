"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (GPU sanitizers are not available on this
pool; the product's host code is covered by tests/test_host_tables.py): oracle/sanitize_main.c drives every
entry point — both cap rules, per-atom tallies, OpenMP bed, integrator, ghosts, half list — on small inputs."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_entry_points_are_clean_under_asan_ubsan():
    d = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", d, "sanitize_main"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="2")
    r = subprocess.run([os.path.join(d, "sanitize_main")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "sanitize ok" in r.stdout and "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
