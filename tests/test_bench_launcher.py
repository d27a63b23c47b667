"""bench.py's self-launcher (`python bench.py --gpus N`, N > 1, no torch.distributed.run around it), exercised where
there is no GPU: the parent starts N fresh rank processes, every rank finds no device and leaves with exit code 3, the
parent relays that — no hang, no JSON line, nothing left running.  The same door on a GPU box: tests/test_bench_contract.py."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}


def _no_gpu():
    import torch
    return torch.cuda.device_count() == 0


@pytest.mark.skipif(not _no_gpu(), reason="the GPU-less behaviour of the launcher")
def test_gpus_2_without_a_launcher_starts_two_ranks_that_report_no_gpu():
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--wait-s", "20"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=ENV)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert r.stderr.count("no GPU visible") == 2, r.stderr[-2000:]          # both ranks said so
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # and no line was made up
    assert time.monotonic() - t0 < 120


@pytest.mark.skipif(not _no_gpu(), reason="the GPU-less behaviour of the launcher")
def test_default_run_without_gpu_exits_3_before_starting_children():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2"], capture_output=True, text=True, timeout=300,
                       cwd=ROOT, env=ENV)
    assert r.returncode == 3 and r.stderr.count("no GPU visible") == 1, r.stderr[-2000:]


def test_run_rank_children_relays_the_worst_exit_code_and_ends_the_survivors(tmp_path):
    """The launcher's process handling on its own, with a stand-in script: rank 1 dies at once with code 7, rank 0 would
    sleep for a minute; the parent must end rank 0 (its exact PID) within its grace period and return 7."""
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "r = int(os.environ['RANK']); assert os.environ['WORLD_SIZE'] == '2' and os.environ['LOCAL_RANK'] == str(r)\n"
                      "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
                      "print('{\"rank\": %d}' % r, flush=True)\n"
                      "if r == 1: sys.exit(7)\n"
                      "time.sleep(60)\n")
    t0 = time.monotonic()
    rc, out = bench.run_rank_children([], 2, 50.0, script=str(script))
    assert rc == 7, rc                       # rank 1's own code; the SIGTERM given to rank 0 is the launcher's doing, not a result
    assert '{"rank": 0}' in out and time.monotonic() - t0 < 40
