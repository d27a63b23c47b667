"""bench.py's self-launcher (`python bench.py --gpus N`, N > 1, no torch.distributed.run around it), exercised where
there is no GPU: the parent starts N fresh rank processes, every rank finds no device and leaves with exit code 3, the
parent relays that — no hang, one JSON line with `value` null and the reason, nothing left running.  The same door on a GPU box: tests/test_bench_contract.py."""
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
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]     # no measurement is made up: one line, value null, the reason
    assert len(lines) == 1 and '"value": null' in lines[0] and "exit code(s) [3, 3]" in lines[0]
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


def _json_lines(out):
    import json
    return [json.loads(ln) for ln in out.splitlines() if ln.startswith("{")]


def test_a_stalled_rank_ends_the_run_with_an_error_line_naming_phase_and_rank():
    """Bounded waits, seen from the driver's side: `bench.py --gpus 2` whose ranks never come back (diagnostic hook
    SHPAIR_BENCH_FAULT=stall, placed before the first GPU call so that it runs here).  The ranks' watchdogs give up
    after --wait-s, the PARENT prints ONE JSON line with `value` null and an `error` that names the phase and the rank,
    and exits 4 — a driver that ends the step after 600 s would have got nothing."""
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--wait-s", "3"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(ENV, SHPAIR_BENCH_FAULT="stall"))
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["value"] is None and lines[0]["n_gpus"] == 2
    assert "diagnostic stall" in lines[0]["error"] and "rank 0" in lines[0]["error"] and "did not finish in time" in lines[0]["error"]
    assert any("rank 1" in n for n in lines[0]["error_notes"])
    assert time.monotonic() - t0 < 60


def test_the_whole_run_bound_speaks_before_the_driver_would():
    """--total-s: phases that each stay inside their own bound can still add up; the whole run is bounded from the start of
    the process, the ranks get what is left of it, and the line says which phase was running when it passed."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--wait-s", "200", "--total-s", "6"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=dict(ENV, SHPAIR_BENCH_FAULT="stall"))
    assert r.returncode == 4, (r.returncode, r.stderr[-2000:])
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["value"] is None and "whole-run bound" in lines[0]["error"]
    # the default bounds fit the driver's 600 s: the whole run, and the sum a rank's phases may reach
    sys.path.insert(0, ROOT)
    import importlib
    bench = importlib.import_module("bench")
    old = sys.argv
    try:
        sys.argv = ["bench.py", "--gpus", "8"]
        a = bench.parse()
    finally:
        sys.argv = old
    assert 0 < a.total_s <= 560 and a.wait_s * 6 <= a.total_s


def test_a_rank_ended_by_its_launcher_still_leaves_a_line():
    """torch.distributed.run ends the surviving ranks with SIGTERM when one fails.  Rank 0 — the owner of stdout's line —
    answers from a thread of its own (signal wake-up pipe), so it works while the main thread sits in a C call."""
    import signal
    env = dict(ENV, SHPAIR_BENCH_FAULT="stall", RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--wait-s", "120"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env)
    time.sleep(6.0)
    p.send_signal(signal.SIGTERM)
    out, err = p.communicate(timeout=60)
    assert p.returncode == 143, (p.returncode, err[-2000:])
    lines = _json_lines(out)
    assert len(lines) == 1 and "SIGTERM" in lines[0]["error"] and "diagnostic stall" in lines[0]["error"]
