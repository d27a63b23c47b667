"""Several ranks, atoms that move: shpair.mrun.MultiRankRun (migration, 26-direction ghosts with periodic shifts,
device-built lists with global ids, forward / reverse every step) against the single-rank device-resident loop on
the same periodic bed.  Rehearsal transport: gloo with all ranks on the one GPU of the box (the product
transport is RCCL; see tests/test_bench_contract.py for what of it can be exercised here)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("grid,periodic,port,rule", [("2x1x1", "111", 29561, 0), ("2x2x1", "110", 29562, 0), ("1x1x1", "111", 29563, 0),
                                                     ("2x1x1", "100", 29564, 1)])
def test_multi_rank_dynamic_run_matches_single_rank(grid, periodic, port, rule):
    world = eval(grid.replace("x", "*"))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "mrun_worker.py"), grid, periodic, "120",
                        str(rule)], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["dx"] < 1e-7 and d["dv"] < 1e-6 and d["dq"] < 1e-9, d        # same trajectory (round-off grows with the steps)
    assert min(d["builds"]) >= 3 and len(set(d["builds"])) == 1, d          # every rank rebuilt, and together
    if world > 1:
        assert d["migrated"] > 0 and sum(d["owned_end"]) == d["n"], d       # atoms changed owner, none lost
        assert min(d["ghosts"]) > 0
