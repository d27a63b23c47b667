"""bench.py's contract, exercised on the GPU box: the one-line JSON of an N = 1 run carries every field the
driver reads (incl. `roofline`, `cpu_baseline`, `timestep`), and the N = 2 path — domain decomposition,
forward/reverse halo exchange around the HIP pair kernel, integrator half-steps — reproduces the
single-domain forces.  The N = 2 run is a rehearsal: both ranks share the one GPU of the box and the
halo buffers travel over `gloo` (the product transport, RCCL over xGMI, needs one GPU per rank)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "20000", "--steps", "5", "--warmup", "1",
                        "--ramp", "2", "--cpu-seconds", "2", "--ts-steps", "10"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "timestep", "occupancy"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["unit"] == "contact-pairs/s" and d["value"] > 1e7 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["contact_pairs_all_ranks"]) < 1e-6 * d["config"]["contact_pairs_all_ranks"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    assert d["occupancy"]["waves_per_cu"] >= 16 and d["occupancy"]["scratch_bytes"] == 0
    ts = d["timestep"]
    assert ts["timesteps_per_s"] > 0 and ts["steps"] == 10 and ts["particles"] > 15000


def test_two_rank_rehearsal_reproduces_single_domain_forces():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo",
                        "--verify", "--particles", "8000", "--steps", "3", "--warmup", "1", "--ramp", "0", "--cpu-seconds", "0",
                        "--multi-ts-steps", "6"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["backend"] == "gloo-rehearsal"
    assert d["verify_rel_err"] is not None and d["verify_rel_err"] < 1e-12
    assert d["config"]["ghost_atoms_rank0"] > 0 and d["config"]["contact_pairs_all_ranks"] > d["config"]["contact_pairs_rank0"]
    m = d["timestep_multi_rank"]                   # the optional leg with migration and rebuilds (shpair.mrun)
    assert m["steps"] == 6 and m["timesteps_per_s"] > 0 and m["particles_all_ranks"] > 15000 and m["ghosts_all_ranks"] > 0


def test_halo_exchange_over_real_rccl_with_the_rank_as_its_own_peer():
    """The product transport on the one GPU there is: `nccl` backend (RCCL), world size 1, send/recv to self
    — the batched P2POp groups, receives straight into the ghost rows and the index_add fold-in of
    shpair/halo.py (tools/nccl_self_halo.py asserts the results)."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29578",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nccl_self_halo.py")], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "RCCL self-peer halo exchange OK" in r.stdout
