"""bench.py's contract, exercised on the GPU box: the one-line JSON of an N = 1 run carries every field the
driver reads (incl. `roofline`, `cpu_baseline`, `timestep`), and the N > 1 line (BASELINE configs[3]: whole timesteps
of the C++ multi-rank loop with migration and ghost exchange) is produced and self-consistent when the ranks are
rehearsed on the one GPU of the box."""
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
    assert "traffic" in rf and rf["traffic_source"]
    vf = d["valu_f64"]
    assert 60.0 < vf["peak_measured"] < 80.0 and abs(vf["frac_of_measured"] - vf["achieved"] / vf["peak_measured"]) < 1e-12
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["contact_pairs_all_ranks"]) < 1e-6 * d["config"]["contact_pairs_all_ranks"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and "tuned" in cb["variant"] and cb["max_rel_dev_from_oracle"] < 1e-11 and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    assert d["occupancy"]["waves_per_cu"] >= 16 and d["occupancy"]["scratch_bytes"] == 0
    ts = d["timestep"]
    assert ts["timesteps_per_s"] > 0 and ts["steps"] == 10 and ts["particles"] > 15000


def test_multi_rank_line_rehearsed_as_rank_threads():
    """bench.py --gpus N, N > 1, with the ranks as threads of one process on the one GPU (--transport local): the same
    C++ loop, plan and pack / unpack kernels as the RCCL path; --verify compares the decomposed initial forces with a
    single-domain compute.  (The product transport itself: tests/test_gpu_mrank.py::test_rccl_self_communicator...)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--transport", "local", "--verify",
                        "--particles", "6000", "--steps", "12", "--warmup", "1", "--ramp", "3", "--peak-ms", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 4 and d["scaling"] == "weak" and d["config"]["proc_grid"] == [2, 2, 1]
    assert d["verify_rel_err"] is not None and d["verify_rel_err"] < 1e-12
    h = d["halo"]
    assert h["transport"] == "local" and h["ranks_reported_by_transport"] == 4 and h["peers_rank0"] == 3
    assert min(h["ghost_atoms"]) > 0 and sum(h["owned_atoms"]) == d["config"]["particles_all_ranks"]
    assert max(h["rebuilds_in_timed_steps"]) >= 1 and len(set(h["rebuilds_in_timed_steps"])) == 1
    assert d["value"] > 1e6 and d["roofline"]["kernel_ms"] > 0 and d["roofline"]["traffic"] is None
    assert d["config"]["contact_pairs_all_ranks"] > d["config"]["contact_pairs_rank0"]
