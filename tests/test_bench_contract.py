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
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--particles", "20000", "--steps", "20", "--warmup", "1",
                        "--ramp", "2", "--cpu-seconds", "2", "--ts-steps", "10"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "timestep", "occupancy"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["unit"] == "contact-pairs/s" and d["value"] > 1e7 and d["higher_is_better"] is True
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert "traffic" in rf and rf["traffic_source"] and "traffic_all_pair_kernels" in rf
    assert "speed ratio" in d["valu_f64"]["note"].lower() and "source" in d["utilisation"]
    vf = d["valu_f64"]
    assert 60.0 < vf["peak_measured"] < 80.0 and abs(vf["frac_of_measured"] - vf["achieved"] / vf["peak_measured"]) < 1e-12
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - d["config"]["contact_pairs_all_ranks"]) < 1e-6 * d["config"]["contact_pairs_all_ranks"]
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and "tuned" in cb["variant"] and cb["max_rel_dev_from_oracle"] < 1e-11 and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    assert d["occupancy"]["waves_per_cu"] >= 16 and d["occupancy"]["scratch_bytes"] == 0
    ts = d["timestep"]
    assert ts["timesteps_per_s"] > 0 and ts["steps"] == 10 and ts["particles"] > 15000
    # the like-for-like N = 1 point of the scaling curve: the N > 1 workload and code path on one rank
    sr = d["scale_ref"]
    assert "error" not in sr, sr
    assert sr["transport"] == "rccl" and sr["ranks_reported_by_transport"] == 1 and sr["particles"] > 120000
    assert sr["library"] == d["library"] == "libshpair.so" and "--multi" in sr["cmd"]
    assert d["occupancy"]["kernel_hash"] and d["occupancy"]["kernel_symbol"].startswith("_ZN3shp19pair_contact_kernel")
    assert d["roofline"]["stale"] in (None, True, False) and d["utilisation"]["stale"] == d["roofline"]["stale"]
    assert sr["value"] > 1e7 and sr["steps"] == 20 and "configs[3]" in sr["workload"] and sr["ghost_atoms"] > 0
    assert abs(sr["value"] * sr["ms_per_step"] * 1e-3 - sr["contact_pairs"]) < 1e-6 * sr["contact_pairs"]
    assert sr["verify_overlap_ok"] is True and sr["overlap_used"] in (0, 2) and set(sr["overlap_ab_ms"]) == {"0", "2", "2p"}
    # every other single-GPU workload of BASELINE.json, in the same line (3 warm-up + 5 timed steps each, after the headline)
    cf = d["configs"]
    assert cf["steps"] == 5 and cf["warmup"] == 3
    for key, lmax, nq, nshapes in (("configs[2]", 6, 16, 4), ("configs[4]", 12, 32, 1), ("configs[0]-shape", 4, 10, 1)):
        c = cf[key]
        assert "error" not in c, c
        for k in ("value", "ms_per_step", "kernel_ms", "valu_f64_frac", "roofline_frac", "stale", "utilisation", "contact_pairs", "kernel_hash"):
            assert k in c, (key, k)
        assert (c["lmax"], c["nq"], c["nshapes"]) == (lmax, nq, nshapes) and c["value"] > 1e6 and 0 < c["kernel_ms"] <= c["ms_per_step"]
        assert c["stale"] in (None, True, False) and c["utilisation"]["stale"] == c["stale"] and c["contact_pairs"] > 500000
    c0 = cf["configs[0]"]      # the settled 1000-particle bed itself, held to the committed fixture inside the run
    assert "error" not in c0, c0
    assert c0["fixture_ok"] is True and c0["counts_match_fixture"] is True and c0["rel_err_vs_fixture"] < 1e-9
    assert c0["particles"] == 1400 and c0["half_list_pairs"] == 10168 and c0["value"] > 1e6 and c0["us_per_step"] > 0
    assert cf["configs[4]"]["waves_per_pair"] == 2 and cf["configs[4]"]["ms_per_step"] > 3 * cf["configs[2]"]["ms_per_step"]
    # the boundary north_star names: host arrays in, host arrays out (what an unmodified LAMMPS pays); never `value`
    hp = d["host_path"]
    assert "error" not in hp, hp
    assert hp["compute_call_ms_pinned"] > hp["compute_kernel_ms"] > 0 and 0 < hp["compute_overhead_ms_pinned"] < 5.0
    assert hp["compute_overhead_ms_pageable"] > 0 and hp["set_neighbors_ms"] > 0 and hp["set_neighbors_csr_ms"] > 0
    assert hp["half_list_pairs"] == d["config"]["half_list_pairs_rank0"] and hp["bytes_up_per_call"] > hp["bytes_down_per_call"] > 0
    assert d["elapsed_s"] > 0


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_rccl_body_at_world_1_under_torch_distributed_run():
    """The code that produces the driver's 8-GPU line — torch.distributed.run -> gloo rendezvous -> ncclGetUniqueId
    broadcast -> ncclCommInitRank -> shhalo_run_device -> gathered JSON — run end to end as a FRESH child process with
    one rank (grid 1x1x1, self-periodic in x and y, RCCL self-communicator): everything but the bytes between two
    devices.  The decomposed forces are verified against a single-domain compute inside the run (verify_ok)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--multi", "--particles", "30000",
           "--steps", "12", "--warmup", "1", "--ramp", "3", "--peak-ms", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 12 and d["config"]["proc_grid"] == [1, 1, 1] and d["config"]["backend"] == "rccl"
    h = d["halo"]
    assert h["transport"] == "rccl" and h["ranks_reported_by_transport"] == 1 and h["rccl_version"] > 20000
    assert h["owned_atoms"] == [d["config"]["particles_all_ranks"]] and h["ghost_atoms"][0] > 0     # no atom lost
    assert d["verify_ok"] is True and d["verify_rel_err"] < 1e-12 and d["transport_selftest"].startswith("ok")
    assert d["value"] > 1e6 and h["rebuilds_in_timed_steps"][0] >= 1
    # the same without a launcher (bench.py picks its own rendezvous port)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--multi", "--particles", "8000", "--steps", "4",
                        "--warmup", "1", "--ramp", "1", "--peak-ms", "0"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")})
    assert r.returncode == 0, r.stderr[-3000:]
    assert _last_json(r.stdout)["halo"]["transport"] == "rccl"


def test_self_launcher_gives_the_line_of_torch_distributed_run():
    """`bench.py --gpus 1 --multi --launch`: the door `bench.py --gpus 8` goes through when nothing launched it — the
    parent makes no GPU call, starts the rank as a fresh child with RANK / WORLD_SIZE / MASTER_* set and relays its
    line.  Same workload, same fields, same physics as under torch.distributed.run (above); timing differs by noise."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--gpus", "1", "--multi", "--particles", "20000", "--steps", "8", "--warmup", "1", "--ramp", "2", "--peak-ms", "0"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + ["--launch"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    a = _last_json(r.stdout)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    b = _last_json(r.stdout)
    assert "self-launch" in a["launcher"] and "launcher" not in b
    ca, cb = a["config"], b["config"]
    assert set(b) <= set(a) and set(ca) == set(cb) and a["halo"]["owned_atoms"] == b["halo"]["owned_atoms"]
    assert all(ca[k] == cb[k] for k in ("workload", "particles_all_ranks", "proc_grid", "backend", "lmax", "nq"))
    # atomics order the sums differently from run to run and the bed is chaotic: the contact counts agree closely, not exactly
    assert abs(ca["contact_pairs_all_ranks"] - cb["contact_pairs_all_ranks"]) < 0.01 * cb["contact_pairs_all_ranks"]
    assert a["verify_ok"] is True and b["verify_ok"] is True and a["halo"]["transport"] == "rccl"
    # the path the timed steps take was checked inside both runs: halo_overlap 2 against 0 from one saved state
    for d in (a, b):
        assert d["verify_overlap_ok"] is True and d["verify_overlap_rel_err"] < 1e-9 and d["overlap_candidate"] == 2
        assert d["overlap_used"] in (0, 2) and d["halo"]["overlap_option"] == d["overlap_used"] and set(d["overlap_ab_ms"]) == {"0", "2", "2p"} and d["overlap_stream_priority_used"] in (0, 1)
        assert "first_build" in d["setup_s"]["rank0"] and "overlap_check" in d["setup_s"]["max_over_ranks"]
    assert a["scale_ref_cmd"] == b["scale_ref_cmd"] and a["library"] == b["library"] == "libshpair.so"
    assert 0.5 < a["value"] / b["value"] < 2.0


def test_self_launcher_with_more_ranks_than_gpus_ends_cleanly():
    """`bench.py --gpus 2` on a box with ONE GPU: rank 1 finds no device of its own and leaves with exit code 3 while rank 0
    sits in the rendezvous; the parent gives rank 0 ten seconds, ends it, and returns rank 1's code — no hang, and the one line
    that comes has `value` null and says why."""
    import time
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "4000", "--steps", "2", "--wait-s", "60"],
                       capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert "only 1 GPU(s) visible" in r.stderr and "ended rank process(es) [0]" in r.stderr, r.stderr[-2000:]
    d = _last_json(r.stdout)     # one line, and it is not a measurement: value null, the reason in `error`
    assert d["value"] is None and d["n_gpus"] == 2 and d["error"] and any("[1] ended with exit code(s) [3]" in n for n in d.get("error_notes", [d["error"]]))
    assert time.monotonic() - t0 < 200


def test_a_rank_that_never_arrives_ends_the_run_with_a_message():
    """Bounded waits: WORLD_SIZE = 2 but only rank 0 exists.  The gloo rendezvous must give up within --wait-s with a
    message and a non-zero exit code instead of hanging (RCCL and the launcher have no timeout of their own here)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles", "4000", "--steps", "2", "--wait-s", "8"],
                       capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "did not finish in time" in r.stderr or "imeout" in r.stderr, r.stderr[-2000:]


def test_multi_rank_line_rehearsed_as_rank_threads():
    """bench.py --gpus N, N > 1, with the ranks as threads of one process on the one GPU (--transport local): the same
    C++ loop, plan and pack / unpack kernels as the RCCL path; --verify compares the decomposed initial forces with a
    single-domain compute.  (The product transport itself: tests/test_gpu_mrank.py::test_rccl_self_communicator...)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--transport", "local", "--verify",
                        "--particles", "5000", "--steps", "12", "--warmup", "1", "--ramp", "3", "--peak-ms", "0"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 8 and d["scaling"] == "weak" and d["config"]["proc_grid"] == [2, 2, 2]
    assert d["verify_rel_err"] is not None and d["verify_rel_err"] < 1e-12
    # halo_overlap 2 (exchanges on a second stream beside the owned-only slots) against 0, 4 timesteps from one saved state
    assert d["verify_overlap_ok"] is True and d["verify_overlap_rel_err"] < 1e-9 and d["overlap_used"] in (0, 2)
    assert set(d["overlap_ab_ms"]) == {"0", "2", "2p"} and d["overlap_stream_priority_used"] in (0, 1) and min(d["overlap_ab_ms"].values()) > 0
    h = d["halo"]
    assert h["transport"] == "local" and h["ranks_reported_by_transport"] == 8 and h["peers_rank0"] == 7
    assert min(h["ghost_atoms"]) > 0 and sum(h["owned_atoms"]) == d["config"]["particles_all_ranks"]
    assert max(h["rebuilds_in_timed_steps"]) >= 1 and len(set(h["rebuilds_in_timed_steps"])) == 1
    assert d["value"] > 1e6 and d["roofline"]["kernel_ms"] > 0 and d["roofline"]["traffic"] is None
    assert d["config"]["contact_pairs_all_ranks"] > d["config"]["contact_pairs_rank0"]


def test_a_wrong_overlap_result_is_not_used_and_fails_the_run():
    """The fall-back of the in-run halo_overlap check, forced with the diagnostic hook (SHPAIR_BENCH_FAULT=overlap perturbs one
    force component of the candidate's result on rank 0 by 1e-5 of max |F|): the line still comes — timed with
    halo_overlap 0 — says so (`overlap_used` 0, `verify_overlap_ok` false) and the exit code is 1."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "local", "--particles", "5000",
                        "--steps", "8", "--warmup", "1", "--ramp", "3", "--peak-ms", "0", "--halo-overlap", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, SHPAIR_BENCH_FAULT="overlap"))
    assert r.returncode == 1, (r.returncode, r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["verify_overlap_ok"] is False and d["verify_overlap_rel_err"] > 1e-7 and d["overlap_used"] == 0 and d["overlap_candidate"] == 2
    assert d["halo"]["overlap_option"] == 0 and d["value"] > 1e6 and "differs from 0" in r.stderr
    # asked for explicitly and correct: used whatever the A/B says
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "local", "--particles", "5000",
                        "--steps", "8", "--warmup", "1", "--ramp", "3", "--peak-ms", "0", "--halo-overlap", "2"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["verify_overlap_ok"] is True and d["overlap_used"] == 2 and d["halo"]["overlap_option"] == 2
    assert set(d["verify_overlap_rel_err_by_mode"]) == {"2", "2p"}


def test_real_rank_processes_on_one_gpu_through_the_host_staged_transport():
    """The N > 1 door with N REAL processes — everything the driver's 8-GPU run goes through except the RCCL wire: the
    self-launcher (2 ranks) and torch.distributed.run (4 ranks), gloo rendezvous, one context per process, the plan and the
    pack / unpack kernels per process, the bytes between ranks staged through host memory and torch.distributed's CPU
    backend (shhalo_create_staged), the in-run force verification and the halo_overlap check, the gathered line.  All
    ranks share the box's one GPU (--one-device; rank threads in one process would share an address space, and bugs that
    only separate processes show stay hidden)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    common = ["--transport", "staged", "--one-device", "--particles", "6000", "--steps", "24", "--warmup", "1", "--ramp", "3", "--peak-ms", "0",
              "--ab-steps", "6"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["halo"]["transport"] == "staged" and d["halo"]["ranks_reported_by_transport"] == 2
    assert d["transport_fallback"] is None and "host-staged" in d["config"]["backend"] and "self-launch" in d["launcher"]
    assert d["verify_ok"] is True and d["verify_rel_err"] < 1e-12 and d["verify_overlap_ok"] is True
    assert sum(d["halo"]["owned_atoms"]) == d["config"]["particles_all_ranks"] and min(d["halo"]["ghost_atoms"]) > 0
    assert d["halo"]["forward_bytes_per_step_rank0"] > 0 and d["value"] > 1e6
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4"] + common
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(env, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)     # ONE line on the launcher's stdout: RCCL's and gloo's banners went to stderr
    assert d["n_gpus"] == 4 and d["config"]["proc_grid"] == [2, 2, 1] and d["halo"]["transport"] == "staged" and d["halo"]["peers_rank0"] == 3
    assert d["verify_ok"] is True and d["verify_overlap_ok"] is True and len(d["halo"]["owned_atoms"]) == 4
    assert max(d["halo"]["rebuilds_in_timed_steps"]) >= 1 and len(set(d["halo"]["rebuilds_in_timed_steps"])) == 1
    assert d["halo"]["atoms_migrated_in_timed_steps"] >= 0 and sum(d["halo"]["owned_atoms"]) == d["config"]["particles_all_ranks"]


@pytest.mark.parametrize("fault", ["rccl_init", "rccl_forces"])
def test_an_rccl_attempt_that_fails_falls_back_to_the_staged_transport_on_every_rank(fault):
    """ncclCommInitRank returning an error on any rank, or decomposed forces over RCCL that are wrong, must not cost the run:
    every rank — agreed on through the control plane — takes the host-staged transport instead and the line says so
    (diagnostic hooks stand in for the failures: SHPAIR_BENCH_FAULT=rccl_init / rccl_forces)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--multi", "--particles", "8000", "--steps", "6", "--warmup", "1",
                        "--ramp", "2", "--peak-ms", "0", "--ab-steps", "4"], capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(env, SHPAIR_BENCH_FAULT=fault))
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["halo"]["transport"] == "staged" and d["transport_fallback"] and "FALLBACK" in d["config"]["backend"]
    assert ("shhalo_create_rccl failed" in d["transport_fallback"]) == (fault == "rccl_init")
    assert d["verify_ok"] is True and d["value"] > 1e6 and "falls back to the host-staged transport" in r.stderr


@pytest.mark.parametrize("launcher", ["self", "torchrun"])
def test_a_rank_lost_in_mid_run_leaves_one_line_that_says_so(launcher):
    """Three real rank processes (host-staged transport, one GPU); rank 1 dies after the first build (diagnostic hook).  No
    hang: the survivors sit in a gloo collective until their launcher ends them — torch.distributed.run and the
    self-launcher both send SIGTERM — and rank 0, the owner of stdout's line, still prints ONE line with `value` null and
    the phase it was in (a thread on the signal wake-up pipe answers while the main thread is inside the C call)."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["SHPAIR_BENCH_FAULT"] = "die_rank1"
    common = ["--gpus", "3", "--transport", "staged", "--one-device", "--particles", "4000", "--steps", "4", "--warmup", "1", "--ramp", "1",
              "--peak-ms", "0", "--wait-s", "60"]
    if launcher == "self":
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + common
        env["MASTER_ADDR"] = "127.0.0.1"
    t0 = time.monotonic()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and time.monotonic() - t0 < 150, (r.returncode, r.stderr[-2000:])
    assert "diagnostic exit" in r.stderr
    d = _last_json(r.stdout)
    assert d["value"] is None and d["n_gpus"] == 3 and d["error"]
    # what rank 0 saw first: its launcher's SIGTERM, or the dead peer's closed connection inside a gloo collective
    assert any(k in d["error"] for k in ("SIGTERM", "rank(s) [1]", "did not finish", "rank 0 failed")), d["error"]
    if launcher == "self":
        assert any("ended with exit code(s)" in n and "9" in n for n in d.get("error_notes", [])), d    # (rank 0 may have given up in the same instant)


def test_a_real_rccl_refusal_takes_the_fallback():
    """No hook: two rank processes on ONE device is something RCCL itself refuses (ncclCommInitRank: invalid usage — two ranks
    of a communicator on one GPU).  Every rank gets the error, the ranks agree on it, close what they have and run the
    host-staged transport; the line carries the reason."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "rccl", "--one-device", "--particles", "6000",
                        "--steps", "6", "--warmup", "1", "--ramp", "2", "--peak-ms", "0", "--ab-steps", "4", "--wait-s", "60"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["halo"]["transport"] == "staged" and "ncclCommInitRank failed" in d["transport_fallback"]
    assert d["verify_ok"] is True and d["verify_overlap_ok"] is True and d["n_gpus"] == 2 and d["value"] > 1e6


def test_an_optional_leg_that_hangs_does_not_cost_the_run_its_number():
    """The K timed steps come FIRST, in the plain mode, and their line is secured; the halo_overlap check and A/B are optional
    legs after it.  Here the last rank never comes back from the check (diagnostic hook): every rank's watchdog gives up
    after the phase bound, rank 0 leaves with the secured line plus `experiment_error`, and the exit code is 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--transport", "staged", "--one-device", "--particles", "5000",
                        "--steps", "8", "--warmup", "1", "--ramp", "3", "--peak-ms", "0", "--wait-s", "8"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=dict(env, SHPAIR_BENCH_FAULT="overlap_stall"))
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    d = _last_json(r.stdout)
    assert d["value"] > 1e6 and d["steps"] == 8 and d["verify_ok"] is True and d["overlap_used"] == 0
    # (which rank speaks first: the stalled rank's peer gives up inside the gloo exchange, or a watchdog at the phase bound)
    assert "halo_overlap check" in d["experiment_error"] and d["verify_overlap_ok"] is None and set(d["timed_by_mode"]) == {"0"}
    assert "the measurement taken before it stands" in r.stderr
