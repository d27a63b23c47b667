#!/usr/bin/env python3
"""bench.py — contact-pairs/s of the `pair_style sh` hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   — or the plain
   command above: with N > 1 and no WORLD_SIZE in the environment this process makes no GPU call and starts the N rank
   processes itself, fresh children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's line and
   returns the worst exit code; `self_launch`)

N = 1 — BASELINE.json configs[1]: 100k particles, one L_max = 6 shape, dense packed bed resident in HBM, n_q = 16
(Q = 512 cap nodes per pair), general force law (exponent 1.25: the overlap-volume root finder runs for every
touching node).  A "step" is one pass of the hot path inside the two integrator half-steps that surround it in a
timestep: initial_integrate -> clear f/torque -> shpair_compute_device() -> final_integrate.

N > 1 — BASELINE.json configs[3]: 125k particles per GPU (1 M at N = 8) of the same shape in a box periodic in x
and y on a frozen floor, gravity, bricks of the processor grid, one rank and one shpair context per GPU.  A step is a
whole timestep of the C++ loop over all ranks (shhalo_run_device, include/shhalo.h): initial_integrate -> rebuild
test over all ranks [-> atom migration, ghost plan, list build] -> forward halo -> clear -> pair compute -> reverse
halo -> gravity -> final_integrate; every byte between GPUs travels as ncclSend/ncclRecv issued by libshpair.so
(RCCL over xGMI).  torch.distributed (gloo) only hands out the ncclUniqueId and reduces the timings.
`--transport local` rehearses the same N ranks as threads of one process on one GPU (tests).
`--gpus 1 --multi` runs that N > 1 body with ONE rank (grid 1x1x1, self-periodic, RCCL self-communicator), also
under `python -m torch.distributed.run --nproc-per-node 1`: the code path of the driver's 8-GPU run, rehearsed on the
one GPU of a box (tests/test_bench_contract.py launches it as a fresh child process).
The N = 1 default line carries the same workload on one rank as `scale_ref` (125k particles, whole timesteps through
shhalo_run_device): parallel efficiency of N GPUs = value(N) / (N x scale_ref.value) compares like with like — the
N = 1 headline `value` is BASELINE configs[1] (static 100k bed, pair compute inside the two integrator half-steps).
Every wait on another rank is bounded (gloo rendezvous, ncclCommInitRank, the whole run): a rank that does not come
back ends the process with a message and exit code 4, never a hang and never a re-exec.

Prints ONE JSON line on rank 0 (fields: DESIGN.md §7).
"""
import argparse
import json
import os
import re
import signal
import sys
import threading
import time

T_START = time.monotonic()    # the whole-run bound (--total-s) counts from here

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

BYTES_PER_PAIR = 236          # algorithmic HBM bytes per contact pair
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6   # spec: 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz; the box's own v_fma_f64 rate
                              # is measured at run time (shpair_fp64_peak) and printed beside it


def flops_per_pair(lmax, nq):
    """SURVEY.md §8(d) algorithmic count: (60 + 6L + 9T) FLOP per cap node, Q = 2 nq^2 nodes."""
    T = (lmax + 1) * (lmax + 2) // 2
    return (60 + 6 * lmax + 9 * T) * 2 * nq * nq


def pmc_entry(args, family=0):
    """(entry, source) of profiles/pmc_traffic.json for this exact workload and kernel family, or (None, reason): HBM-side
    bytes per launch and utilisation figures measured with rocprofv3 PMC passes (tools/pmc_run.sh, tools/pmc_table.py).
    A static table: counters cannot be read from inside the timed run."""
    path = os.path.join("profiles", "pmc_traffic.json")
    key = f"{args.particles}:{args.lmax}:{args.nq}:{args.nshapes}:{args.exponent:g}:{args.rule}"
    if family == 1:   # the table's plain keys are the body-frame kernels
        key += ":jpoly"
    try:
        tab = json.load(open(os.path.join(ROOT, path)))
    except (OSError, ValueError):
        return None, f"{path} unreadable"
    if key not in tab:
        return None, f"no PMC measurement of workload {key} in {path}"
    e = tab[key]
    return e, f"{path}[{key}] (static; separate rocprofv3 --pmc passes, {e.get('files', '')})"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=0, help="particles per GPU (default: 100000 at N = 1, 125000 at N > 1)")
    ap.add_argument("--lmax", type=int, default=6)
    ap.add_argument("--nq", type=int, default=16)
    ap.add_argument("--nshapes", type=int, default=1)
    ap.add_argument("--exponent", type=float, default=1.25)
    ap.add_argument("--jpoly", type=int, default=-1, choices=[-1, 0, 1],
                    help="kernel family of the compiled orders (shpair_set_option \"jpoly\"): -1 the library's rule")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--transport", default="rccl", choices=["rccl", "local", "staged"],
                    help="N > 1: rccl = one process per GPU, ncclSend/ncclRecv over xGMI (the product path; if ncclCommInitRank "
                         "fails on any rank, or the forces over it are wrong, every rank falls back to `staged` and the line says "
                         "so); staged = one process per rank, the bytes between ranks staged through host memory and "
                         "torch.distributed's CPU backend (shhalo_create_staged: what a host without RCCL uses); local = "
                         "rehearsal: the N ranks are threads of this one process on GPU 0 (no torch.distributed.run)")
    ap.add_argument("--verify", dest="verify", action="store_true", default=None,
                    help="N > 1: compare the decomposed initial forces with a single-domain compute on rank 0 (untimed; the "
                         "default with --transport rccl: the first thing to know about a run between GPUs is whether its "
                         "forces are right)")
    ap.add_argument("--no-verify", dest="verify", action="store_false")
    ap.add_argument("--multi", action="store_true", help="with --gpus 1: run the N > 1 body (configs[3] workload, shhalo_run_device, "
                    "RCCL self-communicator) instead of the configs[1] headline")
    ap.add_argument("--launch", action="store_true", help="with --gpus 1 --multi: go through the self-launcher (this process starts "
                    "the rank as a fresh child, as `--gpus N` without a launcher does for N > 1)")
    ap.add_argument("--scale-ref", type=int, default=1, help="N = 1 default line: add the `scale_ref` object (0 = skip)")
    ap.add_argument("--wait-s", type=float, default=75.0, help="bound of every wait on another rank (rendezvous, "
                    "ncclCommInitRank, first build, verification, barriers; warm-up and timed steps get 2 x this)")
    ap.add_argument("--total-s", type=float, default=540.0, help="bound of the WHOLE run, counted from the start of this process "
                    "(the driver ends a bench step after 600 s without a word: before that, the process that owns the JSON line "
                    "prints one with an `error` field naming the phase and the rank, and exits 4); 0 = none")
    ap.add_argument("--verify-overlap", dest="verify_overlap", action="store_true", default=None,
                    help="N > 1: run 4 timesteps from one saved state with halo_overlap 0 and with the run's value, compare owned "
                         "positions / forces / torques by tag (1e-9), time both, and fall back to 0 when they differ (default on)")
    ap.add_argument("--no-verify-overlap", dest="verify_overlap", action="store_false")
    ap.add_argument("--ab-steps", type=int, default=12, help="N > 1: timesteps of each leg of the in-run halo_overlap A/B")
    ap.add_argument("--configs", type=int, default=1, help="N = 1 default line: add the `configs` object (BASELINE configs[2], "
                    "configs[4] and configs[0]'s shape at scale: 3 warm-up + 5 timed steps each, after the headline; 0 = skip)")
    ap.add_argument("--host-path", type=int, default=1, help="N = 1 default line: add the `host_path` object (what an unmodified "
                    "LAMMPS pays: shpair_compute on pinned host arrays, shpair_set_neighbors from firstneigh rows; 0 = skip)")
    ap.add_argument("--ramp", type=int, default=8, help="extra untimed passes before the W warm-up steps: the first "
                    "~8 launches of a fresh process run up to 25 %% slower while the GPU clock ramps (rocprof per-launch "
                    "durations in profiles/); they are never part of the K timed steps")
    ap.add_argument("--rule", default="sharp", choices=["sharp", "weighted"],
                    help="cap rule: sharp inside test (docs/SPEC.md §2.5, the headline) or covered-fraction weights (§2.8)")
    ap.add_argument("--ts-steps", type=int, default=40, help="steps of the whole-timestep leg (N = 1 only; 0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the cpu_baseline leg "
                    "(16 = the host-core share of one GPU on the bench box)")
    ap.add_argument("--peak-ms", type=float, default=20.0, help="length of the v_fma_f64 peak measurement (0 = skip)")
    ap.add_argument("--vthermal", type=float, default=0.25, help="N > 1: initial velocity scale of the bed")
    ap.add_argument("--halo-overlap", type=int, default=-1, choices=[-1, 0, 1, 2],
                    help="N > 1: option \"halo_overlap\" (1: the forward exchange runs on a stream of its own beside the pair "
                         "kernels of the slots that touch owned atoms only; 2: the reverse exchange hidden too, beside the second half of "
                         "those slots); -1 = auto: 2 if the in-run check against 0 passes AND the in-run A/B says it is not slower, else 0")
    ap.add_argument("--halo-stream-priority", type=int, default=-1, choices=[-1, 0, 1], help="N > 1: option \"halo_stream_priority\" (the "
                    "exchange stream of halo_overlap: 0 an ordinary stream, 1 one at the highest stream priority); -1 = both are "
                    "checked and timed in the run, the faster correct one is used")
    ap.add_argument("--one-device", action="store_true", help="N > 1 with --transport rccl: every rank uses GPU 0 (only to probe "
                    "what RCCL does with several ranks on one device; RCCL normally refuses)")
    a = ap.parse_args()
    multi = a.gpus > 1 or a.multi
    if a.particles <= 0:
        a.particles = 125000 if multi else 100000
    if a.verify is None:
        a.verify = multi and a.transport in ("rccl", "staged")
    if a.verify_overlap is None:
        a.verify_overlap = multi
    return a


_OUT_FD = None


def own_stdout():
    """From here on file descriptor 1 of this process is stderr, and the JSON line goes out through a private copy of the
    real stdout: RCCL prints a version banner on stdout when its first communicator comes up, gloo a line of its own —
    libraries this script loads, writing into the stream whose ONE line the driver parses."""
    global _OUT_FD
    if _OUT_FD is None:
        sys.stdout.flush()
        _OUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(text):
    if _OUT_FD is None:
        print(text, flush=True)
    else:
        sys.stdout.flush()
        os.write(_OUT_FD, (text + "\n").encode())


def error_line(args, what):
    """The JSON line of a run that did not produce a measurement: same envelope, `value` null, `error` says which
    phase on which rank gave up.  Printed by whoever owns stdout's line (rank 0, or the self-launcher's parent)."""
    return json.dumps({"metric": "contact_pairs_per_sec", "value": None, "unit": "contact-pairs/s", "n_gpus": args.gpus,
                       "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True,
                       "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "error": what,
                       "elapsed_s": round(time.monotonic() - T_START, 1)})


def _kfd_gpu_nodes():
    """Number of GPU nodes the kernel driver exposes, read from sysfs — no HIP / HSA call, so a parent that only wants
    to know whether starting GPU children makes sense stays free of any GPU context (torch.cuda.device_count() falls
    back to hipGetDeviceCount when amdsmi is not usable).  0 when the KFD topology is absent (ROCr itself enumerates
    from it: no topology, no device); None when it is there but unreadable (unknown: let the child find out)."""
    top = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(top):
        return 0
    n = 0
    try:
        for node in os.listdir(top):
            with open(os.path.join(top, node, "properties")) as fh:
                for ln in fh:
                    if ln.startswith("simd_count"):
                        n += 1 if int(ln.split()[1]) > 0 else 0
                        break
    except (OSError, ValueError):
        return None
    return n


class Watchdog:
    """Bounded waits: a daemon thread that ends the process (message on stderr, exit code 4) when the phase that was
    declared with `phase(what, seconds)` has not been left in time, or when the whole run has lasted --total-s.  The
    calls it guards sit in C (ncclCommInitRank, gloo collectives, hipStreamSynchronize behind an ncclRecv whose sender
    died) and cannot be interrupted from Python; RCCL itself has no timeout.  os._exit, not an exec: the GPU is
    initialised.  `emit`: this process owns the JSON line on stdout (rank 0) and prints one with an `error` field
    before it leaves.  watch_sigterm(): a launcher that ends this rank because ANOTHER rank failed (torch.distributed.run
    sends SIGTERM) gets the same courtesy — the C-level signal handler writes to a wake-up pipe that a thread of its
    own reads, so it works while the main thread sits in a C call that never returns."""

    def __init__(self, args=None, emit=False):
        self.args, self.emit = args, emit
        self._lock = threading.Lock()
        self._deadline, self._what = None, "start-up"
        self._secured = None
        self._failed_in = None
        self._total = (T_START + args.total_s) if (args is not None and getattr(args, "total_s", 0) > 0) else None
        t = threading.Thread(target=self._loop, daemon=True)
        t.start()

    def secure(self, line):
        """A measurement is in hand (rank 0: the line as a dict; other ranks: True).  What follows are optional legs — if one of
        them does not come back, the process leaves with THAT line (plus `experiment_error`) and exit code 0 instead of an
        error line: an experiment must never cost the run its number."""
        with self._lock:
            self._secured = line

    def _leave(self, why, code):
        rank = os.environ.get("RANK", "0")
        with self._lock:
            w, sec = (self._failed_in or self._what), self._secured
        msg = f"rank {rank}: '{w}' {why}"
        if sec is not None:
            print(f"bench.py: {msg}; the measurement taken before it stands, leaving with it", file=sys.stderr, flush=True)
            if self.emit and isinstance(sec, dict):
                emit(json.dumps(dict(sec, experiment_error=msg)))
            else:
                time.sleep(3.0)    # rank THREADS of one process share its exit: the thread that owns the line goes first
            os._exit(0)
        print(f"bench.py: {msg}; giving up with exit code {code}", file=sys.stderr, flush=True)
        if self.emit and self.args is not None:
            emit(error_line(self.args, msg))
        os._exit(code)

    def _loop(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                d = self._deadline
            now = time.monotonic()
            if d is not None and now > d:
                self._leave("did not finish in time (another rank lost, or RCCL / gloo cannot reach its peers)", 4)
            if self._total is not None and now > self._total:
                self._leave(f"was still running when the whole-run bound of {self.args.total_s:.0f} s passed", 4)

    def watch_sigterm(self):
        """Main thread only."""
        r, w = os.pipe()
        os.set_blocking(w, False)
        signal.signal(signal.SIGTERM, lambda *_: None)    # a Python-level handler, so that the C handler feeds the pipe
        signal.set_wakeup_fd(w, warn_on_full_buffer=False)

        def _wait():
            while True:
                b = os.read(r, 1)
                if b and b[0] == signal.SIGTERM:
                    self._leave("was interrupted by SIGTERM (the launcher ends every rank when one of them fails: see that "
                                "rank's message on stderr)", 143)
        threading.Thread(target=_wait, daemon=True).start()

    def phase(self, what, seconds):
        wd = self

        class _P:
            def __enter__(self_inner):
                with wd._lock:
                    self_inner.prev = (wd._deadline, wd._what)
                    wd._deadline, wd._what = time.monotonic() + seconds, what

            def __exit__(self_inner, *exc):
                with wd._lock:
                    if exc and exc[0] is not None and wd._failed_in is None:
                        wd._failed_in = wd._what    # the innermost phase an exception came out of: what a later message names
                    wd._deadline, wd._what = self_inner.prev
                return False
        return _P()


def _library_name():
    """Base name of the shared library this process computes with (a diagnostic build can only be selected with
    SHPAIR_LIB + SHPAIR_DIAGNOSTIC=1, and then says so here)."""
    from shpair import capi
    return capi.library_name()


def make_ctx(args, shp, device):
    from shpair import ShPair
    sp = ShPair(device)
    sp.settings(args.nq)
    sp.set_ntypes(1, args.nshapes)
    for s, a in enumerate(shp):
        sp.set_shape(s, args.lmax, a)
    sp.coeff("*", "*", 1000.0, args.exponent)
    sp.set_option("rule", 1 if args.rule == "weighted" else 0)
    sp.set_option("jpoly", args.jpoly)
    return sp


def roofline_objects(args, sp, n_contact, kernel_ms, world):
    fpp = flops_per_pair(args.lmax, args.nq)
    fam = sp.kernel_info()["family"]
    kernels = ("pair_setup_kernel + pair_rotate_lane_kernel + pair_contact_kernel" if fam == 1
               else "pair_setup_kernel + pair_contact_kernel")
    achieved_gbs = BYTES_PER_PAIR * n_contact / (kernel_ms * 1e-3) / 1e9
    achieved_tf = fpp * n_contact / (kernel_ms * 1e-3) / 1e12
    ent, src = pmc_entry(args, fam) if (world == 1 and not getattr(args, "multi", False)) else (None, "N > 1 workload: not measured")
    traffic = ent["traffic_bytes"] if ent else None
    peak_meas = None
    if args.peak_ms > 0:
        peak_meas = sp.fp64_peak(0, args.peak_ms)[0]
    ki = sp.kernel_info()
    from shpair import capi, codeobj
    ksym, khash = codeobj.contact_kernel_hash(capi.library_path(), ki["lmax"] if ki["compiled_order"] else -1, ki["needv"], ki["weighted"],
                                              ki["family"], ki["waves_per_pair"], ki.get("specialised", 0))
    # is the static PMC table's entry a measurement of the code that just ran?  (hash of the kernel's machine code + the
    # launch shape; tools/pmc_table.py stores both with every entry)
    stale = None
    if ent and khash is not None and ent.get("kernel_hash") is not None:   # no hash on either side: nothing was compared -> unknown
        stale = not (ent.get("kernel_hash") == khash and ent.get("ring_rows") == ki["ring_rows"]
                     and ent.get("waves_per_pair") == ki["waves_per_pair"])
    roof = {
        "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": src, "stale": stale,
        "traffic_all_pair_kernels": (ent.get("traffic_all_pair_kernels_bytes") if ent else None),
        "traffic_note": "traffic = pair_contact_kernel alone (the dominant kernel); traffic_all_pair_kernels adds the set-up and "
                        "rotation kernels that kernel_ms also covers (pair records and rotated coefficient vectors: a deliberate "
                        "trade of HBM bytes for VALU instructions, DESIGN.md 4.5)",
        "kernel": "pair_contact_kernel", "kernels_timed": kernels, "kernel_ms": kernel_ms, "bytes_per_pair": BYTES_PER_PAIR,
        "pairs_per_launch": int(n_contact), "algorithmic_bytes_per_launch": BYTES_PER_PAIR * int(n_contact),
        "note": "north_star asks for the HBM fraction; the kernel is FP64-VALU bound (see valu_f64)",
    }
    valu = {
        "bound": "valu_f64", "achieved": achieved_tf, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
        "frac": achieved_tf / F64_VALU_PEAK_TFLOPS, "peak_measured": peak_meas,
        "frac_of_measured": (achieved_tf / peak_meas) if peak_meas else None, "flop_per_pair": fpp,
        "note": "naive-algorithm FLOP (SURVEY §8d formula: every cap node evaluated with the full expansion) / kernel time / "
                "peak: a SPEED RATIO against that algorithm at the FP64 peak, not a utilisation — the kernels execute a "
                "different, cheaper algorithm (DESIGN.md 4.3, 4.7), so it can exceed 1; executed work is in `utilisation`. "
                "peak = spec; peak_measured = independent v_fma_f64 "
                "chains on every SIMD of this box, run beside the bench (shpair_fp64_peak). No MFMA: measured on MI355X, "
                "v_mfma_f64 and v_fma_f64 share one FP64 datapath (side by side they add up to the single-pipe rate, "
                "profiles/r02_a_fp64_peak.json)",
    }
    occ = dict(ki, kernel_symbol=ksym, kernel_hash=khash, note="static footprint of pair_contact_kernel as launched: one wave = one pair = one "
               "workgroup; waves_per_cu = min(4 x VGPR limit, LDS limit) of a gfx950 CU; family 1 = neighbour radius from "
               "per-azimuth polynomials in the pair's common frame (DESIGN.md 4.7), 0 = body-frame Horner evaluation")
    util = {"source": src, "stale": stale,
            "stale_note": "false: the table entry was measured on this very contact-kernel code (SHA-256 of its machine code, "
                          "occupancy.kernel_hash) with the same ring groups / waves per pair; true: on another build — re-run "
                          "tools/pmc_run.sh; null: no entry, or no hash on one side (nothing was compared)"}
    if ent:
        util["measured_on"] = {k: ent.get(k) for k in ("kernel_hash", "commit", "ring_rows", "waves_per_pair")}
        for k in ("valu_busy", "lds_busy", "lds_bank_conflict_share", "fp64_instr_share", "int32_instr_share", "valu_instr_per_pair",
                  "fp64_flop_per_pair_executed", "kernel_ms_of_the_profiled_run"):
            util[k] = ent.get(k)
        if ent.get("fp64_flop_per_pair_executed"):
            ex = ent["fp64_flop_per_pair_executed"] * n_contact / (kernel_ms * 1e-3) / 1e12
            util["fp64_executed_tflops"] = ex
            util["fp64_executed_frac_of_peak"] = ex / F64_VALU_PEAK_TFLOPS
        util["note"] = ("pair_contact_kernel, from PMC passes of this workload (static table): valu_busy = SQ_ACTIVE_INST_VALU x 4 / "
                        "(GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); lds_busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE / 8 x 256 CUs); "
                        "lds_bank_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; fp64_instr_share = FP64 FMA + MUL + ADD + "
                        "TRANS instructions / SQ_INSTS_VALU; fp64_executed_* = executed FP64 FLOP (FMA = 2) per pair x pairs / this "
                        "run's kernel time")
    return roof, valu, occ, util


class StaticBed:
    """One static-bed workload of BASELINE.json (configs[1], [2], [4], and configs[0]'s shape at scale) resident in HBM:
    shapes, context, bed, half list, the atom arrays, and the step that is timed — initial_integrate -> clear f /
    torque -> shpair_compute_device() -> final_integrate."""

    def __init__(self, args, passes):
        import torch
        from shpair import shapes, bed
        self.torch, self.args = torch, args
        dev = torch.device("cuda", 0)
        lmax, nshapes = args.lmax, args.nshapes
        self.shp = [shapes.random_shape(lmax, bed.SEED0 + 2 + s) for s in range(nshapes)]
        self.sp = sp = make_ctx(args, self.shp, 0)
        self.rmax = [sp.rmax(s) for s in range(nshapes)]
        self.gbed = gbed = bed.make_bed(args.particles, self.rmax, nshapes, seed=bed.SEED0 + 2)
        self.nlocal = nall = args.particles
        self.il, self.of, self.jl = bed.half_neighbor_list(gbed["x"], gbed["shtype"], self.rmax)
        sp.set_neighbors_csr(self.il, self.of, self.jl)
        self.x = torch.from_numpy(gbed["x"]).to(dev)
        self.q = torch.from_numpy(gbed["quat"]).to(dev)
        self.ty = torch.from_numpy(gbed["type"]).to(dev)
        self.sh = torch.from_numpy(gbed["shtype"]).to(dev)
        self.f = torch.zeros(nall, 3, dtype=torch.float64, device=dev)
        self.tq = torch.zeros_like(self.f)
        self.stream = torch.cuda.current_stream()
        # the integrator either side of the hot path (include/shstep.h).  The bed starts at rest and dt is sized so that
        # nothing moves further than 1e-2 of the neighbour skin during the whole run: the half list (and the
        # contact-pair count) stay valid without a rebuild.
        self.v = torch.zeros(self.nlocal, 3, dtype=torch.float64, device=dev)
        self.angmom = torch.zeros_like(self.v)
        self.mask = torch.ones(self.nlocal, dtype=torch.int32, device=dev)
        self.dt = min(1.0e-4, 4.0e-3 / (passes + 1))

    def integrate(self, phase):
        self.sp.nve_device(phase, self.nlocal, self.dt, self.x.data_ptr(), self.v.data_ptr(), self.q.data_ptr(), self.angmom.data_ptr(),
                           self.f.data_ptr(), self.tq.data_ptr(), self.sh.data_ptr(), self.mask.data_ptr(), stream=self.stream.cuda_stream)

    def step(self, events=None):
        self.integrate(0)
        self.sp.force_clear_device(self.nlocal, self.f.data_ptr(), self.tq.data_ptr(), stream=self.stream.cuda_stream)   # Verlet::force_clear
        if events:
            events[0].record(self.stream)   # HIP events on the stream the pair kernel is launched on
        self.sp.compute_device(self.nlocal, 0, self.x.data_ptr(), self.q.data_ptr(), self.ty.data_ptr(), self.sh.data_ptr(),
                               self.f.data_ptr(), self.tq.data_ptr(), stream=self.stream.cuda_stream)
        if events:
            events[1].record(self.stream)
        self.integrate(1)

    def count(self):
        """untimed: the contact pairs of this bed (static positions)"""
        self.sp.set_option("count", 1)
        self.step()
        self.torch.cuda.synchronize()
        st = self.sp.stats()
        self.sp.set_option("count", 0)
        return st["n_contact"], st["n_touching"]

    def timed(self, untimed, steps):
        """`untimed` passes, then exactly `steps` steps between two synchronize: (seconds, mean kernel ms per step)."""
        torch = self.torch
        for _ in range(untimed):
            self.step()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            self.step(ev[k])
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        fh = self.f[:self.nlocal].cpu().numpy()
        assert np.all(np.isfinite(fh)) and np.abs(fh).max() > 0
        return elapsed, float(np.mean([a.elapsed_time(b) for a, b in ev]))

    def close(self):
        self.sp.close()


# ======================================================================================================== N = 1
def main_single(args):
    import torch
    if _kfd_gpu_nodes() == 0:    # sysfs only: the parent makes no HIP / HSA call before the scale_ref child has come and gone
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    own_stdout()
    wd = Watchdog(args, emit=True)   # only the whole-run bound: one GPU, no peer to wait for
    scale_ref = scale_ref_leg(args) if args.scale_ref else None   # a fresh child process, before this one's first GPU call
    if scale_ref is not None:
        # the child has just released ~2 GB of HBM: the driver frees it in the background; twice in ~40 builder runs a
        # bench pass that started in the first second after another process's exit ran 25 % (once 20x) slow
        time.sleep(1.5)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(0)
    lmax, nq, nshapes = args.lmax, args.nq, args.nshapes
    sb = StaticBed(args, args.steps + args.warmup + args.ramp)
    sp = sb.sp
    n_contact, n_touching = sb.count()
    # ---- timed region: exactly K steps between synchronize (StaticBed.timed)
    elapsed, kernel_ms = sb.timed(args.ramp + args.warmup, args.steps)

    roof, valu, occ, util = roofline_objects(args, sp, n_contact, kernel_ms, 1)
    out = {
        "metric": "contact_pairs_per_sec", "value": n_contact * args.steps / elapsed, "unit": "contact-pairs/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ramp_passes": args.ramp,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[1]: {args.particles} particles/GPU, {nshapes} SH shape(s) L_max={lmax}, dense packed "
                        f"bed (jittered HCP, spacing 1.9 mean radii), n_q={nq} (Q={2 * nq * nq} nodes/pair), "
                        f"pair_coeff kn=1000 exponent={args.exponent} (overlap volume + force + torque), "
                        "inputs resident in HBM",
            "particles_per_gpu": args.particles, "lmax": lmax, "nq": nq, "nshapes": nshapes,
            "exponent": args.exponent, "rule": args.rule, "proc_grid": [1, 1, 1], "backend": "none",
            "half_list_pairs_rank0": int(sb.jl.size), "contact_pairs_rank0": int(n_contact),
            "touching_pairs_rank0": int(n_touching), "contact_pairs_all_ranks": int(n_contact), "ghost_atoms_rank0": 0,
        },
        "timesteps_per_sec": args.steps / elapsed,
        "timestep_note": "one step = initial_integrate + clear + pair compute + final_integrate, "
                         f"dt = {sb.dt:g} from rest, no list rebuild inside the timed steps (see the `timestep` object for whole "
                         "steps with rebuilds)",
        "roofline": roof, "occupancy": occ, "valu_f64": valu, "utilisation": util, "library": _library_name(),
    }
    # everything below comes AFTER the headline's timed region and cannot disturb it
    if args.host_path:
        try:
            out["host_path"] = host_path_leg(args, sb, kernel_ms, n_contact)
        except Exception as e:  # noqa: BLE001 — an extra leg never takes the headline with it
            out["host_path"] = {"error": repr(e)}
    if args.configs:
        out["configs"] = configs_leg(args)
    if args.ts_steps > 0:
        out["timestep"] = timestep_leg(args, sb.shp)
    if scale_ref is not None:
        out["scale_ref"] = scale_ref
    if args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(args, sb.shp, sb.rmax, sb.gbed, sb.il, sb.of, sb.jl)
    out["elapsed_s"] = round(time.monotonic() - T_START, 1)
    emit(json.dumps(out))
    sp.close()
    del wd


CONFIG_LEGS = (
    ("configs[2]", "100k particles, 4 mixed SH shape types L_max=6, n_q=16", dict(nshapes=4, lmax=6, nq=16)),
    ("configs[4]", "100k particles, L_max=12 high-order shape, n_q=32", dict(nshapes=1, lmax=12, nq=32)),
    ("configs[0]-shape", "configs[0]'s L_max=4 / n_q=10 at 100k particles (configs[0] itself is the 1000-particle CPU case: "
                         "tests/golden/cfg1_L4_ellipsoid.npz)", dict(nshapes=1, lmax=4, nq=10)),
)


def configs_leg(args, warm=3, steps=5):
    """The other single-GPU workloads of BASELINE.json, each through the very code of the headline (StaticBed): the
    headline's clock-ramp passes + 3 warm-up + 5 timed steps, a few seconds in all, run after the headline's timed region.  Numbers of a 5-step sample: they
    put every BASELINE config into the driver's own record; the headline stays the K-step figure."""
    out = {"steps": steps, "warmup": warm, "ramp_passes": args.ramp,
           "note": "same step as the headline (initial_integrate + clear + pair compute + final_integrate), same bed generator and "
                   "force law (kn=1000, exponent as --exponent), fresh context per workload; valu_f64_frac / roofline_frac as in the "
                   "headline's objects; utilisation from the static PMC table with its own `stale` flag"}
    for key, what, over in CONFIG_LEGS:
        a2 = argparse.Namespace(**{**vars(args), **over, "particles": 100000, "jpoly": -1, "rule": "sharp", "peak_ms": 0.0})
        try:
            sb = StaticBed(a2, args.ramp + warm + steps + 1)
            nc, nt = sb.count()
            el, kms = sb.timed(args.ramp + warm, steps)   # the clock-ramp passes too: the GPU idled while the host built this bed
            roof, valu, occ, util = roofline_objects(a2, sb.sp, nc, kms, 1)
            out[key] = {
                "workload": what, "lmax": a2.lmax, "nq": a2.nq, "nshapes": a2.nshapes, "exponent": a2.exponent,
                "value": nc * steps / el, "unit": "contact-pairs/s", "ms_per_step": 1e3 * el / steps, "kernel_ms": kms,
                "contact_pairs": int(nc), "touching_pairs": int(nt),
                "valu_f64_frac": valu["frac"], "roofline_frac": roof["frac"], "traffic": roof["traffic"], "stale": roof["stale"],
                "waves_per_pair": occ["waves_per_pair"], "waves_per_cu": occ["waves_per_cu"], "vgprs": occ.get("vgprs"),
                "ring_rows": occ["ring_rows"], "family": occ["family"], "kernel_hash": occ["kernel_hash"],
                "utilisation": {k: util.get(k) for k in ("valu_busy", "lds_busy", "fp64_instr_share", "valu_instr_per_pair",
                                                         "fp64_executed_frac_of_peak", "stale", "source")},
            }
            sb.close()
            del sb
        except Exception as e:  # noqa: BLE001 — an extra leg never takes the headline with it
            out[key] = {"workload": what, "error": repr(e)}
    try:
        out["configs[0]"] = config0_leg()
    except Exception as e:  # noqa: BLE001
        out["configs[0]"] = {"error": repr(e)}
    return out


def config0_leg(steps=200):
    """BASELINE configs[0] itself — 1000 identical L_max = 4 ellipsoid-like particles, gravity-settled on a frozen floor
    (tests/golden/settled_cfg1_L4.npz: positions, orientations and the oracle's forces of the settled bed, a committed data
    file) — on the GPU: ghosts and half list built on the device, then `steps` passes of clear + pair compute + ghost
    reverse.  A 10k-pair list is launch-bound on an MI355X (three kernels of a few microseconds): the number says what
    a small system costs, not what the kernels can do.  The forces are held to the fixture's in the run."""
    import torch
    from shpair import ShPair
    g = np.load(os.path.join(ROOT, "tests", "golden", "settled_cfg1_L4.npz"))
    lmax, nq, n = int(g["lmax"]), int(g["nq"]), g["x"].shape[0]
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, lmax, g["anm"][0])
    sp.coeff(1, 1, float(g["kn"]), float(g["exponent"]))
    sp.set_box(g["lo"], g["hi"], g["periodic"], float(g["skin"]))
    nmax = 3 * n
    dev = torch.device("cuda", 0)
    x = torch.zeros(nmax, 3, dtype=torch.float64, device=dev)
    q = torch.zeros(nmax, 4, dtype=torch.float64, device=dev)
    x[:n] = torch.from_numpy(g["x"]).to(dev)
    q[:n] = torch.from_numpy(g["quat"]).to(dev)
    ty = torch.ones(nmax, dtype=torch.int32, device=dev)
    sh = torch.zeros(nmax, dtype=torch.int32, device=dev)
    f = torch.zeros(nmax, 3, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(f)
    torch.cuda.synchronize()
    ng = sp.borders_device(n, nmax, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
    npairs = sp.neighbor_build_device(n, ng, x.data_ptr(), sh.data_ptr())
    st = torch.cuda.current_stream()

    def one():
        sp.force_clear_device(n + ng, f.data_ptr(), tq.data_ptr(), stream=st.cuda_stream)
        sp.compute_device(n, ng, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(), tq.data_ptr(), stream=st.cuda_stream)
        sp.reverse_device(f.data_ptr(), tq.data_ptr(), stream=st.cuda_stream)
    sp.set_option("count", 1)
    one()
    torch.cuda.synchronize()
    c = sp.stats()
    sp.set_option("count", 0)
    fs = float(np.abs(g["f"]).max())
    err = float(max(np.abs(f[:n].cpu().numpy() - g["f"]).max(), np.abs(tq[:n].cpu().numpy() - g["torque"]).max()) / fs)
    for _ in range(20):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    out = {"workload": "BASELINE configs[0]: 1000 identical L_max=4 ellipsoid-like particles settled under gravity on a 400-particle "
                       "frozen floor (tests/golden/settled_cfg1_L4.npz), n_q=10, periodic in x,y: clear + pair compute + ghost reverse",
           "particles": int(n), "ghosts": int(ng), "half_list_pairs": int(npairs), "contact_pairs": int(c["n_contact"]),
           "touching_pairs": int(c["n_touching"]), "steps": steps, "value": c["n_contact"] * steps / el, "unit": "contact-pairs/s",
           "us_per_step": 1e6 * el / steps, "rel_err_vs_fixture": err, "fixture_ok": bool(err < 1e-9),
           "counts_match_fixture": bool([c["n_candidates"], c["n_contact"], c["n_touching"]] == g["counts"].tolist()),
           "note": "launch-bound (a few microseconds of kernels per pass); the reference's own form of this config is its CPU pair style"}
    sp.close()
    return out


def host_path_leg(args, sb, kernel_ms_device, n_contact=0, calls=7):
    """What an UNMODIFIED LAMMPS pays on top of the kernels (north_star's boundary: PairSH::compute on host arrays): per
    call, shpair_compute() with LAMMPS-layout host arrays page-locked by shpair_pin_host (x, quat, type, shtype up;
    f, torque up, added to on the device, down) — wall time minus the kernels' time = the staging overhead; and, per
    reneighbouring, shpair_set_neighbors(inum, ilist, numneigh, firstneigh) from real per-row pointers (flatten into the
    pinned stage + one upload + expansion on the device).  Never part of `value`."""
    import ctypes as C
    from shpair import capi
    g = sb.gbed
    n = sb.nlocal
    # The headline's own context, as a LAMMPS rank has ONE: its two streams (compute, f / torque upload) were created
    # first in this process and sit on hardware queues of their own.  (A second context's streams share queues with the
    # first's — HIP maps streams onto a few hardware queues round robin — and the upload beside the set-up and rotation
    # kernels then serialises with them: +0.11 ms per call, tools/host_path_probe.py, profiles/r05_e_host_path_probe.txt.)
    sp = getattr(sb, "sp", None) or make_ctx(args, sb.shp, 0)
    own_ctx = sp is not getattr(sb, "sp", None)
    lib = capi.load_library()
    il = np.ascontiguousarray(sb.il, dtype=np.int32)
    of = np.ascontiguousarray(sb.of, dtype=np.int32)
    jl = np.ascontiguousarray(sb.jl, dtype=np.int32)
    # LAMMPS' NeighList: numneigh and firstneigh are indexed by ATOM, firstneigh[i] points at row i's neighbours
    numneigh = np.zeros(n, dtype=np.int32)
    numneigh[il] = np.diff(of)
    rows = np.zeros(n, dtype=np.uint64)
    rows[il] = jl.ctypes.data + 4 * of[:-1].astype(np.uint64)
    ip = C.POINTER(C.c_int)
    first = rows.ctypes.data_as(C.POINTER(ip))

    def t_list(fn):
        ts = []
        for _ in range(5):
            t = time.perf_counter()
            rc = fn()
            ts.append(1e3 * (time.perf_counter() - t))
            assert rc == 0, rc
        return min(ts), float(np.median(ts))
    rows_min, rows_med = t_list(lambda: lib.shpair_set_neighbors(sp._h, il.size, il.ctypes.data_as(ip), numneigh.ctypes.data_as(ip), first))
    csr_min, csr_med = t_list(lambda: lib.shpair_set_neighbors_csr(sp._h, il.size, il.ctypes.data_as(ip), of.ctypes.data_as(ip),
                                                                      jl.ctypes.data_as(ip)))
    rc = lib.shpair_set_neighbors(sp._h, il.size, il.ctypes.data_as(ip), numneigh.ctypes.data_as(ip), first)
    assert rc == 0
    sp.set_option("timing", 1)
    x, q, ty, sh = (np.ascontiguousarray(g[k]) for k in ("x", "quat", "type", "shtype"))
    fb, tb = np.zeros((n, 3)), np.zeros((n, 3))

    def calls_ms(k):
        w, km = [], []
        for _ in range(k):
            fb[:] = 0.0
            tb[:] = 0.0
            t = time.perf_counter()
            sp.compute(n, x, q, ty, sh, f=fb, torque=tb)
            w.append(1e3 * (time.perf_counter() - t))
            km.append(sp.stats()["kernel_ms"])
        return w, km
    calls_ms(2)
    w_page, k_page = calls_ms(3)
    arrs = [x, q, ty, sh, fb, tb]
    for a_ in arrs:
        sp.pin_host(a_)
    calls_ms(1)
    w_pin, k_pin = calls_ms(calls)
    for a_ in arrs:
        sp.unpin_host(a_)
    assert np.all(np.isfinite(fb)) and np.abs(fb).max() > 0
    up = x.nbytes + q.nbytes + ty.nbytes + sh.nbytes + fb.nbytes + tb.nbytes
    down = fb.nbytes + tb.nbytes
    i_pin = int(np.argmin(w_pin))
    i_page = int(np.argmin(w_page))
    out = {
        "workload": "the headline's bed and list (BASELINE configs[1]) through the HOST-pointer entry points of include/shpair.h, "
                    "as lammps/pair_sh.cpp calls them",
        "compute_call_ms_pinned": w_pin[i_pin], "compute_call_ms_pinned_median": float(np.median(w_pin)),
        "compute_kernel_ms": k_pin[i_pin], "compute_overhead_ms_pinned": w_pin[i_pin] - k_pin[i_pin],
        "compute_call_ms_pageable": w_page[i_page], "compute_overhead_ms_pageable": w_page[i_page] - k_page[i_page],
        "contact_pairs_per_sec_pinned": (n_contact / (w_pin[i_pin] * 1e-3) if n_contact else None), "bytes_up_per_call": int(up), "bytes_down_per_call": int(down), "calls": calls,
        "set_neighbors_ms": rows_min, "set_neighbors_ms_median": rows_med, "set_neighbors_csr_ms": csr_min,
        "set_neighbors_bytes_uploaded": int(4 * (2 * il.size + 1 + jl.size)), "half_list_pairs": int(jl.size),
        "device_resident_kernel_ms": kernel_ms_device,
        "note": "PCIe-inclusive: a reported cost of the drop-in boundary, never `value`.  compute_overhead = wall time of one "
                "shpair_compute() (upload, kernels, download, host-side synchronisation) minus the hipEvent time of its kernels; "
                "set_neighbors = one call per reneighbouring, from firstneigh row pointers indexed by atom (min of 5 / median)",
    }
    sp.set_option("timing", 0)
    if own_ctx:
        sp.close()
    return out


def timestep_leg(args, shp):
    """Whole device-resident timesteps per second (the second half of BASELINE.json's metric): NVE run of
    a fully periodic dense bed of the same shapes — integrate, rebuild test, ghosts, pair forces, reverse,
    integrate — everything through the C ABI (include/shstep.h), nothing on the host but launches."""
    import torch
    from shpair import bed
    from shpair.run import DeviceRun
    sp = make_ctx(args, shp, 0)
    pts, lo, hi = bed.periodic_hcp(args.particles, 1.9, (1, 1, 1))
    rng = np.random.default_rng(bed.SEED0 + 7)
    n = pts.shape[0]
    pts = pts + rng.uniform(-0.04, 0.04, pts.shape)
    quat = bed.random_quaternions(n, rng)
    shtype = rng.integers(0, args.nshapes, n).astype(np.int32) if args.nshapes > 1 else np.zeros(n, np.int32)
    skin, dt = 0.1, 1.0e-3
    run = DeviceRun(sp, pts, quat, shtype, lo, hi, (1, 1, 1), skin, dt=dt, device="cuda:0")
    run.run(5)
    sp.set_option("count", 1)
    run.force()
    torch.cuda.synchronize()
    contact0 = sp.stats()["n_contact"]
    sp.set_option("count", 0)
    b0 = run.builds
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run.run(args.ts_steps)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    rebuilds = run.builds - b0
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    run.rebuild()
    torch.cuda.synchronize()
    rebuild_ms = 1e3 * (time.perf_counter() - t1)
    run.force(eflag=True)
    pe, kt, kr, _ = run.energies()
    out = {"timesteps_per_s": args.ts_steps / el, "ms_per_step": 1e3 * el / args.ts_steps, "steps": args.ts_steps,
           "particles": int(n), "ghosts": int(run.nghost), "half_list_pairs": int(run.npairs),
           "contact_pairs": int(contact0), "particle_steps_per_s": n * args.ts_steps / el,
           "rebuilds_in_timed_steps": int(rebuilds), "rebuild_ms": rebuild_ms, "dt": dt, "skin": skin,
           "periodic": [1, 1, 1], "energy": {"contact": pe, "ke_trans": kt, "ke_rot": kr},
           "what": "initial_integrate + rebuild test + forward + clear + pair compute + reverse + final_integrate, "
                   "all arrays resident in HBM (shpair.run.DeviceRun over include/shpair.h + include/shstep.h)"}
    sp.close()
    return out


def scale_ref_leg(args):
    """The N > 1 workload (BASELINE configs[3]: 125k particles per GPU, box periodic in x and y on a frozen floor,
    gravity, thermal start, whole timesteps of shhalo_run_device with rebuild tests, migration bookkeeping and ghost
    exchange) on ONE rank through the same code as `bench.py --gpus N`: RCCL self-communicator, grid 1x1x1.  The
    like-for-like N = 1 point of the scaling curve: efficiency(N) = value(N) / (N x scale_ref.value).

    Runs as a FRESH child process (`bench.py --gpus 1 --multi`, started through the self-launcher's machinery) BEFORE
    this process makes its first GPU call: the leg's watchdog ends a stalled ncclCommInitRank with os._exit(4), which
    must never take the headline line with it, and two HIP contexts never share the card."""
    argv = ["--gpus", "1", "--multi", "--particles", "125000", "--steps", str(args.steps), "--warmup", str(args.warmup),
            "--ramp", str(args.ramp), "--lmax", str(args.lmax), "--nq", str(args.nq), "--nshapes", str(args.nshapes),
            "--exponent", repr(float(args.exponent)), "--jpoly", str(args.jpoly), "--rule", args.rule, "--vthermal", repr(float(args.vthermal)),
            "--no-verify", "--peak-ms", "0", "--wait-s", str(args.wait_s), "--halo-overlap", str(args.halo_overlap),
            "--halo-stream-priority", str(args.halo_stream_priority),
            "--total-s", f"{max(20.0, min(3.0 * args.wait_s, 0.4 * args.total_s if args.total_s > 0 else 1e9)):.0f}"]
    try:
        rc, out = run_rank_children(argv, 1, max(25.0, min(3.0 * args.wait_s, 0.4 * args.total_s if args.total_s > 0 else 1e9) + 5.0))
        lines = [ln for ln in out.splitlines() if ln.startswith("{")]
        ln = json.loads(lines[-1]) if lines else None
        if ln is not None and ln.get("error"):
            return {"error": ln["error"], "exit_code": rc}
        if ln is None or (rc != 0 and ln.get("verify_overlap_ok") is not False):   # exit code 1 with a line: the overlap check failed, the line says so
            return {"error": f"child `bench.py {' '.join(argv)}` ended with exit code {rc} and {len(lines)} JSON line(s)"}
        return {"value": ln["value"], "unit": ln["unit"], "ms_per_step": ln["ms_per_step"], "timesteps_per_sec": ln["timesteps_per_sec"],
                "steps": ln["steps"], "particles": ln["config"]["particles_all_ranks"],
                "contact_pairs": ln["config"]["contact_pairs_all_ranks"], "ghost_atoms": ln["config"]["ghost_atoms_rank0"],
                "pair_kernel_ms": ln["roofline"]["kernel_ms"], "overlap_used": ln.get("overlap_used"),
                "overlap_stream_priority_used": ln.get("overlap_stream_priority_used"),
                "verify_overlap_ok": ln.get("verify_overlap_ok"), "overlap_ab_ms": ln.get("overlap_ab_ms"),
                "transport": ln["halo"]["transport"], "ranks_reported_by_transport": ln["halo"]["ranks_reported_by_transport"],
                "rebuilds_in_timed_steps": ln["halo"]["rebuilds_in_timed_steps"][0], "workload": ln["config"]["workload"],
                "library": ln.get("library"), "cmd": "python bench.py " + " ".join(argv),
                "use": "the N = 1 reference of the scaling curve: parallel efficiency of `bench.py --gpus N` = value(N) / (N x "
                       "scale_ref.value) — same workload per GPU, same C++ loop, same transport code, measured in a fresh child "
                       "process before the headline; the headline `value` of this line is BASELINE configs[1] (static bed, no "
                       "halo) and is NOT that reference"}
    except Exception as e:  # noqa: BLE001 — the headline line must not die with this leg
        return {"error": repr(e)}


# ======================================================================================================== N > 1
def config4_bed(args, world, grid):
    """BASELINE configs[3] as a synthetic bed: `particles` per rank on an HCP lattice in a box periodic in x and y
    whose aspect follows the processor grid (equal bricks), one frozen layer as the floor, thermal velocities."""
    from shpair import bed
    n_target = args.particles * world
    dx, dy, dz = 1.9, 1.9 * np.sqrt(3.0) / 2.0, 1.9 * np.sqrt(2.0 / 3.0)
    s = (n_target * dx * dy * dz / (grid[0] * grid[1] * grid[2])) ** (1.0 / 3.0)   # brick edge
    nx = max(2, int(round(grid[0] * s / dx)))
    ny = max(2, 2 * int(round(grid[1] * s / dy / 2)))
    nz = max(2, 2 * int(round(n_target / (nx * ny) / 2)))
    k, j, i = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    # no offset in x and y: lattice planes then coincide with the brick faces, so that the thermal motion carries atoms
    # across them (migration) from the first rebuild on
    pts = np.stack([((i + 0.5 * (j % 2) + 0.5 * (k % 2)) * dx).ravel(), ((j + (k % 2) / 3.0) * dy).ravel(),
                    (k * dz).ravel() + 0.25 * 1.9], axis=1)
    lo = np.zeros(3)
    hi = np.array([nx * dx, ny * dy, nz * dz + 1.9])
    pts[:, 0] = np.mod(pts[:, 0], hi[0])
    pts[:, 1] = np.mod(pts[:, 1], hi[1])
    rng = np.random.default_rng(bed.SEED0 + 4)
    n = pts.shape[0]
    x = pts + rng.uniform(-0.04, 0.04, pts.shape)
    quat = bed.random_quaternions(n, rng)
    shtype = rng.integers(0, args.nshapes, n).astype(np.int32) if args.nshapes > 1 else np.zeros(n, np.int32)
    mask = np.where(pts[:, 2] < 0.25 * 1.9 + 0.5 * dz, 2, 1).astype(np.int32)    # the bottom layer is the floor
    v = args.vthermal * rng.normal(size=(n, 3))
    v[mask == 2] = 0.0
    return dict(x=x, quat=quat, shtype=shtype, mask=mask, v=v, tag=np.arange(n, dtype=np.int32), lo=lo, hi=hi,
                periodic=(1, 1, 0), n=n)


class _Collective:
    """What the ranks need of each other besides the data path: a barrier, max / sum of a few host numbers, one
    broadcast.  torch.distributed (gloo) between processes, a threading.Barrier between rank threads."""

    def __init__(self, world, dist=None):
        self.world, self.dist = world, dist
        if dist is None:
            self.bar = threading.Barrier(world)
            self.slots = [None] * world

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        else:
            self.bar.wait()

    def gather(self, rank, value):
        """List of every rank's value (on every rank)."""
        if self.dist is not None:
            out = [None] * self.world
            self.dist.all_gather_object(out, value)
            return out
        self.slots[rank] = value
        self.bar.wait()
        out = list(self.slots)
        self.bar.wait()
        return out


def _cmp_owned(a, b, box):
    """Largest deviation between two (tag, x, v, quat, f, torque) snapshots of one rank's owned atoms sorted by tag:
    (|dx| / box edge, |df| and |dtorque| absolute, max |f| of the first) — inf when the two hold different atoms."""
    if a[0].shape != b[0].shape or not np.array_equal(a[0], b[0]):
        return float("inf"), float("inf"), 0.0
    if a[0].size == 0:
        return 0.0, 0.0, 0.0
    ex = float(np.abs(a[1] - b[1]).max() / box)
    ef = float(max(np.abs(a[4] - b[4]).max(), np.abs(a[5] - b[5]).max()))
    return ex, ef, float(np.abs(a[4]).max())


def multi_rank_body(args, rank, world, device, coll, hub, uid, result, wd=None, dist=None):
    """One rank of the N > 1 run.  Transports (include/shhalo.h): uid given -> RCCL (ncclSend / ncclRecv between the GPUs);
    hub given -> rank threads of one process; neither, with `dist` -> host-staged through torch.distributed's CPU
    backend (--transport staged).  An RCCL attempt that fails cleanly on any rank (ncclCommInitRank returns an error),
    or whose decomposed forces are wrong, is replaced by the host-staged transport on EVERY rank — agreed on through the
    control plane — and the line says so (`halo.transport`, `transport_fallback`): a slower curve with a flag instead
    of none."""
    import torch
    from shpair import shapes, bed, mrank
    from shpair.capi import ShPairError
    wd = wd or Watchdog(args, emit=(rank == 0))    # (rank threads of --transport local: one watchdog each; rank 0's owns the line)
    setup = {}
    t_lap = [time.perf_counter()]

    def lap(name):
        now = time.perf_counter()
        setup[name] = round(now - t_lap[0], 3)
        t_lap[0] = now
    torch.cuda.set_device(device)
    shp = [shapes.random_shape(args.lmax, bed.SEED0 + 2 + s) for s in range(args.nshapes)]
    sp = make_ctx(args, shp, device)
    # "halo_overlap": the run starts at 0 (the exchanges and the pair kernels follow each other on one stream); the
    # candidate — 2 unless --halo-overlap names another — is used only after it has been checked against 0 IN THIS RUN
    candidate = 2 if args.halo_overlap < 0 else args.halo_overlap
    check_overlap = bool(args.verify_overlap) and candidate > 0
    sp.set_option("halo_overlap", 0 if check_overlap else candidate)
    sp.set_option("halo_stream_priority", max(0, args.halo_stream_priority))
    skin = 0.1
    grid = mrank.proc_grid(world)
    cfg = config4_bed(args, world, grid)
    lap("bed_on_host")
    cut = 2.0 * max(sp.rmax(s) for s in range(args.nshapes)) + skin
    geo = mrank.plan_geometry(grid, cfg["lo"], cfg["hi"], cfg["periodic"], cut, rank)
    xw, owner = mrank.plan_owner(geo, cfg["x"])
    mine = owner == rank
    lap("owner_plan")
    dt = 1.0e-3
    fallback = None
    staged_keep = []
    selftest = {}

    def make_halo(kind, ctx):
        """kind: "rccl" | "local" | "staged".  Collective.  Returns (halo or None, error text or None) — agreed on by all ranks."""
        what = {"rccl": "ncclCommInitRank (shhalo_create_rccl)", "local": "shhalo_create_local", "staged": "shhalo_create_staged"}[kind]
        with wd.phase(what, args.wait_s):
            h, err = None, None
            try:
                if kind == "rccl":
                    if os.environ.get("SHPAIR_BENCH_FAULT") == "rccl_init":    # diagnostic hook (tests): RCCL refuses on every rank
                        raise ShPairError(-5, "diagnostic: ncclCommInitRank refused (SHPAIR_BENCH_FAULT=rccl_init)")
                    h = mrank.Halo(ctx, rank, world, grid, cfg["lo"], cfg["hi"], cfg["periodic"], skin, unique_id_bytes=uid)
                    # a megabyte to itself through ncclSend / ncclRecv, and an all-reduce over all ranks, checked byte by byte,
                    # before anything depends on the wire (shhalo_transport_selftest)
                    h.transport_selftest(1 << 20, ctx.own_stream())
                    selftest["rccl"] = "ok: 1 MiB sent to self through ncclSend / ncclRecv and all-reduces over all ranks, checked"
                elif kind == "staged":
                    g = mrank.GlooStaged(dist)
                    staged_keep.append(g)
                    h = mrank.Halo(ctx, rank, world, grid, cfg["lo"], cfg["hi"], cfg["periodic"], skin, staged=g)
                else:
                    h = mrank.Halo(ctx, rank, world, grid, cfg["lo"], cfg["hi"], cfg["periodic"], skin, hub=hub)
            except ShPairError as e:
                err = f"rank {rank}: {e}"
            if kind == "local":
                if err:
                    raise RuntimeError(err)
                return h, None
            errs = [e for e in coll.gather(rank, err) if e]
            if errs:
                if h is not None:
                    h.close()
                return None, errs[0]
            return h, None

    def make_run(h, ctx):
        with wd.phase("first migration + ghost plan + list build + forces (RankRun)", args.wait_s):
            return mrank.RankRun(ctx, h, xw[mine], cfg["quat"][mine], cfg["shtype"][mine], cfg["tag"][mine], v=cfg["v"][mine],
                                 mask=cfg["mask"][mine], dt=dt, gravity=(0.0, 0.0, -1.0), device=f"cuda:{device}",
                                 capacity=int(1.5 * mine.sum()) + 4096)

    def verify_forces(r):
        """Decomposed forces of the initial configuration against a single-domain compute on rank 0; the same number on every rank."""
        with wd.phase("gather of the initial forces (verify)", args.wait_s):
            t, _, _, _, f0, tq0 = r.owned()
            parts = coll.gather(rank, (t, f0, tq0))
        verr = None
        if rank == 0:
            with wd.phase("single-domain reference forces on rank 0 (verify)", args.wait_s):
                from shpair.run import DeviceRun
                ref_sp = make_ctx(args, shp, device)
                ref = DeviceRun(ref_sp, cfg["x"], cfg["quat"], cfg["shtype"], cfg["lo"], cfg["hi"], cfg["periodic"], skin, mask=cfg["mask"],
                                dt=dt, gravity=(0.0, 0.0, -1.0), device=f"cuda:{device}")
                ref.v[:] = torch.from_numpy(cfg["v"]).to(ref.v.device)
                ref.force()
                torch.cuda.synchronize()
                n = cfg["n"]
                fr, tr = ref.f[:n].cpu().numpy(), ref.tq[:n].cpu().numpy()
                fg, tg = np.zeros_like(fr), np.zeros_like(tr)
                for t_, f_, q_ in parts:
                    fg[t_] = f_
                    tg[t_] = q_
                verr = float(max(np.abs(fg - fr).max(), np.abs(tg - tr).max()) / np.abs(fr).max())
                if os.environ.get("SHPAIR_BENCH_FAULT") == "rccl_forces" and r.halo.stats()["transport"] == 1:
                    verr = 1.0    # diagnostic hook (tests): the RCCL attempt's forces are declared wrong
                if not verr < 1e-9:   # reported in the line (verify_ok) and by the exit code; the run goes on so that every rank ends together
                    print(f"bench.py: decomposed forces differ from single-domain forces: rel err {verr}", file=sys.stderr, flush=True)
                ref_sp.close()
                del ref
        with wd.phase("agreement on the verification (verify)", 2 * args.wait_s):
            verr = [v for v in coll.gather(rank, verr) if v is not None][0]
        return verr

    kind = "rccl" if uid is not None else ("local" if (hub is not None or dist is None) else "staged")
    halo, herr = make_halo(kind, sp)
    if halo is None and kind == "rccl" and dist is not None:
        fallback = f"shhalo_create_rccl failed ({herr})"
        if rank == 0:
            print(f"bench.py: {fallback}: every rank falls back to the host-staged transport", file=sys.stderr, flush=True)
        kind = "staged"
        halo, herr = make_halo(kind, sp)
    if halo is None:
        raise RuntimeError(f"no transport: {herr}")
    lap("comm_init")
    run = make_run(halo, sp)
    lap("first_build")
    if os.environ.get("SHPAIR_BENCH_FAULT") == "die_rank1" and rank == 1:    # diagnostic hook (tests): a rank is lost in mid-run
        print("bench.py: rank 1: diagnostic exit (SHPAIR_BENCH_FAULT=die_rank1)", file=sys.stderr, flush=True)
        os._exit(9)
    verify_err = None
    if args.verify:
        verify_err = verify_forces(run)
        if not verify_err < 1e-9 and kind == "rccl" and dist is not None:
            fallback = f"the decomposed forces over RCCL were wrong (rel err {verify_err:.3g})"
            if rank == 0:
                print(f"bench.py: {fallback}: every rank falls back to the host-staged transport", file=sys.stderr, flush=True)
            del run
            halo.close()
            sp.close()
            sp = make_ctx(args, shp, device)
            sp.set_option("halo_overlap", 0 if check_overlap else candidate)
            sp.set_option("halo_stream_priority", max(0, args.halo_stream_priority))
            kind = "staged"
            halo, herr = make_halo(kind, sp)
            if halo is None:
                raise RuntimeError(f"no transport: {herr}")
            run = make_run(halo, sp)
            verify_err = verify_forces(run)
        lap("verify")

    def count_contacts():
        sp.set_option("count", 1)
        run.force()
        st = sp.stats()
        sp.set_option("count", 0)
        return st["n_contact"], st["n_touching"]

    def mode_key(m):
        return str(m[0]) + ("p" if m[1] else "")

    def set_mode(m):
        sp.set_option("halo_overlap", m[0])
        sp.set_option("halo_stream_priority", m[1])
    prios = [0, 1] if args.halo_stream_priority < 0 else [args.halo_stream_priority]
    cands = [(candidate, pr) for pr in prios] if candidate > 0 else []
    base = (0, 0) if (check_overlap or not cands) else cands[0]    # the mode of the FIRST timed region: plain unless asked otherwise without a check
    ov = {"requested": args.halo_overlap, "candidate": candidate, "checked": False, "used": base[0], "prio_used": base[1]}
    set_mode(base)

    def timed_region(nwarm):
        """nwarm untimed timesteps (in chunks of 4, so that rebuilds happen too), the contact count, then exactly K timesteps
        between barriers and device synchronisations; this rank's numbers."""
        with wd.phase("warm-up timesteps", 2 * args.wait_s):
            for _ in range(max(1, nwarm // 4)):
                run.run(4)
            c0, t0_ = count_contacts()
        b0, k0 = run.builds, run.kernel_ms
        s0 = halo.stats()
        with wd.phase("timed timesteps", 2 * args.wait_s + 0.1 * args.steps):   # (a long run asked for by hand gets a bound that grows with it; --total-s still applies)
            run.sync()
            coll.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run.run(args.steps, timed=True)
            torch.cuda.synchronize()
            coll.barrier()
            elapsed = time.perf_counter() - t0
        with wd.phase("contact count + gather of the results", args.wait_s):
            c1, t1_ = count_contacts()
            s1 = halo.stats()
            mine_out = dict(elapsed=elapsed, contact=0.5 * (c0 + c1), touching=0.5 * (t0_ + t1_), kernel_ms=(run.kernel_ms - k0) / args.steps,
                            rebuilds=run.builds - b0, migrated=s1["migrated_out"] - s0["migrated_out"], nlocal=run.n, nghost=run.nghost,
                            npairs=run.npairs, stats=s1, setup=dict(setup))
            return coll.gather(rank, mine_out)

    def build_line(allr, timed_by_mode, setups):
        el = max(r["elapsed"] for r in allr)
        contact_all = sum(r["contact"] for r in allr)
        assert sum(r["nlocal"] for r in allr) == cfg["n"], "atoms lost"
        roof, valu, occ, util = roofline_objects(args, sp, allr[0]["contact"], allr[0]["kernel_ms"], world)
        roof["kernel_ms_note"] = ("sum of the hipEvent pairs around each slot range of a step (with halo_overlap up to three): the pair "
                                  "kernels only, the waits for the exchange between the ranges are not in it")
        st = allr[0]["stats"]
        return {
            "metric": "contact_pairs_per_sec", "value": contact_all * args.steps / el, "unit": "contact-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ramp_passes": args.ramp,
            "ms_per_step": 1e3 * el / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"BASELINE configs[3]: {args.particles} particles/GPU ({cfg['n']} in all), {args.nshapes} SH shape(s) "
                            f"L_max={args.lmax}, n_q={args.nq}, exponent={args.exponent}, box periodic in x,y on a frozen floor, gravity, "
                            f"thermal start (|v| ~ {args.vthermal}), dt={dt}, skin {skin}: whole timesteps with rebuild tests, atom "
                            "migration, ghost exchange (shhalo_run_device); arrays resident in HBM",
                "particles_per_gpu": args.particles, "particles_all_ranks": int(cfg["n"]), "lmax": args.lmax, "nq": args.nq,
                "nshapes": args.nshapes, "exponent": args.exponent, "rule": args.rule, "proc_grid": list(grid),
                "backend": {1: "rccl", 2: "host-staged (torch.distributed gloo) — " + ("FALLBACK: " + fallback if fallback else "asked for"),
                            0: "local-hub-rehearsal (rank threads on one GPU)"}[st["transport"]],
                "contact_pairs_rank0": int(allr[0]["contact"]), "contact_pairs_all_ranks": int(contact_all),
                "ghost_atoms_rank0": int(allr[0]["nghost"]), "half_list_pairs_rank0": int(allr[0]["npairs"]),
            },
            "timesteps_per_sec": args.steps / el,
            "halo": {
                "transport": {0: "local", 1: "rccl", 2: "staged"}.get(st["transport"], str(st["transport"])),
                "ranks_reported_by_transport": st["nranks_transport"], "rccl_version": st["rccl_version"],
                "overlap_option": ov["used"], "peers_rank0": st["npeers"], "send_rows_rank0": st["nsend_rows"], "ghost_rows_rank0": st["nghost_rows"],
                "forward_bytes_per_step_rank0": st["forward_bytes_per_step"], "reverse_bytes_per_step_rank0": st["reverse_bytes_per_step"],
                "rebuilds_in_timed_steps": [r["rebuilds"] for r in allr], "atoms_migrated_in_timed_steps": int(sum(r["migrated"] for r in allr)),
                "owned_atoms": [r["nlocal"] for r in allr], "ghost_atoms": [r["nghost"] for r in allr],
                "what": "per step and direction of travel one pack kernel, one ncclGroupStart..ncclSend/ncclRecv per peer.."
                        "ncclGroupEnd, one unpack kernel; no host wait except at the rebuild test.  overlap_option 0: all of it on "
                        "the compute stream; 1 / 2: the forward (and the reverse) exchange on a second stream beside the pair "
                        "kernels of the owned-only slots",
            },
            "transport_fallback": fallback, "transport_selftest": selftest.get("rccl"), "timed_by_mode": timed_by_mode,
            "overlap_requested": ov["requested"], "overlap_candidate": ov["candidate"], "overlap_used": ov["used"],
            "overlap_stream_priority_used": ov["prio_used"],
            "verify_overlap_rel_err": ov.get("rel_err"), "verify_overlap_ok": ov.get("ok"), "verify_overlap_rel_err_by_mode": ov.get("rel_err_by_mode"),
            "overlap_ab_ms": ov.get("ab_ms_per_step"), "overlap_ab_steps": ov.get("ab_steps"),
            "verify_overlap_note": "the path the timed steps take, checked in this run: from one saved state (x, v, quat, angmom of "
                                   "every rank) 4 timesteps of shhalo_run_device with halo_overlap 0 and again with overlap_candidate — on an "
                                   "ordinary second stream (key \"2\") and on one at the highest stream priority (key \"2p\"); "
                                   "owned positions (per box edge), forces and torques (per max |F|) compared by tag on every rank, bar "
                                   "1e-9; overlap_ab_ms: ms per timestep of both over overlap_ab_steps steps from that state (min of 2 "
                                   "legs each, max over ranks).  A mode whose check fails is not used (and the exit code is 1); with "
                                   "--halo-overlap -1 the fastest of 0 and the correct modes is used, else the fastest correct mode",
            "value_note": "contact pairs of all ranks (mean of the counts before and after the timed steps) x K / max-over-ranks time",
            "verify_rel_err": verify_err, "verify_ok": (None if verify_err is None else bool(verify_err < 1e-9)),
            "verify_note": "decomposed forces and torques of the initial configuration against a single-domain compute of the "
                           "whole bed on rank 0 (max abs difference / max |F|), untimed; bar 1e-9",
            "setup_s": {"rank0": setups[0], "max_over_ranks": {k: max(q.get(k, 0.0) for q in setups) for k in setups[0]}},
            "elapsed_s": round(time.monotonic() - T_START, 1),
            "roofline": roof, "occupancy": occ, "valu_f64": valu, "utilisation": util, "library": _library_name(),
            "scale_ref_cmd": f"python bench.py --gpus 1 --multi --steps {args.steps} --warmup {args.warmup}",
            "scale_ref_note": "parallel efficiency of this line = value / (n_gpus x value of scale_ref_cmd's line): the same workload "
                              "per GPU through the same C++ loop and transport on one rank; the default N = 1 line carries that "
                              "number as its `scale_ref` object",
        }

    allr = timed_region(args.ramp + args.warmup)
    lap("first_timed_region")
    timed = {mode_key(base): dict(ms_per_step=1e3 * max(r["elapsed"] for r in allr) / args.steps,
                                  value=sum(r["contact"] for r in allr) * args.steps / max(r["elapsed"] for r in allr))}
    if rank == 0:
        result["line"] = build_line(allr, timed, [r["setup"] for r in allr])
    wd.secure(result.get("line", True))     # from here on every leg is optional: a leg that does not come back leaves with this line

    # ---- the path the timed steps take, checked in the run itself: from ONE saved state, k timesteps of shhalo_run_device
    # with "halo_overlap" 0 and again with every candidate mode — the overlap value on an ordinary second stream and on
    # one at the highest stream priority ("halo_stream_priority") — owned x / f / torque compared by tag on every rank
    # (4 steps: no chaos yet); then all modes are timed.  A candidate that differs is not used (and the exit code says so).
    used = base
    if check_overlap:
        box = float(np.max(cfg["hi"] - cfg["lo"]))
        with wd.phase("halo_overlap check: 4 timesteps at 0 and at each candidate from one saved state", 2 * args.wait_s):
            if os.environ.get("SHPAIR_BENCH_FAULT") == "overlap_stall" and rank == world - 1:    # diagnostic hook (tests): an optional leg hangs
                time.sleep(1.0e6)
            state = run.save_state()

            def leg(m, nsteps):
                set_mode(m)
                run.restore_state(state)    # collective: migration, ghost plan, list (partitioned when overlap > 0), forces
                run.sync()
                coll.barrier()
                t0 = time.perf_counter()
                run.run(nsteps)
                run.sync()
                coll.barrier()
                return time.perf_counter() - t0
            leg((0, 0), 4)
            ref_owned = run.owned()
            errs_of = {}
            for m in cands:
                leg(m, 4)
                got = run.owned()
                if os.environ.get("SHPAIR_BENCH_FAULT") == "overlap" and rank == 0 and got[4].size:
                    got[4][0, 0] += 1e-5 * max(1.0, float(np.abs(got[4]).max()))   # diagnostic hook (tests): a wrong candidate
                errs = coll.gather(rank, _cmp_owned(ref_owned, got, box))
                fscale = max(e[2] for e in errs) or 1.0
                errs_of[m] = max(max(e[0] for e in errs), max(e[1] for e in errs) / fscale)
                del got
            del ref_owned
            good = [m for m in cands if errs_of[m] < 1e-9]
        ov.update(checked=True, rel_err=max(errs_of.values()), ok=(len(good) == len(cands)), steps=4,
                  rel_err_by_mode={mode_key(m): errs_of[m] for m in cands})
        with wd.phase("halo_overlap A/B timing", 2 * args.wait_s):
            modes = [(0, 0)] + cands
            ab = {m: [] for m in modes}
            for _ in range(2):
                for m in modes:
                    el = leg(m, args.ab_steps)
                    ab[m].append(max(coll.gather(rank, el)))     # max over ranks, as the timed region
            ms = {m: 1e3 * min(v) / args.ab_steps for m, v in ab.items()}
            ov["ab_ms_per_step"] = {mode_key(m): ms[m] for m in modes}
            ov["ab_steps"] = args.ab_steps
            for m in cands:
                if m not in good and rank == 0:
                    print(f"bench.py: halo_overlap {mode_key(m)} differs from 0 after 4 timesteps (rel err {errs_of[m]}): not used",
                          file=sys.stderr, flush=True)
            pool = good + ([(0, 0)] if (args.halo_overlap < 0 or not good) else [])   # auto: 0 competes; asked for: only if nothing is right
            used = min(pool, key=lambda m: ms[m])
            set_mode(used)
            run.restore_state(state)
            del state
        lap("overlap_check")
        # the winner, if it is not the mode already timed, gets its own K timed steps; the line reports the faster of the two
        if used != base:
            allr2 = timed_region(4)
            ms2 = 1e3 * max(r["elapsed"] for r in allr2) / args.steps
            timed[mode_key(used)] = dict(ms_per_step=ms2, value=sum(r["contact"] for r in allr2) * args.steps / max(r["elapsed"] for r in allr2))
            if args.halo_overlap >= 0 or ms2 < timed[mode_key(base)]["ms_per_step"]:    # asked for: its number, whatever it is
                allr = allr2
            else:
                used = base
        ov["used"], ov["prio_used"] = used
        with wd.phase("gather of the set-up times", args.wait_s):
            setups = coll.gather(rank, dict(setup))
        if rank == 0:
            result["line"] = build_line(allr, timed, setups)
        wd.secure(result.get("line", True))

    with wd.phase("final barrier", args.wait_s):
        coll.barrier()
    halo.close()
    sp.close()


def _line_rc(line):
    return 1 if (line.get("verify_ok") is False or line.get("verify_overlap_ok") is False) else 0


def main_multi(args):
    own_stdout()
    wd = Watchdog(args, emit=(os.environ.get("RANK", "0") == "0"))
    wd.watch_sigterm()
    if os.environ.get("SHPAIR_BENCH_FAULT") == "stall":    # diagnostic hook (tests, no GPU needed): a rank that never comes back
        with wd.phase("diagnostic stall (SHPAIR_BENCH_FAULT=stall)", args.wait_s):
            time.sleep(1.0e6)
    import torch
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    world = args.gpus
    result = {}
    if args.transport == "local":
        from shpair import mrank
        hub = mrank.Hub(world)
        coll = _Collective(world)
        errs = []

        def work(r):
            try:
                multi_rank_body(args, r, world, 0, coll, hub, None, result)   # a watchdog per rank thread
            except BaseException as e:  # noqa: BLE001
                import traceback
                if "line" in result:    # an optional leg failed after the measurement: the measurement stands
                    print(traceback.format_exc(), file=sys.stderr, flush=True)
                    emit(json.dumps(dict(result["line"], experiment_error=f"rank {r}: {e!r}")))
                    os._exit(0)
                errs.append(traceback.format_exc())
                try:
                    coll.bar.abort()
                except Exception:  # noqa: BLE001
                    pass
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            print(errs[0], file=sys.stderr)
            sys.exit(1)
        emit(json.dumps(result["line"]))
        hub.close()
        sys.exit(_line_rc(result["line"]))
    wsz = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if wsz != world:
        if rank == 0:
            print(f"bench.py: --gpus {world} but WORLD_SIZE={wsz}; launch with torch.distributed.run (or --transport local)",
                  file=sys.stderr)
        sys.exit(2)
    import datetime
    import torch.distributed as dist
    from shpair import mrank
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world == 1 and "MASTER_PORT" not in os.environ:   # `--gpus 1 --multi` without a launcher
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    if args.one_device:
        local_rank = 0
    ndev = torch.cuda.device_count()
    if local_rank >= ndev:
        print(f"bench.py: rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible", file=sys.stderr, flush=True)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    with wd.phase("gloo rendezvous (torch.distributed.init_process_group) + ncclGetUniqueId broadcast", args.wait_s):
        # control plane only: id broadcast, barriers, timings
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=args.wait_s))
        box = [mrank.unique_id() if (rank == 0 and args.transport == "rccl") else None]
        dist.broadcast_object_list(box, src=0)
    coll = _Collective(world, dist)
    rc = 0
    try:
        multi_rank_body(args, rank, world, local_rank, coll, None, box[0], result, wd, dist=dist)
    except BaseException:  # noqa: BLE001 — a rank that failed must not leave the others in a collective for ever: say why, then end
        import traceback
        tb = traceback.format_exc()
        print(f"bench.py: rank {rank} failed:\n{tb}", file=sys.stderr, flush=True)
        if wd._secured is not None:    # an optional leg failed after the measurement: the measurement stands (Watchdog.secure)
            wd._leave(f"failed ({tb.strip().splitlines()[-1]})", 0)
        if rank == 0:
            emit(error_line(args, f"rank 0 failed: {tb.strip().splitlines()[-1]}"))
        os._exit(1)
    if rank == 0:
        emit(json.dumps(result["line"]))
        rc = _line_rc(result["line"])
    with wd.phase("shutdown barrier", args.wait_s):
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


def cpu_baseline(args, shp, rmax, gbed, il, of, jl):
    """The CPU baseline on a bounded sample of the same bed (the first rows of the same half list, OpenMP over rows):
    the build's own TUNED CPU implementation of docs/SPEC.md (bench/cpu_tuned.c: tabulated recurrence constants, SIMD
    over cap nodes) — the reference's PairSH is not in the mount, so `kind` is "port".  A slice of the sample is
    also run through the plain oracle (the checker) and compared.  The weighted rule only exists in the oracle."""
    import importlib.util
    from oracle import oracle as O  # checker / baseline only
    O.build()
    O.set_rule(args.rule)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nthreads = max(1, min(args.cpu_threads, avail))
    K = np.full((2, 2), 1000.0)
    E = np.full((2, 2), args.exponent)
    sh_list = [(args.lmax, a, r) for a, r in zip(shp, rmax)]
    tuned = None
    if args.rule == "sharp":
        spec = importlib.util.spec_from_file_location("cpu_tuned", os.path.join(ROOT, "bench", "cpu_tuned.py"))
        tuned = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(tuned)
        tuned.lib()

    def run(nrows, nt, plain=False):
        t = time.perf_counter()
        if tuned is not None and not plain:
            o = tuned.compute(sh_list, K, E, args.nq, gbed["x"].shape[0], gbed["x"], gbed["quat"], gbed["type"], gbed["shtype"],
                              il[:nrows], of[:nrows + 1], jl[:of[nrows]], nthreads=nt)
        else:
            o = O.compute(sh_list, K, E, args.nq, gbed["x"].shape[0], gbed["x"], gbed["quat"], gbed["type"],
                          gbed["shtype"], il[:nrows], of[:nrows + 1], jl[:of[nrows]], nthreads=nt)
        return time.perf_counter() - t, int(o["counts"][1]), o
    probe_rows = min(len(il), 4000)
    t_probe, c_probe, _ = run(probe_rows, nthreads)
    rate = c_probe / max(t_probe, 1e-6)
    per_row = max(c_probe / probe_rows, 1e-9)
    nrows = int(min(len(il), max(probe_rows, 0.7 * args.cpu_seconds * rate / per_row)))
    t_main, c_main, _ = run(nrows, nthreads)
    passes = 1
    if nrows == len(il):
        # the whole list is shorter than the budget (the tuned port does 100k particles in ~1 s on 16 threads): repeat the
        # pass until ~0.6 of the budget is spent, so that the figure is the mean over several seconds of CPU work
        while t_main < 0.6 * args.cpu_seconds and passes < 64:
            t2, c2, _ = run(nrows, nthreads)
            t_main += t2
            c_main += c2
            passes += 1
    rows1 = int(min(len(il), max(200, 0.15 * args.cpu_seconds * (rate / nthreads) / per_row)))
    t_one, c_one, _ = run(rows1, 1)
    out = {"value": c_main / t_main, "unit": "contact-pairs/s", "cores": nthreads, "kind": "port",
           "variant": "tuned (bench/cpu_tuned.c)" if tuned is not None else "plain oracle (the weighted rule has no tuned port)",
           "value_one_core": c_one / t_one,
           "sample": f"first {nrows} rows of the same half list x {passes} pass(es) ({c_main} contact pairs, {t_main:.1f} s, "
                     f"OpenMP x{nthreads}; one core: first {rows1} rows, {t_one:.1f} s); own CPU implementation of "
                     "docs/SPEC.md, not the reference's PairSH (absent from the mount)"}
    if tuned is not None:
        # the checker's word on the baseline, and what the plain oracle itself would have scored
        rows_c = min(len(il), 1500)
        _, _, ot = run(rows_c, nthreads)
        tp, cp, op = run(rows_c, nthreads, plain=True)
        fs = np.abs(op["f"]).max()
        out["max_rel_dev_from_oracle"] = float(max(np.abs(ot["f"] - op["f"]).max(), np.abs(ot["torque"] - op["torque"]).max()) / fs)
        out["plain_oracle_value"] = cp / tp
        assert out["max_rel_dev_from_oracle"] < 1e-11
    return out


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _rank_env(rank, world, port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK",
                                                            "LOCAL_WORLD_SIZE", "ROLE_RANK", "TORCHELASTIC_RUN_ID")}
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SHPAIR_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # the host driver only supports dmabuf IPC (RCCL between processes)
    return env


_WD_MSG = re.compile(r"^bench\.py: (rank \d+: '.*?' .*?); giving up with exit code \d+")


def run_rank_children(argv, world, bound_s, stdout_of_rank0=True, script=None, notes=None):
    """Starts `world` FRESH processes of this script — one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set —
    waits for all of them within `bound_s`, ends the others (their exact PIDs) when one dies or the bound passes, and
    returns (worst exit code, rank 0's stdout).  The caller has not touched the GPU: children are ordinary
    fork + exec of the interpreter, never a re-exec of a process that holds a HIP context.  The ranks' stderr is relayed
    line by line; what their watchdogs said before giving up is also appended to `notes` (a list)."""
    import subprocess
    port = _free_port()
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(world):
        procs.append(subprocess.Popen(cmd, env=_rank_env(r, world, port), cwd=ROOT, text=True,
                                      stdout=(subprocess.PIPE if r == 0 else subprocess.DEVNULL), stderr=subprocess.PIPE))
    out0 = []
    readers = []

    def _relay(p):
        for ln in p.stderr:
            sys.stderr.write(ln)
            sys.stderr.flush()
            m = _WD_MSG.match(ln)
            if m and notes is not None:
                notes.append(m.group(1))
    for p in procs:
        t = threading.Thread(target=_relay, args=(p,), daemon=True)
        t.start()
        readers.append(t)
    if stdout_of_rank0:
        def _read():
            for ln in procs[0].stdout:
                out0.append(ln)
        t = threading.Thread(target=_read, daemon=True)
        t.start()
        readers.append(t)
    deadline = time.monotonic() + bound_s
    codes = [None] * world
    worst = 0
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        failed = [r for r, c in enumerate(codes) if c not in (None, 0)]
        late = time.monotonic() > deadline
        if failed or late:
            # the ranks that are left would wait for the dead one until their own watchdog fires: give them a moment to
            # report by themselves, then end exactly the processes started here
            grace = time.monotonic() + (0.0 if late else 10.0)
            while time.monotonic() < grace and any(p.poll() is None for p in procs):
                time.sleep(0.2)
            ended_here = []
            for r, p in enumerate(procs):
                if p.poll() is None:
                    ended_here.append(r)
                    p.terminate()
            t_kill = time.monotonic() + 5.0
            for r, p in enumerate(procs):
                try:
                    p.wait(timeout=max(0.1, t_kill - time.monotonic()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            # the exit code reported is the worst among the ranks that ended BY THEMSELVES; the ones ended here only say so
            codes = [(0 if r in ended_here else p.returncode) for r, p in enumerate(procs)]
            if ended_here:
                print(f"bench.py: ended rank process(es) {ended_here} after " + ("the time bound" if late and not failed else
                      f"rank(s) {failed} failed"), file=sys.stderr, flush=True)
            if late and not failed:
                print(f"bench.py: the {world} rank processes did not finish within {bound_s:.0f} s", file=sys.stderr, flush=True)
                if notes is not None:
                    notes.append(f"the {world} rank processes did not finish within the launcher's bound of {bound_s:.0f} s")
                worst = 4
            elif notes is not None and failed:
                notes.append(f"rank(s) {failed} ended with exit code(s) {[codes[r] for r in failed]}")
            break
        time.sleep(0.2)
    for t in readers:
        t.join(timeout=10.0)
    for c in codes:
        if c is not None and c != 0:
            worst = max(worst, c if c > 0 else 128 - c)   # a signal's negative code as the shell would print it
    return worst, "".join(out0)


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process becomes the
    launcher.  It has made no GPU call (importing torch is all that happened) and makes none: it starts N fresh rank
    processes, relays rank 0's single JSON line and returns the worst exit code.  The line carries `scale_ref_cmd` —
    the command whose `value` is the like-for-like one-GPU point of the scaling curve (same workload per GPU, same C++
    loop, same transport code).  Bounded by --total-s from THIS process's start (the ranks get what is left, less a
    margin, as their own --total-s): when no measurement comes back in time the line printed here has `value` null and
    an `error` field naming the phase and the rank that gave up."""
    argv = [a for a in sys.argv[1:] if a != "--launch"]
    bound = 6.0 * args.wait_s + 120.0
    if args.total_s > 0:
        bound = max(20.0, args.total_s - (time.monotonic() - T_START) - 5.0)
        argv += ["--total-s", f"{max(10.0, bound - 15.0):.0f}"]
    notes = []
    rc, out = run_rank_children(argv, args.gpus, bound, notes=notes)
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    for ln in out.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        try:
            d = json.loads(lines[-1])
            d["launcher"] = "bench.py self-launch: N fresh rank processes (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* set by the parent)"
            if d.get("error") and notes:
                d["error_notes"] = notes
            print(json.dumps(d), flush=True)
        except ValueError:
            print(lines[-1], flush=True)
    else:
        if rc == 0:
            rc = 1
        # no measurement: one line all the same, `value` null and the reason (what the ranks' watchdogs said, or their exit codes)
        d = json.loads(error_line(args, "; ".join(notes) if notes else f"the rank processes ended with exit code {rc} and no line"))
        d["launcher"] = "bench.py self-launch"
        print(json.dumps(d), flush=True)
    sys.exit(rc)


if __name__ == "__main__":
    _args = parse()
    if _args.gpus == 1 and not _args.multi:
        main_single(_args)
    elif _args.transport in ("rccl", "staged") and "WORLD_SIZE" not in os.environ and (_args.gpus > 1 or _args.launch):
        self_launch(_args)
    else:
        main_multi(_args)
