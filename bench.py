#!/usr/bin/env python3
"""bench.py — contact-pairs/s of the `pair_style sh` hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one synthetic bed resident in HBM,
inside the two integrator half-steps that surround it in a timestep:
initial_integrate -> [forward halo exchange, N > 1] -> clear f/torque ->
shpair_compute_device() -> [reverse halo exchange, N > 1] -> final_integrate.  Workload at N = 1: BASELINE.json
configs[1] — 100k particles, one L_max = 6 shape, dense packed bed, n_q = 16
(Q = 512 cap nodes per pair), general force law (exponent 1.25, so the overlap
volume root finder runs for every touching node).  N > 1: the same bed per
rank (weak scaling), bricks of the processor grid, RCCL point-to-point halo.

Prints ONE JSON line on rank 0 (fields: module docstring of DESIGN.md §Measurement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

BYTES_PER_PAIR = 236          # algorithmic HBM bytes per contact pair
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6   # 256 CU x 4 SIMD x 16 lanes/clk x 2 flop x 2.4 GHz (= half the FP32 vector peak)


def flops_per_pair(lmax, nq):
    """SURVEY.md §8(d) algorithmic count: (60 + 6L + 9T) FLOP per cap node, Q = 2 nq^2 nodes."""
    T = (lmax + 1) * (lmax + 2) // 2
    return (60 + 6 * lmax + 9 * T) * 2 * nq * nq


def pmc_traffic(args):
    """HBM traffic per launch measured with rocprofv3 PMC passes for this exact workload, or None."""
    try:
        tab = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        key = f"{args.particles}:{args.lmax}:{args.nq}:{args.nshapes}:{args.exponent:g}"
        return tab[key]["traffic_bytes"] if key in tab else None
    except (OSError, ValueError, KeyError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--particles", type=int, default=100000, help="particles per GPU")
    ap.add_argument("--lmax", type=int, default=6)
    ap.add_argument("--nq", type=int, default=16)
    ap.add_argument("--nshapes", type=int, default=1)
    ap.add_argument("--exponent", type=float, default=1.25)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product path). gloo = rehearsal only: every rank uses "
                         "GPU 0 and the halo buffers are staged through the host")
    ap.add_argument("--verify", action="store_true",
                    help="N > 1: gather the owned forces and compare them with a single-domain compute of the "
                         "whole bed on rank 0 (small beds only)")
    ap.add_argument("--ramp", type=int, default=8, help="extra untimed passes before the W warm-up steps: the first "
                    "~8 launches of a fresh process run up to 25 %% slower while the GPU clock ramps (rocprof per-launch "
                    "durations in profiles/); they are never part of the K timed steps")
    ap.add_argument("--rule", default="sharp", choices=["sharp", "weighted"],
                    help="cap rule: sharp inside test (docs/SPEC.md §2.5, the headline) or covered-fraction weights (§2.8)")
    ap.add_argument("--ts-steps", type=int, default=40, help="steps of the whole-timestep leg (N = 1 only; 0 = skip)")
    ap.add_argument("--multi-ts-steps", type=int, default=0,
                    help="N > 1 only, off by default: whole timesteps with atom migration and rebuilds through "
                         "shpair.mrun.MultiRankRun on a periodic bed of `particles` per rank (weak scaling)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the cpu_baseline leg "
                    "(16 = the host-core share of one GPU on the bench box)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    from shpair import ShPair, shapes, bed
    from shpair.halo import Decomposition, HaloExchange, proc_grid

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    rehearsal = args.backend == "gloo"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    lmax, nq, nshapes = args.lmax, args.nq, args.nshapes
    shp = [shapes.random_shape(lmax, bed.SEED0 + 2 + s) for s in range(nshapes)]
    sp = ShPair(local_rank)
    sp.settings(nq)
    sp.set_ntypes(1, nshapes)
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
    sp.coeff("*", "*", 1000.0, args.exponent)
    sp.set_option("rule", 1 if args.rule == "weighted" else 0)
    rmax = [sp.rmax(s) for s in range(nshapes)]

    # ---- the bed: world x particles, bricks of the processor grid (weak scaling)
    grid = proc_grid(world)
    gbed = bed.make_bed(args.particles * world, rmax, nshapes, seed=bed.SEED0 + 2,
                        aspect=tuple(float(g) for g in grid))
    halo = None
    if world == 1:
        gid = np.arange(args.particles)
        nlocal = args.particles
        il, of, jl = bed.half_neighbor_list(gbed["x"], gbed["shtype"], rmax)
    else:
        dec = Decomposition(gbed["x"], gbed["shtype"], rmax, grid)
        view = dec.plan(rank)
        gid, nlocal = view["gid"], view["nlocal"]
        il, of, jl = dec.neighbor_list(view)
        halo = HaloExchange(view, dev, dist, host_staged=rehearsal)
    nall = gid.size
    sp.set_neighbors_csr(il, of, jl)

    x = torch.from_numpy(gbed["x"][gid]).to(dev)
    q = torch.from_numpy(gbed["quat"][gid]).to(dev)
    ty = torch.from_numpy(gbed["type"][gid]).to(dev)
    sh = torch.from_numpy(gbed["shtype"][gid]).to(dev)
    f = torch.zeros(nall, 3, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(f)
    stream = torch.cuda.current_stream()
    # the integrator either side of the hot path (include/shstep.h): owned rows only.  The bed starts at
    # rest and dt is sized so that nothing moves further than 1e-2 of the neighbour skin during the whole
    # run: the half list (and the contact-pair count) stay valid without a rebuild.  --verify keeps dt = 0.
    v = torch.zeros(nlocal, 3, dtype=torch.float64, device=dev)
    angmom = torch.zeros_like(v)
    mask = torch.ones(nlocal, dtype=torch.int32, device=dev)
    dt = 0.0 if args.verify else min(1.0e-4, 4.0e-3 / (args.steps + args.warmup + args.ramp + 1))

    def integrate(phase):
        sp.nve_device(phase, nlocal, dt, x.data_ptr(), v.data_ptr(), q.data_ptr(), angmom.data_ptr(), f.data_ptr(),
                      tq.data_ptr(), sh.data_ptr(), mask.data_ptr(), stream=stream.cuda_stream)

    def step():
        integrate(0)
        if halo is not None:
            halo.forward(x, q)
        f.zero_()
        tq.zero_()
        sp.compute_device(nlocal, nall - nlocal, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(),
                          f.data_ptr(), tq.data_ptr(), stream=stream.cuda_stream)
        if halo is not None:
            halo.reverse(f, tq)
        integrate(1)

    # ---- untimed: count the contact pairs of this bed (static positions)
    sp.set_option("count", 1)
    step()
    torch.cuda.synchronize()
    st = sp.stats()
    n_contact, n_touching = st["n_contact"], st["n_touching"]
    sp.set_option("count", 0)

    for _ in range(args.ramp + args.warmup):
        step()

    # ---- timed region: exactly K steps between barrier + synchronize
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        integrate(0)
        if halo is not None:
            halo.forward(x, q)
        f.zero_()
        tq.zero_()
        ev[k][0].record(stream)   # HIP events on the stream the pair kernel is launched on
        sp.compute_device(nlocal, nall - nlocal, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(),
                          f.data_ptr(), tq.data_ptr(), stream=stream.cuda_stream)
        ev[k][1].record(stream)
        if halo is not None:
            halo.reverse(f, tq)
        integrate(1)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    tot = torch.tensor([elapsed, float(n_contact), float(n_touching), kernel_ms], dtype=torch.float64,
                       device="cpu" if rehearsal else dev)
    if dist is not None:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tot.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        contact_all, touching_all = float(sm[1]), float(sm[2])
    else:
        contact_all, touching_all = float(n_contact), float(n_touching)

    # sanity: forces are finite and (N = 1) sum to zero
    fh = f[:nlocal].cpu().numpy()
    assert np.all(np.isfinite(fh)) and np.abs(fh).max() > 0

    verify_err = None
    if args.verify and dist is not None:
        mine = (gid[:nlocal], f[:nlocal].cpu().numpy(), tq[:nlocal].cpu().numpy())
        parts = [None] * world if rank == 0 else None
        dist.gather_object(mine, parts, dst=0)
        if rank == 0:
            ntot = args.particles * world
            fg = np.zeros((ntot, 3))
            tg = np.zeros((ntot, 3))
            for g_, f_, t_ in parts:
                fg[g_] = f_
                tg[g_] = t_
            ref = ShPair(local_rank)
            ref.settings(nq)
            ref.set_ntypes(1, nshapes)
            for s_, a_ in enumerate(shp):
                ref.set_shape(s_, lmax, a_)
            ref.coeff("*", "*", 1000.0, args.exponent)
            ril, rof, rjl = bed.half_neighbor_list(gbed["x"], gbed["shtype"], rmax)
            ref.set_neighbors_csr(ril, rof, rjl)
            fr_, tr_, _, _ = ref.compute(ntot, gbed["x"], gbed["quat"], gbed["type"], gbed["shtype"])
            ref.close()
            sc = np.abs(fr_).max()
            verify_err = float(max(np.abs(fg - fr_).max(), np.abs(tg - tr_).max()) / sc)
            assert verify_err < 1e-9, f"decomposed forces differ from single-domain forces: {verify_err}"

    mts = None
    if world > 1 and args.multi_ts_steps > 0:
        mts = multi_rank_timestep_leg(args, shp, local_rank, dist, rank, world, grid, rehearsal)
    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = contact_all * args.steps / elapsed
        achieved_gbs = BYTES_PER_PAIR * n_contact / (kernel_ms * 1e-3) / 1e9
        fpp = flops_per_pair(lmax, nq)
        achieved_tf = fpp * n_contact / (kernel_ms * 1e-3) / 1e12
        out = {
            "metric": "contact_pairs_per_sec", "value": value, "unit": "contact-pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ramp_passes": args.ramp,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.particles} particles/GPU, {nshapes} SH shape(s) L_max={lmax}, dense packed "
                            f"bed (jittered HCP, spacing 1.9 mean radii), n_q={nq} (Q={2 * nq * nq} nodes/pair), "
                            f"pair_coeff kn=1000 exponent={args.exponent} (overlap volume + force + torque), "
                            "inputs resident in HBM",
                "particles_per_gpu": args.particles, "lmax": lmax, "nq": nq, "nshapes": nshapes,
                "exponent": args.exponent, "rule": args.rule, "proc_grid": list(grid),
                "backend": "rccl" if (world > 1 and not rehearsal) else ("gloo-rehearsal" if world > 1 else "none"),
                "half_list_pairs_rank0": int(jl.size), "contact_pairs_rank0": int(n_contact),
                "touching_pairs_rank0": int(n_touching), "contact_pairs_all_ranks": int(contact_all),
                "ghost_atoms_rank0": int(nall - nlocal),
            },
            "timesteps_per_sec": args.steps / elapsed,
            "timestep_note": "one step = initial_integrate + [forward halo] + clear + pair compute + [reverse halo] + "
                             f"final_integrate on every rank, dt = {dt:g} from rest, no list rebuild inside the timed "
                             "steps (see the `timestep` object for whole steps with rebuilds at N = 1)",
            "verify_rel_err": verify_err,
            "roofline": {
                "bound": "hbm", "achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_gbs / HBM_PEAK_GBS,
                "traffic": pmc_traffic(args) if world == 1 else None,
                "kernel": "pair_contact_kernel", "kernel_ms": kernel_ms,
                "bytes_per_pair": BYTES_PER_PAIR, "pairs_per_launch": int(n_contact),
                "algorithmic_bytes_per_launch": BYTES_PER_PAIR * int(n_contact),
                "note": "north_star asks for the HBM fraction; the kernel is FP64-VALU bound (see valu_f64). "
                        "traffic: bytes/launch from profiles/pmc_traffic.json (rocprofv3 FETCH_SIZE + WRITE_SIZE "
                        "passes of this workload; uncalibrated access widths, see the file)",
            },
            "occupancy": dict(sp.kernel_info(), note="static footprint of pair_contact_kernel as launched: one wave = one pair "
                              "= one workgroup; waves_per_cu = min(4 x VGPR limit, LDS limit) of a gfx950 CU"),
            "valu_f64": {
                "bound": "valu_f64", "achieved": achieved_tf, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / F64_VALU_PEAK_TFLOPS, "flop_per_pair": fpp,
                "note": "algorithmic FLOP (SURVEY §8d formula) / kernel time; no MFMA by design",
            },
        }
        if world == 1 and args.ts_steps > 0:
            out["timestep"] = timestep_leg(args, shp, local_rank)
        if mts is not None:
            out["timestep_multi_rank"] = mts
        if world == 1 and args.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(args, shp, rmax, gbed, il, of, jl)
        print(json.dumps(out), flush=True)
    sp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def timestep_leg(args, shp, device):
    """Whole device-resident timesteps per second (the second half of BASELINE.json's metric): NVE run of
    a fully periodic dense bed of the same shapes — integrate, rebuild test, ghosts, pair forces, reverse,
    integrate — everything through the C ABI (include/shstep.h), nothing on the host but launches."""
    import torch
    from shpair import ShPair, bed
    from shpair.run import DeviceRun
    sp = ShPair(device)
    sp.settings(args.nq)
    sp.set_ntypes(1, args.nshapes)
    for s, a in enumerate(shp):
        sp.set_shape(s, args.lmax, a)
    sp.coeff("*", "*", 1000.0, args.exponent)
    sp.set_option("rule", 1 if args.rule == "weighted" else 0)
    pts, lo, hi = bed.periodic_hcp(args.particles, 1.9, (1, 1, 1))
    rng = np.random.default_rng(bed.SEED0 + 7)
    n = pts.shape[0]
    pts = pts + rng.uniform(-0.04, 0.04, pts.shape)
    quat = bed.random_quaternions(n, rng)
    shtype = rng.integers(0, args.nshapes, n).astype(np.int32) if args.nshapes > 1 else np.zeros(n, np.int32)
    skin, dt = 0.1, 1.0e-3
    run = DeviceRun(sp, pts, quat, shtype, lo, hi, (1, 1, 1), skin, dt=dt, device=f"cuda:{device}")
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
    # cost of one rebuild (borders + bins + half list), timed on its own
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


def multi_rank_timestep_leg(args, shp, device, dist, rank, world, grid, rehearsal):
    """Whole timesteps on N ranks with everything LAMMPS does around the pair style when atoms move: migration,
    ghosts, list rebuilds (shpair.mrun.MultiRankRun).  Periodic bed of `particles` per rank; every rank calls this."""
    import torch
    from shpair import ShPair, bed
    from shpair.mrun import MultiRankRun
    sp = ShPair(device)
    sp.settings(args.nq)
    sp.set_ntypes(1, args.nshapes)
    for s, a in enumerate(shp):
        sp.set_shape(s, args.lmax, a)
    sp.coeff("*", "*", 1000.0, args.exponent)
    sp.set_option("rule", 1 if args.rule == "weighted" else 0)
    pts, lo, hi = bed.periodic_hcp(args.particles * world, 1.9, (1, 1, 1))
    rng = np.random.default_rng(bed.SEED0 + 7)
    n = pts.shape[0]
    pts = pts + rng.uniform(-0.04, 0.04, pts.shape)
    quat = bed.random_quaternions(n, rng)
    shtype = rng.integers(0, args.nshapes, n).astype(np.int32) if args.nshapes > 1 else np.zeros(n, np.int32)
    blen = (hi - lo) / np.array(grid)
    c = np.minimum(((np.mod(pts - lo, hi - lo)) / blen).astype(int), np.array(grid) - 1)
    mine = ((c[:, 0] * grid[1] + c[:, 1]) * grid[2] + c[:, 2]) == rank
    run = MultiRankRun(sp, dist, rank, world, grid, lo, hi, (1, 1, 1), 0.1, pts[mine], quat[mine], shtype[mine],
                       np.arange(n, dtype=np.int32)[mine], dt=1.0e-3, device=f"cuda:{device}", staged=rehearsal)
    run.run(5)
    b0 = run.builds
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    run.run(args.multi_ts_steps)
    torch.cuda.synchronize()
    dist.barrier()
    el = time.perf_counter() - t0
    tot = torch.tensor([el, float(run.n), float(run.nghost), float(run.migrated)], dtype=torch.float64,
                       device="cpu" if rehearsal else f"cuda:{device}")
    mx = tot.clone()
    dist.all_reduce(mx, op=dist.ReduceOp.MAX)
    sm = tot.clone()
    dist.all_reduce(sm, op=dist.ReduceOp.SUM)
    out = {"timesteps_per_s": args.multi_ts_steps / float(mx[0]), "ms_per_step": 1e3 * float(mx[0]) / args.multi_ts_steps,
           "steps": args.multi_ts_steps, "particles_all_ranks": int(sm[1]), "ghosts_all_ranks": int(sm[2]),
           "migrated_atoms": int(sm[3]), "rebuilds": run.builds - b0, "proc_grid": list(grid), "dt": 1.0e-3, "skin": 0.1,
           "what": "initial_integrate + rebuild test (all-reduce) + [migration, ghosts, list build] + forward + pair compute "
                   "+ reverse + final_integrate on every rank (shpair.mrun.MultiRankRun)"}
    sp.close()
    return out


def cpu_baseline(args, shp, rmax, gbed, il, of, jl):
    """The CPU oracle (a port of docs/SPEC.md — the reference's PairSH is not in the mount) on a bounded
    sample of the same bed: the first rows of the same half list, all host cores via OpenMP."""
    from oracle import oracle as O  # checker / baseline only
    O.build()
    O.set_rule(args.rule)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nthreads = max(1, min(args.cpu_threads, avail))
    K = np.full((2, 2), 1000.0)
    E = np.full((2, 2), args.exponent)
    sh_list = [(args.lmax, a, r) for a, r in zip(shp, rmax)]

    def run(nrows, nt):
        t = time.perf_counter()
        o = O.compute(sh_list, K, E, args.nq, gbed["x"].shape[0], gbed["x"], gbed["quat"], gbed["type"],
                      gbed["shtype"], il[:nrows], of[:nrows + 1], jl[:of[nrows]], nthreads=nt)
        return time.perf_counter() - t, int(o["counts"][1])
    probe_rows = min(len(il), 2000)
    t_probe, c_probe = run(probe_rows, nthreads)
    rate = c_probe / max(t_probe, 1e-6)
    per_row = max(c_probe / probe_rows, 1e-9)
    nrows = int(min(len(il), max(probe_rows, args.cpu_seconds * rate / per_row)))
    t_main, c_main = run(nrows, nthreads)
    # one thread, on a sample sized for about a fifth of the budget (a plain LAMMPS rank is one core)
    rows1 = int(min(len(il), max(200, 0.2 * args.cpu_seconds * (rate / nthreads) / per_row)))
    t_one, c_one = run(rows1, 1)
    return {"value": c_main / t_main, "unit": "contact-pairs/s", "cores": nthreads, "kind": "port",
            "value_one_core": c_one / t_one,
            "sample": f"first {nrows} rows of the same half list ({c_main} contact pairs, {t_main:.1f} s, "
                      f"OpenMP x{nthreads}; one core: first {rows1} rows, {t_one:.1f} s); own CPU restatement of "
                      "docs/SPEC.md, not the reference's PairSH (absent from the mount)"}


if __name__ == "__main__":
    main()
