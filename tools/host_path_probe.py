"""Where does the host-pointer call overhead of bench.py's `host_path` leg (0.45 ms) come from when tools/gpu_check.py
measures 0.32 ms for the same call?  The same leg (A) before this process has touched torch, (B) after torch has
initialised the device, (C) after the headline's StaticBed has run (the state bench.py is in)."""
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from shpair import shapes, bed  # noqa: E402

args = bench.parse()
shp = [shapes.random_shape(args.lmax, bed.SEED0 + 2)]
sp0 = bench.make_ctx(args, shp, 0)
rmax = [sp0.rmax(0)]
sp0.close()
g = bed.make_bed(args.particles, rmax, 1, seed=bed.SEED0 + 2)
il, of, jl = bed.half_neighbor_list(g["x"], g["shtype"], rmax)
fake = types.SimpleNamespace(gbed=g, nlocal=args.particles, shp=shp, il=il, of=of, jl=jl)


def show(tag):
    h = bench.host_path_leg(args, fake, 0.0)
    print(f"{tag}: call pinned {h['compute_call_ms_pinned']:.3f} (median {h['compute_call_ms_pinned_median']:.3f}) kernels {h['compute_kernel_ms']:.3f} "
          f"overhead pinned {h['compute_overhead_ms_pinned']:.3f} pageable {h['compute_overhead_ms_pageable']:.3f}; set_neighbors {h['set_neighbors_ms']:.3f}", flush=True)


show("A no torch yet      ")
import torch  # noqa: E402
torch.cuda.set_device(0)
t = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
show("B torch initialised ")
sb = bench.StaticBed(args, 100)
sb.count()
sb.timed(8, 20)
show("C after StaticBed   ")
wd = bench.Watchdog(args, emit=False)
show("D + watchdog thread ")
fake.sp = sb.sp     # what bench.py does: the leg on the headline's own context (the process's first streams)
show("E on the first ctx  ")
