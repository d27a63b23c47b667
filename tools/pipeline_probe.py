"""Would the HBM-bound set-up and rotation kernels hide beside the FP64-bound contact kernel of another slot range?
The pair path over chunks of the slot list on two streams (shp_compute_range, the entry the halo loop uses), against
the whole list on one stream.   python tools/pipeline_probe.py [lmax nq [chunks ...]]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
import torch  # noqa: E402
from shpair import capi, shapes, bed, ShPair  # noqa: E402

lmax = int(sys.argv[1]) if len(sys.argv) > 1 else 6
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 16
chunk_list = [int(a) for a in sys.argv[3:]] or [1, 2, 4, 8]
n = 100000
a = shapes.random_shape(lmax, bed.SEED0 + 2)
sp = ShPair(0)
sp.settings(nq)
sp.set_ntypes(1, 1)
sp.set_shape(0, lmax, a)
sp.coeff("*", "*", 1000.0, 1.25)
rmax = [sp.rmax(0)]
b = bed.make_bed(n, rmax, seed=bed.SEED0 + 2)
il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
sp.set_neighbors_csr(il, of, jl)
npairs = int(jl.size)
dev = torch.device("cuda", 0)
x = torch.from_numpy(b["x"]).to(dev)
q = torch.from_numpy(b["quat"]).to(dev)
ty = torch.from_numpy(b["type"]).to(dev)
sh = torch.from_numpy(b["shtype"]).to(dev)
f = torch.zeros(n, 3, dtype=torch.float64, device=dev)
tq = torch.zeros_like(f)
ev = torch.zeros(8, dtype=torch.float64, device=dev)
lib = capi.load_library()
fn = lib.shp_compute_range
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_void_p] * 3 + [C.c_void_p, C.c_int, C.c_int, C.c_int]
s0 = torch.cuda.Stream()
s1 = torch.cuda.Stream()


def run(chunks):
    """One pair compute: `chunks` slot ranges, alternating between the two streams (pre on the first, post on the last)."""
    bounds = [((npairs * k // chunks) // 64) * 64 for k in range(chunks)] + [npairs]
    e0 = torch.cuda.Event()
    e0.record(s0)
    s1.wait_event(e0)
    for k in range(chunks):
        st = s0 if (k % 2 == 0 or chunks == 1) else s1
        part = (1 if k == 0 else 0)
        if k == chunks - 1:
            # the last range carries the post part: it must follow everything
            if chunks > 1:
                other = s1 if st is s0 else s0
                ej = torch.cuda.Event()
                ej.record(other)
                st.wait_event(ej)
            part |= 2
        rc = fn(sp._h, n, 0, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), 1, 0, 0, f.data_ptr(), tq.data_ptr(),
                ev.data_ptr(), C.c_void_p(st.cuda_stream), bounds[k], bounds[k + 1], part)
        assert rc == 0, rc
    last = s0 if ((chunks - 1) % 2 == 0 or chunks == 1) else s1
    if last is not s0:
        ej = torch.cuda.Event()
        ej.record(s1)
        s0.wait_event(ej)


def timed(chunks, reps=10):
    f.zero_(); tq.zero_()
    torch.cuda.synchronize()
    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a_.record(s0)
    for _ in range(reps):
        run(chunks)
    b_.record(s0)
    torch.cuda.synchronize()
    return a_.elapsed_time(b_) / reps


for c in chunk_list:
    run(c)
torch.cuda.synchronize()
ref = None
for c in chunk_list:
    f.zero_(); tq.zero_()
    run(c)
    torch.cuda.synchronize()
    if ref is None:
        ref = f.clone()
    print(f"chunks {c}: max |f - f(1 chunk)| / max |f| = {float((f - ref).abs().max() / ref.abs().max()):.2e}")
for rnd in range(4):
    print("round", rnd, "  ".join(f"{c} chunk(s): {timed(c):.3f} ms" for c in chunk_list))
