"""Exercises shpair.halo.HaloExchange over real RCCL on a one-GPU box: world_size 1, the rank is its own
peer (NCCL allows send/recv to self inside a group), so the batched P2POp pattern, the direct receives
into the ghost rows and the index_add fold-in run through the product transport."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair.halo import HaloExchange  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
nlocal, ng = 5000, 700
rng = np.random.default_rng(0)
send = np.sort(rng.choice(nlocal, ng, replace=False))
view = dict(nlocal=nlocal, recv={0: (nlocal, nlocal + ng)}, send={0: send})
h = HaloExchange(view, dev, dist)
x = torch.randn(nlocal + ng, 3, dtype=torch.float64, device=dev)
q = torch.randn(nlocal + ng, 4, dtype=torch.float64, device=dev)
for _ in range(3):
    h.forward(x, q)
torch.cuda.synchronize()
assert torch.equal(x[nlocal:], x[torch.from_numpy(send).to(dev)]) and torch.equal(q[nlocal:], q[torch.from_numpy(send).to(dev)])
f = torch.randn(nlocal + ng, 3, dtype=torch.float64, device=dev)
t = torch.randn(nlocal + ng, 3, dtype=torch.float64, device=dev)
f0, t0 = f.clone(), t.clone()
h.reverse(f, t)
torch.cuda.synchronize()
fe, te = f0[:nlocal].clone(), t0[:nlocal].clone()
fe.index_add_(0, torch.from_numpy(send).to(dev), f0[nlocal:])
te.index_add_(0, torch.from_numpy(send).to(dev), t0[nlocal:])
assert torch.allclose(f[:nlocal], fe, atol=1e-14) and torch.allclose(t[:nlocal], te, atol=1e-14)
# the primitives of shpair/mrun.py (MultiRankRun) on RCCL: counts and variable-size rows by all_to_all_single,
# one-element int64 messages and row slices of int32 / float64 arrays in one batched point-to-point group
cnt = torch.tensor([37], dtype=torch.int64, device=dev)
got = torch.zeros_like(cnt)
dist.all_to_all_single(got, cnt)
assert got.item() == 37
rows = torch.randn(37, 16, dtype=torch.float64, device=dev)
back = torch.empty(37, 16, dtype=torch.float64, device=dev)
dist.all_to_all_single(back, rows, output_split_sizes=[37], input_split_sizes=[37])
assert torch.equal(back, rows)
ns = torch.tensor([5, 9], dtype=torch.int64, device=dev)
nr = torch.zeros_like(ns)
ti = torch.arange(40, dtype=torch.int32, device=dev)
ri = torch.zeros(40, dtype=torch.int32, device=dev)
ops = [dist.P2POp(dist.irecv, nr[0:1], 0), dist.P2POp(dist.irecv, nr[1:2], 0), dist.P2POp(dist.irecv, ri[10:25], 0),
       dist.P2POp(dist.isend, ns[0:1], 0), dist.P2POp(dist.isend, ns[1:2], 0), dist.P2POp(dist.isend, ti[3:18].contiguous(), 0)]
for w in dist.batch_isend_irecv(ops):
    w.wait()
torch.cuda.synchronize()
assert nr.tolist() == [5, 9] and torch.equal(ri[10:25], ti[3:18])
tot = torch.tensor([1.0, 2.0], dtype=torch.float64, device=dev)
dist.all_reduce(tot, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("RCCL self-peer halo exchange OK")
