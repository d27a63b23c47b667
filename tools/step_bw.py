"""Achieved HBM bandwidth of the Part II streaming kernels vs particle count (diagnostic, GPU).
Algorithmic bytes per particle: nve<0> 284, nve<1> 184, post_force (gamma_r != 0) 164+48."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import ShPair, shapes  # noqa: E402

sp = ShPair(0)
sp.settings(8)
sp.set_ntypes(1, 2)
for s in range(2):
    sp.set_shape(s, 6, shapes.random_shape(6, 10 + s, amp=0.2))
sp.coeff(1, 1, 1000.0, 1.0)
dev = "cuda:0"
for n in (100_000, 1_000_000, 8_000_000):
    g = torch.Generator(device=dev).manual_seed(1)
    x, v, L, f, t = (torch.randn(n, 3, dtype=torch.float64, device=dev, generator=g) for _ in range(5))
    q = torch.randn(n, 4, dtype=torch.float64, device=dev, generator=g)
    q /= q.norm(dim=1, keepdim=True)
    sh = torch.randint(0, 2, (n,), dtype=torch.int32, device=dev, generator=g)
    mask = torch.ones(n, dtype=torch.int32, device=dev)
    a = [x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), f.data_ptr(), t.data_ptr(), sh.data_ptr(), mask.data_ptr()]

    def timed(fn, reps=20):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e-3
    t0 = timed(lambda: sp.nve_device(0, n, 1e-6, *a))
    t1 = timed(lambda: sp.nve_device(1, n, 1e-6, *a))
    tp = timed(lambda: sp.post_force_device(n, [0, 0, -1.0], 0.1, 0.1, v.data_ptr(), q.data_ptr(), L.data_ptr(), sh.data_ptr(),
                                            mask.data_ptr(), f.data_ptr(), t.data_ptr()))
    tc = timed(lambda: x.clone())   # 24 B read + 24 B written per particle: the box's copy rate at this size
    print(f"n {n:>8}: nve<0> {t0 * 1e6:8.1f} us {284 * n / t0 / 1e9:7.0f} GB/s | nve<1> {t1 * 1e6:8.1f} us {184 * n / t1 / 1e9:7.0f} GB/s | "
          f"post_force {tp * 1e6:8.1f} us {212 * n / tp / 1e9:7.0f} GB/s | torch copy {48 * n / tc / 1e9:7.0f} GB/s", flush=True)
    del x, v, L, f, t, q, sh, mask
sp.close()
