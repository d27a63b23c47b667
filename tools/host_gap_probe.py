"""Why does the host-pointer entry point (shpair_compute) report ~0.25 ms more KERNEL time per call than the
device-resident loop at the headline (bench.py `host_path.compute_kernel_ms` vs `roofline.kernel_ms`)?  Probe: the
device-resident call with the same hipEvent timing, (a) back to back, (b) with a stream synchronisation and an idle gap
of g ms between calls, as a host that works between calls leaves — if (b) reproduces the difference, it is the GPU's
clock / power state after an idle gap and not something the staging code does.

  python tools/host_gap_probe.py [gap_ms ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
gaps = [float(a) for a in sys.argv[1:]] or [0.0, 0.3, 0.7, 2.0]
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
import torch  # noqa: E402

args = bench.parse()
torch.cuda.set_device(0)
sb = bench.StaticBed(args, 400)
sb.count()
sb.timed(12, 20)
sp = sb.sp
sp.set_option("timing", 1)


def call():
    sb.f.zero_()
    sb.tq.zero_()
    sp.compute_device(sb.nlocal, 0, sb.x.data_ptr(), sb.q.data_ptr(), sb.ty.data_ptr(), sb.sh.data_ptr(), sb.f.data_ptr(), sb.tq.data_ptr(),
                      stream=sb.stream.cuda_stream)


for g in gaps:
    ks = []
    for _ in range(25):
        call()
        torch.cuda.synchronize()
        ks.append(sp.stats()["kernel_ms"])
        if g > 0:
            t = time.perf_counter()
            while 1e3 * (time.perf_counter() - t) < g:
                pass
    print(f"idle gap {g:4.1f} ms between synchronised calls: kernel_ms median {np.median(ks[5:]):.3f}  min {min(ks[5:]):.3f}  max {max(ks[5:]):.3f}", flush=True)
# back to back without any synchronisation (what the bench's timed loop does)
el, kms = sb.timed(4, 20)
print(f"back to back, no synchronisation between steps: kernel_ms mean {kms:.3f}", flush=True)
