"""Energy drift of the device-resident NVE loop on a periodic bed vs dt and n_q (diagnostic, GPU)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
from shpair import ShPair, shapes  # noqa: E402


def run(m, jitter, nq, dt, nsteps, lmax=4, kn=200.0, expo=1.5, skin=0.3, seed=63, rule=0):
    rng = np.random.default_rng(seed)
    shp = shapes.random_shape(lmax, 50, amp=0.2)
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, 1)
    sp.set_shape(0, lmax, shp)
    sp.coeff(1, 1, kn, expo)
    sp.set_option("rule", rule)
    box = np.array([m, m, m]) * 1.9
    g = np.stack(np.meshgrid(*[np.arange(m)] * 3, indexing="ij"), -1).reshape(-1, 3)
    x0 = (g + 0.5) * 1.9 + rng.uniform(-jitter, jitter, (g.shape[0], 3))
    n = x0.shape[0]
    q0 = rng.normal(size=(n, 4))
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    sp.set_box([0, 0, 0], box, (1, 1, 1), skin)
    nmax = 4 * n
    dev = "cuda:0"
    x = torch.zeros(nmax, 3, dtype=torch.float64, device=dev)
    q = torch.zeros(nmax, 4, dtype=torch.float64, device=dev)
    ty = torch.ones(nmax, dtype=torch.int32, device=dev)
    sh = torch.zeros(nmax, dtype=torch.int32, device=dev)
    x[:n] = torch.from_numpy(x0).to(dev)
    q[:n] = torch.from_numpy(q0).to(dev)
    v = torch.zeros(n, 3, dtype=torch.float64, device=dev)
    L = torch.zeros_like(v)
    mask = torch.ones(n, dtype=torch.int32, device=dev)
    f = torch.zeros(nmax, 3, dtype=torch.float64, device=dev)
    tq = torch.zeros_like(f)
    ev = torch.zeros(7, dtype=torch.float64, device=dev)
    ke = torch.zeros(3, dtype=torch.float64, device=dev)
    g0 = np.zeros(3)
    st = dict(ng=0, builds=0)

    def rebuild():
        st["ng"] = sp.borders_device(n, nmax, x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr())
        sp.neighbor_build_device(n, st["ng"], x.data_ptr(), sh.data_ptr())
        st["builds"] += 1

    def force():
        f.zero_(); tq.zero_(); ev.zero_()
        sp.forward_device(x.data_ptr(), q.data_ptr())
        sp.compute_device(n, st["ng"], x.data_ptr(), q.data_ptr(), ty.data_ptr(), sh.data_ptr(), f.data_ptr(), tq.data_ptr(),
                          eflag=True, ev=ev.data_ptr())
        sp.reverse_device(f.data_ptr(), tq.data_ptr())

    def energy():
        ke.zero_()
        sp.energies_device(n, g0, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), sh.data_ptr(), mask.data_ptr(), ke.data_ptr())
        torch.cuda.synchronize()
        return ev[0].item(), ke[0].item(), ke[1].item()
    rebuild()
    force()
    e0 = energy()
    hist = []
    for step in range(nsteps):
        sp.nve_device(0, n, dt, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), f.data_ptr(), tq.data_ptr(), sh.data_ptr(), mask.data_ptr())
        if sp.neighbor_check_device(n, x.data_ptr()):
            rebuild()
        force()
        sp.nve_device(1, n, dt, x.data_ptr(), v.data_ptr(), q.data_ptr(), L.data_ptr(), f.data_ptr(), tq.data_ptr(), sh.data_ptr(), mask.data_ptr())
        if (step + 1) % max(1, nsteps // 5) == 0:
            hist.append(sum(energy()))
    e1 = energy()
    sp.close()
    return e0, e1, hist, st["builds"]


if __name__ == "__main__":
    T = 0.3
    rules = (0, 1) if len(sys.argv) > 1 and sys.argv[1] == "rules" else (0,)
    for rule in rules:
      for jitter in ((0.1,) if len(rules) > 1 else (0.3, 0.1)):
        for nq in ((8, 12) if len(rules) > 1 else (8, 12, 24)):
            for dt in ((1e-3, 5e-4) if len(rules) > 1 else (2e-3, 1e-3, 5e-4)):
                e0, e1, hist, nb = run(8, jitter, nq, dt, int(round(T / dt)), rule=rule)
                tot0, tot1 = sum(e0), sum(e1)
                print(f"rule {rule} jitter {jitter} nq {nq:2d} dt {dt:.0e}: E0 {tot0:.4f} (pe) -> pe {e1[0]:.2f} ket {e1[1]:.2f} ker {e1[2]:.2f}; "
                      f"drift {(tot1 - tot0) / tot0:+.3e}  builds {nb}  hist {[f'{(h - tot0) / tot0:+.1e}' for h in hist]}", flush=True)
