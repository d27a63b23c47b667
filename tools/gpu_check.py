"""Quick GPU sanity run: HIP path vs oracle on a small bed, then timing on a big one.
Usage: python tools/gpu_check.py [n_small] [n_big] [lmax] [nq]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lammps-spherharm_amd"))
sys.path.insert(0, ROOT)
from shpair import ShPair, shapes, bed  # noqa: E402
from oracle import oracle as O  # noqa: E402  (checker only)


def run(n, lmax, nq, nshapes, expo, check, force_volume=0, eflag=False):
    shp = [shapes.random_shape(lmax, 100 + s) for s in range(nshapes)]
    sp = ShPair(0)
    sp.settings(nq)
    sp.set_ntypes(1, nshapes)
    for s, a in enumerate(shp):
        sp.set_shape(s, lmax, a)
    sp.coeff("*", "*", 1000.0, expo)
    rmax = [sp.rmax(s) for s in range(nshapes)]
    b = bed.make_bed(n, rmax, nshapes)
    il, of, jl = bed.half_neighbor_list(b["x"], b["shtype"], rmax)
    sp.set_neighbors_csr(il, of, jl)
    sp.set_option("timing", 1)
    sp.set_option("count", 1 if check else 0)
    sp.set_option("force_volume", force_volume)
    t0 = time.time()
    f, tq, eng, vir = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=eflag, vflag=eflag)
    t1 = time.time()
    st = sp.stats()
    ncontact = st["n_contact"] if check else jl.size
    fbuf, tbuf = np.zeros((n, 3)), np.zeros((n, 3))
    walls = []
    for _ in range(5):
        fbuf[:] = 0.0
        tbuf[:] = 0.0
        tw = time.perf_counter()
        f, tq, eng, vir = sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=eflag, vflag=eflag, f=fbuf, torque=tbuf)
        walls.append(1e3 * (time.perf_counter() - tw))
        st = sp.stats()
    if n >= 50000:
        # the same calls with the caller's arrays page-locked (shpair_pin_host): what PairSH does with LAMMPS' arrays
        arrs = [b["x"], b["quat"], b["type"], b["shtype"], fbuf, tbuf]
        for a_ in arrs:
            sp.pin_host(a_)
        pw = []
        for _ in range(7):
            fbuf[:] = 0.0
            tbuf[:] = 0.0
            tw = time.perf_counter()
            sp.compute(n, b["x"], b["quat"], b["type"], b["shtype"], eflag=eflag, vflag=eflag, f=fbuf, torque=tbuf)
            pw.append(1e3 * (time.perf_counter() - tw))
            stp = sp.stats()
        for a_ in arrs:
            sp.unpin_host(a_)
        assert np.array_equal(fbuf, f) or np.abs(fbuf - f).max() < 1e-12 * np.abs(f).max()
        print(f"   host-pointer call at n={n}: pageable wall {min(walls):.3f} ms, pinned wall {min(pw):.3f} ms, kernels "
              f"{stp['kernel_ms']:.3f} ms -> wall - kernel: pageable {min(walls) - st['kernel_ms']:.3f}, pinned "
              f"{min(pw) - stp['kernel_ms']:.3f} ms ({(b['x'].nbytes + b['quat'].nbytes + 2 * fbuf.nbytes + b['type'].nbytes * 2) / 1e6:.1f} MB up, "
              f"{2 * fbuf.nbytes / 1e6:.1f} MB down)", flush=True)
    print(f"n={n} L={lmax} nq={nq} nshapes={nshapes} expo={expo} fv={force_volume} e={eflag}: pairs={jl.size} "
          f"contact={ncontact} touching={st['n_touching']} kernel_ms={st['kernel_ms']:.3f} "
          f"total_ms={st['total_ms']:.3f} wall_ms={min(walls):.3f} first_call_s={t1 - t0:.2f} "
          f"contact_pairs/s={ncontact / (st['kernel_ms'] * 1e-3):.3e}", flush=True)
    if check:
        kn = np.full((2, 2), 1000.0)
        ex = np.full((2, 2), expo)
        o = O.compute([(lmax, a, r) for a, r in zip(shp, rmax)], kn, ex, nq, n, b["x"], b["quat"], b["type"],
                      b["shtype"], il, of, jl, eflag=eflag, vflag=eflag, force_volume=bool(force_volume),
                      nthreads=0)
        ef = np.abs(f - o["f"]).max() / np.abs(o["f"]).max()
        et = np.abs(tq - o["torque"]).max() / max(np.abs(o["torque"]).max(), np.abs(o["f"]).max())
        print(f"   oracle: counts={o['counts']} rel err f={ef:.2e} torque={et:.2e} "
              f"eng {eng:.12g} vs {o['eng_virial'][0]:.12g} vir err "
              f"{np.abs(vir - o['eng_virial'][1:]).max():.2e}", flush=True)
        assert o["counts"][1] == st["n_contact"], (o["counts"], st)
        assert ef < 1e-9 and et < 1e-9, (ef, et)
    sp.close()


if __name__ == "__main__":
    ns = int(sys.argv[1]) if len(sys.argv) > 1 else 500
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
    run(ns, 6, 16, 1, 1.0, True)
    run(ns, 6, 16, 1, 1.5, True, eflag=True)
    run(ns, 4, 10, 2, 1.0, True, force_volume=1)
    run(ns, 12, 32, 1, 1.25, True)
    run(ns, 14, 8, 1, 1.25, True)   # run-time-order kernel
    run(ns, 0, 8, 1, 1.0, True, force_volume=1)
    if nb > 0:
        run(nb, 6, 16, 1, 1.0, False)
        run(nb, 6, 16, 1, 1.0, False, force_volume=1)
        run(nb, 6, 16, 4, 1.0, False, force_volume=1)
        run(nb, 12, 32, 1, 1.0, False)
        run(nb, 12, 32, 1, 1.0, False, force_volume=1)
        run(nb, 4, 10, 1, 1.0, False, force_volume=1)
