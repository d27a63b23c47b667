"""Builds tests/golden/settled_cfg1_L4.npz from the state tools/settle.py wrote on the GPU box.

BASELINE.json configs[0]: "1000 identical L_max=4 ellipsoid-like SH particles, gravity-settled packed
bed".  The INPUTS (positions, orientations after the settle run, box, frozen floor) come from the
device-resident loop; every EXPECTED number in the fixture (ghosts, half list, forces, torques, energy)
is computed here by the CPU oracle alone.  NOT reference vectors (the reference mount has no code).

  python tests/golden/make_settled.py gpurun_out/settled_cfg1.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SKIN = 0.3


def expected(g):
    lmax, nq = int(g["lmax"]), int(g["nq"])
    anm = g["anm"]
    rmax = np.array([O.shape_rmax(lmax, a) for a in anm])
    n = g["x"].shape[0]
    x = np.ascontiguousarray(g["x"]).copy()
    box = g["hi"] - g["lo"]
    cmax = 2 * rmax.max() + SKIN
    own, shift = O.borders(x, g["lo"], g["hi"], g["periodic"], cmax)
    xa = np.concatenate([x, x[own] + shift * box])
    qa = np.concatenate([g["quat"], g["quat"][own]])
    sha = np.zeros(xa.shape[0], dtype=np.int32)
    tya = np.ones(xa.shape[0], dtype=np.int32)
    tag = np.concatenate([np.arange(n), own]).astype(np.int32)
    offs, jl = O.half_list(n, xa, sha, tag, rmax, SKIN)
    K = np.full((2, 2), float(g["kn"]))
    E = np.full((2, 2), float(g["exponent"]))
    o = O.compute([(lmax, a, r) for a, r in zip(anm, rmax)], K, E, nq, n, xa, qa, tya, sha, np.arange(n, dtype=np.int32),
                  offs, jl, eflag=True, nthreads=8)
    f, t = o["f"][:n].copy(), o["torque"][:n].copy()
    np.add.at(f, own, o["f"][n:])
    np.add.at(t, own, o["torque"][n:])
    return dict(rmax=rmax, x_wrapped=x, ghost_owner=own, ghost_shift=shift, offsets=offs, jlist=jl, f=f, torque=t,
                energy=o["eng_virial"][0], counts=o["counts"])


if __name__ == "__main__":
    src = np.load(sys.argv[1])
    keep = {k: src[k] for k in ("lmax", "nq", "kn", "exponent", "anm", "gravity", "lo", "hi", "periodic", "x", "quat",
                                "mask", "nmobile", "steps", "mass")}
    e = expected(keep)
    out = os.path.join(HERE, "settled_cfg1_L4.npz")
    np.savez_compressed(out, skin=SKIN, **keep, **e)
    nm = int(keep["nmobile"])
    w = float(keep["mass"]) * abs(keep["gravity"][2])
    net = e["f"][:nm] + float(keep["mass"]) * keep["gravity"]
    print("ghosts", e["ghost_owner"].size, "pairs", e["jlist"].size, "counts", e["counts"], "energy", e["energy"],
          "top", keep["x"][:nm, 2].max(), "mean |F_net|/weight", np.linalg.norm(net, axis=1).mean() / w,
          "size", os.path.getsize(out))
