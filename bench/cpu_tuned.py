"""Loader of bench/cpu_tuned.c (the tuned CPU baseline of bench.py; see the header of the C file).  Compiled on the
machine it runs on (`-march=native`), so the library is keyed by the CPU's model and flags."""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _cpu_key():
    try:
        txt = open("/proc/cpuinfo").read()
        model = next((ln for ln in txt.splitlines() if ln.startswith("model name")), "")
        flags = next((ln for ln in txt.splitlines() if ln.startswith("flags")), "")
        return hashlib.sha1((model + flags).encode()).hexdigest()[:12]
    except OSError:
        return "generic"


def build():
    out_dir = os.path.join(HERE, "_build")
    os.makedirs(out_dir, exist_ok=True)
    src = os.path.join(HERE, "cpu_tuned.c")
    lib = os.path.join(out_dir, f"libcpu_tuned_{_cpu_key()}.so")
    if not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(src):
        tmp = lib + f".{os.getpid()}.tmp"
        subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-fno-math-errno", "-shared", "-fPIC", src, "-o", tmp,
                               "-lm"])
        os.replace(tmp, lib)
    return lib


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.sht_compute.restype = None
    return _LIB


def compute(shapes, kn, expo, nq, nlocal, x, quat, type_, shtype, ilist, offs, jlist, newton_pair=True, eflag=False,
            force_volume=False, nthreads=1):
    """Same arguments as oracle.compute (shapes: list of (lmax, anm, rmax)).  Returns dict(f, torque, energy, counts)."""
    def d(a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        return a, a.ctypes.data_as(_dp)

    def i(a):
        a = np.ascontiguousarray(a, dtype=np.int32)
        return a, a.ctypes.data_as(_ip)
    lmax, plm = i([s[0] for s in shapes])
    aoff, pao = i(np.cumsum([0] + [np.asarray(s[1]).size for s in shapes])[:-1])
    anm, paa = d(np.concatenate([np.asarray(s[1], dtype=np.float64).ravel() for s in shapes]))
    rmax, prm = d([s[2] for s in shapes])
    kn, pkn = d(kn)
    expo, pex = d(expo)
    x, px = d(x)
    quat, pq = d(quat)
    type_, pt = i(type_)
    shtype, ps = i(shtype)
    ilist, pil = i(ilist)
    offs, pof = i(offs)
    jlist, pjl = i(jlist)
    nall = x.shape[0]
    f, tq = np.zeros((nall, 3)), np.zeros((nall, 3))
    e = C.c_double(0.0)
    counts = np.zeros(3, dtype=np.int64)
    lib().sht_compute(len(shapes), plm, pao, paa, prm, kn.shape[0] - 1, pkn, pex, int(nq), int(nlocal), px, pq, pt, ps,
                      ilist.size, pil, pof, pjl, int(newton_pair), int(eflag), int(force_volume), f.ctypes.data_as(_dp),
                      tq.ctypes.data_as(_dp), C.byref(e), counts.ctypes.data_as(C.POINTER(C.c_longlong)), int(nthreads))
    return dict(f=f, torque=tq, energy=e.value, counts=counts)
