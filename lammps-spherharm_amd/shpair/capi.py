"""ctypes binding of include/shpair.h (the C ABI of the HIP contact path).

Mirrors the LAMMPS call order of the reference's PairSH (sources ABSENT FROM
MOUNT): settings() -> coeff() -> init_style()/init_one() ->
[neighbour build] -> compute(eflag, vflag).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


class ShPairError(RuntimeError):
    def __init__(self, code, detail=""):
        self.code = code
        msg = load_library().shpair_strerror(code).decode()
        super().__init__(f"shpair error {code}: {msg}" + (f" — {detail}" if detail else ""))


class Stats(C.Structure):
    _fields_ = [("n_candidates", C.c_longlong), ("n_contact", C.c_longlong), ("n_touching", C.c_longlong),
                ("kernel_ms", C.c_double), ("total_ms", C.c_double)]


class KernelInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("lmax", "compiled_order", "vgprs", "scratch_bytes", "lds_bytes_per_wave", "ring_rows",
                                       "waves_per_simd_vgpr", "waves_per_cu_lds", "waves_per_cu", "family", "waves_per_pair",
                                       "needv", "weighted", "queue_entries", "specialised")]


class StepArrays(C.Structure):
    """shstep_arrays of include/shstep.h."""
    _fields_ = [("nlocal", C.c_int), ("nmax", C.c_int),
                ("x", C.c_void_p), ("v", C.c_void_p), ("quat", C.c_void_p), ("angmom", C.c_void_p), ("f", C.c_void_p),
                ("torque", C.c_void_p), ("type", C.c_void_p), ("shtype", C.c_void_p), ("mask", C.c_void_p),
                ("groupbit", C.c_int), ("dt", C.c_double), ("gravity", C.c_double * 3), ("gamma_t", C.c_double),
                ("gamma_r", C.c_double), ("check_every", C.c_int)]


class HaloGeometry(C.Structure):
    """shhalo_geometry of include/shhalo.h."""
    _fields_ = [("grid", C.c_int * 3), ("coord", C.c_int * 3), ("rank", C.c_int), ("nranks", C.c_int),
                ("periodic", C.c_int * 3), ("lo", C.c_double * 3), ("hi", C.c_double * 3), ("blo", C.c_double * 3),
                ("bhi", C.c_double * 3), ("cut", C.c_double), ("peer", C.c_int * 27), ("shift", (C.c_double * 3) * 27)]


class HaloLayout(C.Structure):
    """shhalo_layout of include/shhalo.h."""
    _fields_ = [("send_off", C.c_int * 27), ("send_cnt", C.c_int * 27), ("recv_off", C.c_int * 27), ("recv_cnt", C.c_int * 27),
                ("nsend", C.c_int), ("nghost", C.c_int), ("npeers", C.c_int), ("peer_rank", C.c_int * 26),
                ("peer_send_off", C.c_int * 26), ("peer_send_cnt", C.c_int * 26), ("peer_recv_off", C.c_int * 26),
                ("peer_recv_cnt", C.c_int * 26)]


class HaloArrays(C.Structure):
    """shhalo_arrays of include/shhalo.h."""
    _fields_ = [("nlocal", C.c_int), ("nmax", C.c_int), ("x", C.c_void_p), ("v", C.c_void_p), ("quat", C.c_void_p),
                ("angmom", C.c_void_p), ("f", C.c_void_p), ("torque", C.c_void_p), ("type", C.c_void_p), ("shtype", C.c_void_p),
                ("mask", C.c_void_p), ("tag", C.c_void_p)]


class HaloStats(C.Structure):
    _fields_ = [("nranks_transport", C.c_int), ("npeers", C.c_int), ("nsend_rows", C.c_int), ("nghost_rows", C.c_int),
                ("rebuilds", C.c_longlong), ("migrated_out", C.c_longlong), ("migrated_in", C.c_longlong),
                ("forward_bytes_per_step", C.c_longlong), ("reverse_bytes_per_step", C.c_longlong), ("transport", C.c_int),
                ("rccl_version", C.c_int)]


class HaloRunParams(C.Structure):
    _fields_ = [("dt", C.c_double), ("groupbit", C.c_int), ("gravity", C.c_double * 3), ("gamma_t", C.c_double),
                ("gamma_r", C.c_double), ("check_every", C.c_int), ("eflag_last", C.c_int), ("ev_dev", C.c_void_p)]


_up = C.POINTER(C.c_uint)
_idp = C.c_char_p  # SHHALO_UNIQUE_ID_BYTES raw bytes

# every symbol include/shpair.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "shpair_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "shpair_destroy": (None, [C.c_void_p]),
    "shpair_strerror": (C.c_char_p, [C.c_int]),
    "shpair_last_error": (C.c_char_p, [C.c_void_p]),
    "shpair_version": (C.c_char_p, []),
    "shpair_settings": (C.c_int, [C.c_void_p, C.c_int]),
    "shpair_set_ntypes": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "shpair_set_shape": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, C.c_double]),
    "shpair_set_coeff": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_double]),
    "shpair_get_rmax": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "shpair_shape_radius": (C.c_int, [C.c_int, _dp, _dp, _dp]),
    "shpair_shape_default_rmax": (C.c_int, [C.c_int, _dp, _dp]),
    "shpair_set_neighbors": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, C.POINTER(_ip)]),
    "shpair_set_neighbors_csr": (C.c_int, [C.c_void_p, C.c_int, _ip, _ip, _ip]),
    "shpair_set_neighbors_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_int, C.c_void_p]),
    "shpair_compute": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _ip, _ip, C.c_int, C.c_int, C.c_int,
                                 _dp, _dp, _dp, _dp]),
    "shpair_compute_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p]),
    "shpair_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "shpair_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "shpair_get_kernel_info": (C.c_int, [C.c_void_p, C.POINTER(KernelInfo)]),
    "shpair_set_pair_output": (C.c_int, [C.c_void_p, C.c_void_p]),
    "shpair_set_peratom_output": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "shpair_set_peratom_host": (C.c_int, [C.c_void_p, _dp, _dp]),
    "shpair_pin_host": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "shpair_unpin_host": (C.c_int, [C.c_void_p, C.c_void_p]),
    "shpair_fp64_peak": (C.c_int, [C.c_void_p, C.c_int, C.c_double, _dp, _dp]),
    "shpair_get_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "shpair_synchronize": (C.c_int, [C.c_void_p]),
    # include/shstep.h
    "shstep_shape_mass_props": (C.c_int, [C.c_int, _dp, _dp]),
    "shstep_set_density": (C.c_int, [C.c_void_p, C.c_int, C.c_double]),
    "shstep_get_body": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
    "shstep_nve_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double] + [C.c_void_p] * 8 + [C.c_int, C.c_void_p]),
    "shstep_nve": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp, _dp, _ip, _ip, C.c_int]),
    "shstep_force_clear_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "shstep_post_force_device": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_double, C.c_double] + [C.c_void_p] * 5 +
                                 [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "shstep_energies_device": (C.c_int, [C.c_void_p, C.c_int, _dp] + [C.c_void_p] * 6 + [C.c_int, C.c_void_p, C.c_void_p]),
    "shstep_set_box": (C.c_int, [C.c_void_p, _dp, _dp, _ip, C.c_double]),
    "shstep_borders_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5 + [_ip, C.c_void_p]),
    "shstep_forward_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "shstep_reverse_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "shstep_neighbor_build_device": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, _ip,
                                               C.c_void_p]),
    "shstep_neighbor_check_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _ip, C.c_void_p]),
    "shstep_copy_neighbors": (C.c_int, [C.c_void_p, _ip, _ip]),
    "shstep_run_device": (C.c_int, [C.c_void_p, C.POINTER(StepArrays), C.c_int, C.c_int, _ip, _ip, C.c_void_p]),
    # include/shhalo.h
    "shhalo_proc_grid": (C.c_int, [C.c_int, _ip]),
    "shhalo_plan_geometry": (C.c_int, [_ip, _dp, _dp, _ip, C.c_double, C.c_int, C.POINTER(HaloGeometry)]),
    "shhalo_plan_owner": (C.c_int, [C.POINTER(HaloGeometry), C.c_int, _dp, _ip]),
    "shhalo_plan_ghost_mask": (C.c_int, [C.POINTER(HaloGeometry), C.c_int, _dp, _up]),
    "shhalo_plan_layout": (C.c_int, [C.POINTER(HaloGeometry), _ip, _ip, C.POINTER(HaloLayout)]),
    "shhalo_get_unique_id": (C.c_int, [C.c_void_p]),
    "shhalo_create_rccl": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_int, _ip, _dp, _dp, _ip,
                                     C.c_double]),
    "shhalo_hub_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "shhalo_hub_destroy": (None, [C.c_void_p]),
    "shhalo_create_staged": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, _ip, _dp,
                                       _dp, _ip, C.c_double]),
    "shhalo_create_local": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_int, _ip, _dp, _dp, _ip,
                                      C.c_double]),
    "shhalo_destroy": (None, [C.c_void_p]),
    "shhalo_last_error": (C.c_char_p, [C.c_void_p]),
    "shhalo_get_geometry": (C.c_int, [C.c_void_p, C.POINTER(HaloGeometry)]),
    "shhalo_exchange_device": (C.c_int, [C.c_void_p, C.POINTER(HaloArrays), C.c_void_p]),
    "shhalo_borders_device": (C.c_int, [C.c_void_p, C.POINTER(HaloArrays), _ip, C.c_void_p]),
    "shhalo_neighbor_build_device": (C.c_int, [C.c_void_p, C.POINTER(HaloArrays), C.c_int, _ip, C.c_void_p]),
    "shhalo_forward_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "shhalo_reverse_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "shhalo_check_rebuild_device": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, _ip, C.c_void_p]),
    "shhalo_allreduce_sum_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "shhalo_transport_selftest": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "shhalo_get_stats": (C.c_int, [C.c_void_p, C.POINTER(HaloStats)]),
    "shhalo_run_device": (C.c_int, [C.c_void_p, C.POINTER(HaloArrays), C.POINTER(HaloRunParams), C.c_int, _ip, _ip, _dp,
                                    C.c_void_p]),
}


def library_path():
    """libshpair.so beside this file.  SHPAIR_LIB=<file name in this directory> selects a diagnostic build (ablation /
    statistics libraries made by `make abl` / `make stats`: some give WRONG results by construction) — honoured only
    together with SHPAIR_DIAGNOSTIC=1, refused loudly otherwise; there is no other fallback."""
    name = os.environ.get("SHPAIR_LIB")
    if name and name != "libshpair.so":
        if os.environ.get("SHPAIR_DIAGNOSTIC") != "1":
            raise ImportError(f"SHPAIR_LIB={name} selects a diagnostic build of libshpair; set SHPAIR_DIAGNOSTIC=1 as well "
                              "if that is what you mean (profiling tools do), or unset SHPAIR_LIB")
        return os.path.join(_HERE, name)
    return os.path.join(_HERE, "libshpair.so")


def library_name():
    """Base name of the library this process loaded (bench.py prints it in its JSON line)."""
    return os.path.basename(library_path())


def load_library():
    """Loads libshpair.so. Raises (never falls back) if it is not built."""
    global _LIB
    if _LIB is None:
        path = library_path()
        if not os.path.exists(path):
            raise ImportError(f"{path} is not built: run `make -C lammps-spherharm_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        # PyTorch-ROCm bundles its own libamdhip64.so.7 (same SONAME as /opt/rocm's).
        # Whichever is loaded first serves the whole process, and torch cannot see the
        # GPU behind the system runtime ("No HIP GPUs are available"). So when torch is
        # installed it is imported first and libshpair.so binds to torch's runtime; a
        # LAMMPS process (no torch) binds to /opt/rocm's through the library's RUNPATH.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            if os.environ.get("SHPAIR_AB_OLD_LIB") and not hasattr(lib, name):
                continue  # tools/ab_libs.py timing an older build that predates an entry point
            fn = getattr(lib, name)  # AttributeError if the ABI lost a symbol
            fn.restype = res
            fn.argtypes = args
        _LIB = lib
    return _LIB


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def shape_radius(lmax, anm, u):
    anm, pa = _d(anm)
    u, pu = _d(u)
    r = C.c_double()
    rc = load_library().shpair_shape_radius(lmax, pa, pu, C.byref(r))
    if rc:
        raise ShPairError(rc)
    return r.value


def shape_default_rmax(lmax, anm):
    anm, pa = _d(anm)
    r = C.c_double()
    rc = load_library().shpair_shape_default_rmax(lmax, pa, C.byref(r))
    if rc:
        raise ShPairError(rc)
    return r.value


def shape_mass_props(lmax, anm):
    """(V, c[3], J_c xx,yy,zz,xy,xz,yz) at unit density (docs/SPEC.md §5)."""
    anm, pa = _d(anm)
    out = np.zeros(10)
    rc = load_library().shstep_shape_mass_props(lmax, pa, out.ctypes.data_as(_dp))
    if rc:
        raise ShPairError(rc)
    return out


class ShPair:
    """One context = one rank's `pair_style sh` instance on one GPU."""

    def __init__(self, device=0):
        self._lib = load_library()
        h = C.c_void_p()
        rc = self._lib.shpair_create(C.byref(h), device)
        if rc:
            raise ShPairError(rc, "shpair_create")
        self._h = h
        self.nshapes = 0
        self.ntypes = 0

    def close(self):
        if getattr(self, "_h", None):
            self._lib.shpair_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc):
        if rc:
            raise ShPairError(rc, self._lib.shpair_last_error(self._h).decode())

    # --- PairSH::settings / coeff ------------------------------------------------
    def settings(self, nq):
        self._chk(self._lib.shpair_settings(self._h, int(nq)))

    def set_ntypes(self, ntypes, nshapes):
        self._chk(self._lib.shpair_set_ntypes(self._h, int(ntypes), int(nshapes)))
        self.ntypes, self.nshapes = ntypes, nshapes

    def set_shape(self, ishape, lmax, anm, rmax=0.0):
        anm, pa = _d(anm)
        if anm.size != (lmax + 1) * (lmax + 2):
            raise ValueError(f"anm has {anm.size} doubles, expected {(lmax + 1) * (lmax + 2)}")
        self._chk(self._lib.shpair_set_shape(self._h, int(ishape), int(lmax), pa, float(rmax)))

    def coeff(self, itype, jtype, kn, exponent):
        """`pair_coeff I J kn exponent`; itype/jtype may be '*' or int; mirrored i<->j."""
        its = range(1, self.ntypes + 1) if itype == "*" else [int(itype)]
        jts = range(1, self.ntypes + 1) if jtype == "*" else [int(jtype)]
        for a in its:
            for b in jts:
                self._chk(self._lib.shpair_set_coeff(self._h, a, b, float(kn), float(exponent)))
                self._chk(self._lib.shpair_set_coeff(self._h, b, a, float(kn), float(exponent)))

    def rmax(self, ishape):
        r = C.c_double()
        self._chk(self._lib.shpair_get_rmax(self._h, int(ishape), C.byref(r)))
        return r.value

    def init_one(self, ishape, jshape):
        """Cutoff of a shape pair, as PairSH::init_one returns it."""
        return self.rmax(ishape) + self.rmax(jshape)

    # --- neighbour list -----------------------------------------------------------
    def set_neighbors_csr(self, ilist, offsets, jlist):
        ilist, pi = _i(ilist)
        offsets, po = _i(offsets)
        jlist, pj = _i(jlist)
        if offsets.size != ilist.size + 1:
            raise ValueError("offsets must have inum+1 entries")
        self._chk(self._lib.shpair_set_neighbors_csr(self._h, ilist.size, pi, po, pj))

    def set_neighbors(self, ilist, numneigh, firstneigh):
        """LAMMPS layout: numneigh/firstneigh indexed by atom id; firstneigh = list of int32 arrays."""
        ilist, pi = _i(ilist)
        numneigh, pn = _i(numneigh)
        keep = [np.ascontiguousarray(a, dtype=np.int32) for a in firstneigh]
        arr = (_ip * len(keep))(*[a.ctypes.data_as(_ip) for a in keep])
        self._chk(self._lib.shpair_set_neighbors(self._h, ilist.size, pi, pn, arr))

    def set_neighbors_device(self, inum, ilist_ptr, offsets_ptr, jlist_ptr, npairs, max_atom_index, stream=None):
        """CSR half list already on the device (raw device addresses). Asynchronous on `stream`."""
        self._chk(self._lib.shpair_set_neighbors_device(self._h, int(inum), ilist_ptr, offsets_ptr, jlist_ptr,
                                                        int(npairs), int(max_atom_index), stream))

    # --- PairSH::compute ------------------------------------------------------------
    def compute(self, nlocal, x, quat, type_, shtype, newton_pair=True, eflag=False, vflag=False,
                f=None, torque=None):
        """Host-pointer entry point. Returns (f, torque, eng_vdwl, virial[6]); adds into f/torque if given."""
        x, px = _d(x)
        quat, pq = _d(quat)
        type_, pt = _i(type_)
        shtype, ps = _i(shtype)
        nall = x.shape[0]
        if quat.shape != (nall, 4) or type_.size != nall or shtype.size != nall:
            raise ValueError("atom arrays disagree on nall")
        if f is None:
            f = np.zeros((nall, 3))
        if torque is None:
            torque = np.zeros((nall, 3))
        assert f.flags.c_contiguous and torque.flags.c_contiguous and f.dtype == np.float64
        eng = C.c_double(0.0)
        vir = np.zeros(6)
        self._chk(self._lib.shpair_compute(self._h, int(nlocal), int(nall - nlocal), px, pq, pt, ps,
                                           int(newton_pair), int(eflag), int(vflag),
                                           f.ctypes.data_as(_dp), torque.ctypes.data_as(_dp),
                                           C.byref(eng), vir.ctypes.data_as(_dp)))
        return f, torque, eng.value, vir

    def pin_host(self, array):
        """shpair_pin_host: page-locks a numpy array that will be handed to compute() repeatedly (it must outlive the
        pin; unpin_host(array) or close() releases it)."""
        self._chk(self._lib.shpair_pin_host(self._h, C.c_void_p(array.ctypes.data), C.c_size_t(array.nbytes)))

    def unpin_host(self, array):
        self._chk(self._lib.shpair_unpin_host(self._h, C.c_void_p(array.ctypes.data)))

    def compute_device(self, nlocal, nghost, x, quat, type_, shtype, f, torque, newton_pair=True,
                       eflag=False, vflag=False, ev=None, stream=None):
        """Device-pointer entry point: arguments are raw device addresses (ints). Asynchronous.
        stream: a hipStream_t as int; None/0 = HIP null stream."""
        self._chk(self._lib.shpair_compute_device(self._h, int(nlocal), int(nghost), x, quat, type_, shtype,
                                                  int(newton_pair), int(eflag), int(vflag), f, torque,
                                                  ev, stream))

    def set_option(self, key, value):
        self._chk(self._lib.shpair_set_option(self._h, key.encode(), int(value)))

    def set_peratom_output(self, eatom_ptr, vatom_ptr):
        """Device arrays eatom[nall], vatom[nall,6] that following compute_device calls add into (None = off)."""
        self._chk(self._lib.shpair_set_peratom_output(self._h, eatom_ptr, vatom_ptr))

    def set_peratom_host(self, eatom, vatom):
        """Host arrays (float64, C-contiguous; None = off) that following compute() calls add into."""
        for a in (eatom, vatom):
            assert a is None or (a.dtype == np.float64 and a.flags.c_contiguous)
        self._keep_peratom = (eatom, vatom)
        self._chk(self._lib.shpair_set_peratom_host(self._h, None if eatom is None else eatom.ctypes.data_as(_dp),
                                                    None if vatom is None else vatom.ctypes.data_as(_dp)))

    def set_pair_output(self, dev_ptr):
        self._chk(self._lib.shpair_set_pair_output(self._h, dev_ptr))

    def stats(self):
        s = Stats()
        self._chk(self._lib.shpair_get_stats(self._h, C.byref(s)))
        return dict(n_candidates=s.n_candidates, n_contact=s.n_contact, n_touching=s.n_touching,
                    kernel_ms=s.kernel_ms, total_ms=s.total_ms)

    def kernel_info(self):
        """Registers, LDS and resident waves of the pair kernel the last compute launched."""
        k = KernelInfo()
        self._chk(self._lib.shpair_get_kernel_info(self._h, C.byref(k)))
        return {n: getattr(k, n) for n, _ in KernelInfo._fields_}

    def fp64_peak(self, mode=0, target_ms=20.0):
        """Measured FP64 ceilings of this GPU in TFLOP/s: (valu, mfma) of the waves running each loop
        (mode 0 v_fma_f64 chains, 1 v_mfma_f64_16x16x4_f64, 2 both side by side)."""
        a, b = C.c_double(), C.c_double()
        self._chk(self._lib.shpair_fp64_peak(self._h, int(mode), float(target_ms), C.byref(a), C.byref(b)))
        return a.value, b.value

    def own_stream(self):
        """The context's own hipStream_t as an int (None-safe for compute_device(stream=...))."""
        st = C.c_void_p()
        self._chk(self._lib.shpair_get_stream(self._h, C.byref(st)))
        return st.value

    def synchronize(self):
        self._chk(self._lib.shpair_synchronize(self._h))

    # --- include/shstep.h: integrator, body forces, ghosts, neighbour build (device pointers as ints) ----
    def set_density(self, ishape, rho):
        self._chk(self._lib.shstep_set_density(self._h, int(ishape), float(rho)))

    def body(self, ishape):
        """(mass, com[3], inertia[6]) of a shape."""
        m = C.c_double()
        com, inertia = np.zeros(3), np.zeros(6)
        self._chk(self._lib.shstep_get_body(self._h, int(ishape), C.byref(m), com.ctypes.data_as(_dp),
                                            inertia.ctypes.data_as(_dp)))
        return m.value, com, inertia

    def nve_device(self, phase, nlocal, dt, x, v, quat, angmom, f, torque, shtype, mask, groupbit=1, stream=None):
        self._chk(self._lib.shstep_nve_device(self._h, int(phase), int(nlocal), float(dt), x, v, quat, angmom, f, torque,
                                              shtype, mask, int(groupbit), stream))

    def nve(self, phase, dt, x, v, quat, angmom, f, torque, shtype, mask, groupbit=1):
        """Host-pointer form; x, v, quat, angmom are updated in place (float64 C-contiguous)."""
        for a in (x, v, quat, angmom):
            assert a.dtype == np.float64 and a.flags.c_contiguous
        f, pf = _d(f)
        torque, pt = _d(torque)
        shtype, ps = _i(shtype)
        mask, pm = _i(mask)
        self._chk(self._lib.shstep_nve(self._h, int(phase), x.shape[0], float(dt), x.ctypes.data_as(_dp),
                                       v.ctypes.data_as(_dp), quat.ctypes.data_as(_dp), angmom.ctypes.data_as(_dp),
                                       pf, pt, ps, pm, int(groupbit)))

    def force_clear_device(self, nall, f, torque, stream=None):
        """Verlet::force_clear: f and torque of nall atoms zeroed in one launch (raw device addresses)."""
        self._chk(self._lib.shstep_force_clear_device(self._h, int(nall), f, torque, stream))

    def post_force_device(self, nlocal, gravity, gamma_t, gamma_r, v, quat, angmom, shtype, mask, f, torque,
                          groupbit=1, stream=None):
        g, pg = _d(gravity)
        self._chk(self._lib.shstep_post_force_device(self._h, int(nlocal), pg, float(gamma_t), float(gamma_r), v, quat,
                                                     angmom, shtype, mask, int(groupbit), f, torque, stream))

    def energies_device(self, nlocal, gravity, x, v, quat, angmom, shtype, mask, out3, groupbit=1, stream=None):
        g, pg = _d(gravity)
        self._chk(self._lib.shstep_energies_device(self._h, int(nlocal), pg, x, v, quat, angmom, shtype, mask,
                                                   int(groupbit), out3, stream))

    def set_box(self, lo, hi, periodic, skin):
        lo, plo = _d(lo)
        hi, phi = _d(hi)
        per, pp = _i(periodic)
        self._chk(self._lib.shstep_set_box(self._h, plo, phi, pp, float(skin)))

    def borders_device(self, nlocal, nmax, x, quat, type_, shtype, tag=None, stream=None):
        """Returns nghost (blocks)."""
        ng = C.c_int(0)
        self._chk(self._lib.shstep_borders_device(self._h, int(nlocal), int(nmax), x, quat, type_, shtype, tag,
                                                  C.byref(ng), stream))
        return ng.value

    def forward_device(self, x, quat, stream=None):
        self._chk(self._lib.shstep_forward_device(self._h, x, quat, stream))

    def reverse_device(self, f, torque, stream=None):
        self._chk(self._lib.shstep_reverse_device(self._h, f, torque, stream))

    def neighbor_build_device(self, nlocal, nghost, x, shtype, tag=None, stream=None):
        """Builds and installs the half list; returns npairs (blocks)."""
        n = C.c_int(0)
        self._chk(self._lib.shstep_neighbor_build_device(self._h, int(nlocal), int(nghost), x, shtype, tag,
                                                         C.byref(n), stream))
        return n.value

    def neighbor_check_device(self, nlocal, x, stream=None):
        r = C.c_int(0)
        self._chk(self._lib.shstep_neighbor_check_device(self._h, int(nlocal), x, C.byref(r), stream))
        return bool(r.value)

    def copy_neighbors(self, nlocal, npairs):
        offs = np.zeros(nlocal + 1, dtype=np.int32)
        jl = np.zeros(max(npairs, 1), dtype=np.int32)
        self._chk(self._lib.shstep_copy_neighbors(self._h, offs.ctypes.data_as(_ip), jl.ctypes.data_as(_ip)))
        return offs, jl[:npairs]

    def run_device(self, arrays, nsteps, nghost, use_graph=False, stream=None):
        """shstep_run_device: the whole loop in the library. Returns (nghost, rebuilds). Blocks."""
        ng = C.c_int(int(nghost))
        nr = C.c_int(0)
        self._chk(self._lib.shstep_run_device(self._h, C.byref(arrays), int(nsteps), int(bool(use_graph)), C.byref(ng),
                                              C.byref(nr), stream))
        return ng.value, nr.value
