"""Host-side mirror of include/shhalo.h: the N > 1 path of `pair_style sh` (BASELINE configs[3]).

Everything that moves or decides anything is in libshpair.so (csrc/shhalo_api.hip, halo_kernels.hpp, halo_plan.cpp):
brick geometry, atom migration, ghost selection, the forward / reverse exchange on RCCL point-to-point (or, for
rehearsals of N ranks on one GPU, an in-process hub between host threads) and the timestep loop over all ranks.
This module is the ctypes binding plus `RankRun`, which owns one rank's arrays (torch tensors: memory only) —
what tests and bench.py drive.  The plan functions (`plan_*`) are the library's pure host planner and need no GPU.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import HaloArrays, HaloGeometry, HaloLayout, HaloRunParams, HaloStats, ShPairError

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def _chk(rc, what=""):
    if rc:
        raise ShPairError(rc, what)


def _i3(v):
    return (C.c_int * 3)(*[int(a) for a in v])


def _d3(v):
    return (C.c_double * 3)(*[float(a) for a in v])


# ---- the pure host planner (no GPU) ------------------------------------------------------------------------------
def proc_grid(nranks):
    g = (C.c_int * 3)()
    _chk(capi.load_library().shhalo_proc_grid(int(nranks), g), "shhalo_proc_grid")
    return tuple(g)


def plan_geometry(grid, lo, hi, periodic, cut, rank):
    g = HaloGeometry()
    _chk(capi.load_library().shhalo_plan_geometry(_i3(grid), _d3(lo), _d3(hi), _i3(periodic), float(cut), int(rank), C.byref(g)),
         f"shhalo_plan_geometry(grid={tuple(grid)}, cut={cut})")
    return g


def plan_owner(geo, x):
    """Wrapped copy of x and the owning rank of every row."""
    xw = np.ascontiguousarray(x, dtype=np.float64).copy()
    owner = np.zeros(xw.shape[0], dtype=np.int32)
    _chk(capi.load_library().shhalo_plan_owner(C.byref(geo), xw.shape[0], xw.ctypes.data_as(_dp), owner.ctypes.data_as(_ip)))
    return xw, owner


def plan_ghost_mask(geo, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    mask = np.zeros(x.shape[0], dtype=np.uint32)
    _chk(capi.load_library().shhalo_plan_ghost_mask(C.byref(geo), x.shape[0], x.ctypes.data_as(_dp),
                                                    mask.ctypes.data_as(C.POINTER(C.c_uint))))
    return mask


def plan_layout(geo, send_cnt, recv_cnt):
    lay = HaloLayout()
    s = (C.c_int * 27)(*[int(v) for v in send_cnt])
    r = (C.c_int * 27)(*[int(v) for v in recv_cnt])
    _chk(capi.load_library().shhalo_plan_layout(C.byref(geo), s, r, C.byref(lay)))
    return lay


def unique_id():
    """ncclGetUniqueId as 128 bytes (rank 0; broadcast it to the other ranks)."""
    buf = C.create_string_buffer(128)
    _chk(capi.load_library().shhalo_get_unique_id(buf), "shhalo_get_unique_id (is librccl loadable?)")
    return buf.raw


_szp = C.POINTER(C.c_size_t)
_vpp = C.POINTER(C.c_void_p)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _ip, _vpp, _szp, C.c_int, _ip, _vpp, _szp)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)


class GlooStaged:
    """The caller's side of the host-staged transport (shhalo_create_staged) on torch.distributed's CPU backend: the
    library hands over page-locked host buffers, this class moves them with isend / irecv and combines the all-reduce
    values — what MPI_Isend / MPI_Irecv / MPI_Allreduce would do in a LAMMPS host without GPU-aware MPI.  One instance
    per rank; keep it alive as long as the halo context."""

    def __init__(self, dist, group=None):
        import torch
        self.dist, self.group, self.torch = dist, group, torch
        self.calls = 0
        self.last_error = ""

        def _view(ptr, nbytes, dtype):
            buf = (C.c_char * int(nbytes)).from_address(int(ptr))
            return torch.frombuffer(buf, dtype=dtype)

        def _exchange(user, ns, speer, sptr, sbytes, nr, rpeer, rptr, rbytes):
            try:
                reqs = [dist.irecv(_view(rptr[k], rbytes[k], torch.uint8), src=int(rpeer[k]), group=group) for k in range(nr)]
                reqs += [dist.isend(_view(sptr[k], sbytes[k], torch.uint8), dst=int(speer[k]), group=group) for k in range(ns)]
                for r in reqs:
                    r.wait()
                self.calls += 1
                return 0
            except BaseException as e:  # noqa: BLE001 — never let an exception cross the C frame
                self.last_error = repr(e)
                return 1

        def _allreduce(user, data, n, kind):
            try:
                if kind == 0:
                    t = _view(data, 4 * n, torch.int32)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                else:
                    t = _view(data, 8 * n, torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                return 0
            except BaseException as e:  # noqa: BLE001
                self.last_error = repr(e)
                return 1
        self.exchange_fn = EXCHANGE_FN(_exchange)
        self.allreduce_fn = ALLREDUCE_FN(_allreduce)


class Hub:
    """In-process transport between the rank THREADS of one process (rehearsal of N ranks on one GPU)."""

    def __init__(self, nranks):
        self._lib = capi.load_library()
        h = C.c_void_p()
        _chk(self._lib.shhalo_hub_create(C.byref(h), int(nranks)))
        self._h, self.nranks = h, nranks

    def close(self):
        if getattr(self, "_h", None):
            self._lib.shhalo_hub_destroy(self._h)
            self._h = None


class Halo:
    """One rank's shhalo context."""

    def __init__(self, sp, rank, nranks, grid, lo, hi, periodic, skin, hub=None, unique_id_bytes=None, staged=None):
        self._lib = capi.load_library()
        self.sp = sp
        h = C.c_void_p()
        self._staged = staged    # keeps the callbacks alive
        if staged is not None:
            rc = self._lib.shhalo_create_staged(C.byref(h), sp._h, C.cast(staged.exchange_fn, C.c_void_p), C.cast(staged.allreduce_fn, C.c_void_p),
                                                None, int(rank), int(nranks), _i3(grid), _d3(lo), _d3(hi), _i3(periodic), float(skin))
        elif unique_id_bytes is not None:
            idb = C.create_string_buffer(bytes(unique_id_bytes), 128)
            rc = self._lib.shhalo_create_rccl(C.byref(h), sp._h, idb, int(rank), int(nranks), _i3(grid), _d3(lo), _d3(hi),
                                              _i3(periodic), float(skin))
        else:
            rc = self._lib.shhalo_create_local(C.byref(h), sp._h, hub._h if hub is not None else None, int(rank), int(nranks),
                                               _i3(grid), _d3(lo), _d3(hi), _i3(periodic), float(skin))
        if rc:
            raise ShPairError(rc, self._lib.shpair_last_error(sp._h).decode())
        self._h = h
        self.rank, self.nranks = rank, nranks

    def close(self):
        if getattr(self, "_h", None):
            self._lib.shhalo_destroy(self._h)
            self._h = None

    def _chk(self, rc):
        if rc:
            raise ShPairError(rc, self._lib.shhalo_last_error(self._h).decode())

    def geometry(self):
        g = HaloGeometry()
        self._chk(self._lib.shhalo_get_geometry(self._h, C.byref(g)))
        return g

    def exchange(self, arrays, stream=None):
        self._chk(self._lib.shhalo_exchange_device(self._h, C.byref(arrays), stream))

    def borders(self, arrays, stream=None):
        ng = C.c_int(0)
        self._chk(self._lib.shhalo_borders_device(self._h, C.byref(arrays), C.byref(ng), stream))
        return ng.value

    def neighbor_build(self, arrays, nghost, stream=None):
        """shhalo_neighbor_build_device: the half list over the brick + ghosts; collective (failures are agreed on)."""
        np_ = C.c_int(0)
        self._chk(self._lib.shhalo_neighbor_build_device(self._h, C.byref(arrays), int(nghost), C.byref(np_), stream))
        return np_.value

    def forward(self, x, quat, stream=None):
        self._chk(self._lib.shhalo_forward_device(self._h, x, quat, stream))

    def reverse(self, f, torque, stream=None):
        self._chk(self._lib.shhalo_reverse_device(self._h, f, torque, stream))

    def check_rebuild(self, nlocal, x, stream=None):
        r = C.c_int(0)
        self._chk(self._lib.shhalo_check_rebuild_device(self._h, int(nlocal), x, C.byref(r), stream))
        return bool(r.value)

    def allreduce_sum(self, data_ptr, n, stream=None):
        self._chk(self._lib.shhalo_allreduce_sum_device(self._h, data_ptr, int(n), stream))

    def transport_selftest(self, nbytes=1 << 20, stream=None):
        """shhalo_transport_selftest: this rank sends nbytes to itself through the transport's exchange (RCCL: ncclSend /
        ncclRecv with the caller as peer) and all-reduces three doubles and two ints; raises on a mismatch."""
        self._chk(self._lib.shhalo_transport_selftest(self._h, int(nbytes), stream))

    def stats(self):
        s = HaloStats()
        self._chk(self._lib.shhalo_get_stats(self._h, C.byref(s)))
        return {n: getattr(s, n) for n, _ in HaloStats._fields_}

    def run(self, arrays, params, nsteps, nghost, stream=None, timed=False):
        """shhalo_run_device. Returns (nghost, rebuilds, kernel_ms_sum)."""
        ng, nr, ms = C.c_int(int(nghost)), C.c_int(0), C.c_double(0.0)
        self._chk(self._lib.shhalo_run_device(self._h, C.byref(arrays), C.byref(params), int(nsteps), C.byref(ng), C.byref(nr),
                                              C.byref(ms) if timed else None, stream))
        return ng.value, nr.value, ms.value


class RankRun:
    """One rank of a device-resident multi-rank run: its arrays (torch owns the memory) and the calls into the C ABI.

    Every rank (process with RCCL, or thread with a Hub) constructs one with ITS atoms and then calls the same
    methods in the same order (they are collective where LAMMPS' Comm calls are)."""

    def __init__(self, sp, halo, x, quat, shtype, tag, type_=None, v=None, angmom=None, mask=None, groupbit=1, dt=1e-3,
                 gravity=(0.0, 0.0, 0.0), gamma_t=0.0, gamma_r=0.0, device="cuda:0", capacity=None, check_every=1):
        import torch
        self.torch = torch
        self.sp, self.halo = sp, halo
        self.dt, self.groupbit, self.check_every = float(dt), int(groupbit), max(1, int(check_every))
        self.g = np.asarray(gravity, dtype=np.float64)
        self.gamma_t, self.gamma_r = float(gamma_t), float(gamma_r)
        self.dev = torch.device(device)
        n = int(np.asarray(x).shape[0])
        self.nmax = int(capacity) if capacity is not None else int(3.0 * max(n, 64)) + 256
        f64 = dict(dtype=torch.float64, device=self.dev)
        i32 = dict(dtype=torch.int32, device=self.dev)
        self.x = torch.zeros(self.nmax, 3, **f64)
        self.q = torch.zeros(self.nmax, 4, **f64)
        self.q[:, 0] = 1.0
        self.v = torch.zeros(self.nmax, 3, **f64)
        self.L = torch.zeros(self.nmax, 3, **f64)
        self.f = torch.zeros(self.nmax, 3, **f64)
        self.tq = torch.zeros(self.nmax, 3, **f64)
        self.tag = torch.zeros(self.nmax, **i32)
        self.sh = torch.zeros(self.nmax, **i32)
        self.ty = torch.ones(self.nmax, **i32)
        self.mask = torch.ones(self.nmax, **i32)
        self.ev = torch.zeros(7, **f64)
        self.en = torch.zeros(3, **f64)

        def put(dst, src, dt_):
            if src is not None and n:
                dst[:n] = torch.from_numpy(np.ascontiguousarray(src, dtype=dt_)).to(self.dev)
        put(self.x, x, np.float64); put(self.q, quat, np.float64); put(self.v, v, np.float64); put(self.L, angmom, np.float64)
        put(self.tag, tag, np.int32); put(self.sh, shtype, np.int32); put(self.ty, type_, np.int32); put(self.mask, mask, np.int32)
        self.stream = sp.own_stream()
        a = HaloArrays()
        a.nlocal, a.nmax = n, self.nmax
        a.x, a.v, a.quat, a.angmom = self.x.data_ptr(), self.v.data_ptr(), self.q.data_ptr(), self.L.data_ptr()
        a.f, a.torque = self.f.data_ptr(), self.tq.data_ptr()
        a.type, a.shtype, a.mask, a.tag = self.ty.data_ptr(), self.sh.data_ptr(), self.mask.data_ptr(), self.tag.data_ptr()
        self.a = a
        self.nghost = 0
        self.npairs = 0
        self.builds = 0
        self.kernel_ms = 0.0
        torch.cuda.synchronize()   # the tensors were filled on torch's stream; everything below runs on the context's
        self.rebuild()
        self.force()

    @property
    def n(self):
        return self.a.nlocal

    def sync(self):
        self.sp.synchronize()

    def rebuild(self):
        """Comm::exchange + Comm::borders + Neighbor::build."""
        self.halo.exchange(self.a, self.stream)
        self.nghost = self.halo.borders(self.a, self.stream)
        self.npairs = self.halo.neighbor_build(self.a, self.nghost, self.stream)
        self.builds += 1

    def force(self, eflag=False):
        sp, a, st = self.sp, self.a, self.stream
        self.sync()
        self.f.zero_()
        self.tq.zero_()
        if eflag:
            self.ev.zero_()
        self.torch.cuda.synchronize()
        self.halo.forward(a.x, a.quat, st)
        sp.compute_device(a.nlocal, self.nghost, a.x, a.quat, a.type, a.shtype, a.f, a.torque, eflag=eflag,
                          ev=self.ev.data_ptr() if eflag else None, stream=st)
        self.halo.reverse(a.f, a.torque, st)
        if (np.any(self.g != 0) or self.gamma_t != 0 or self.gamma_r != 0) and a.nlocal:
            sp.post_force_device(a.nlocal, self.g, self.gamma_t, self.gamma_r, a.v, a.quat, a.angmom, a.shtype, a.mask, a.f,
                                 a.torque, groupbit=self.groupbit, stream=st)
        self.sync()

    def run(self, nsteps, eflag_last=False, timed=False):
        """Verlet::run over all ranks inside the library (shhalo_run_device)."""
        p = HaloRunParams()
        p.dt, p.groupbit = self.dt, self.groupbit
        p.gravity = (self.g[0], self.g[1], self.g[2])
        p.gamma_t, p.gamma_r, p.check_every = self.gamma_t, self.gamma_r, self.check_every
        p.eflag_last = 1 if eflag_last else 0
        p.ev_dev = self.ev.data_ptr() if eflag_last else None
        if eflag_last:
            self.ev.zero_()
            self.torch.cuda.synchronize()
        self.nghost, nreb, ms = self.halo.run(self.a, p, nsteps, self.nghost, stream=self.stream, timed=timed)
        self.builds += nreb
        self.kernel_ms += ms
        return nreb

    def save_state(self):
        """Everything a rank needs to come back to this instant: the owned rows of every per-atom array (clones on the
        device) and their count.  Ghosts, plan and list are NOT saved: restore_state() rebuilds them."""
        self.sync()
        n = self.a.nlocal
        keep = {k: getattr(self, k)[:n].clone() for k in ("x", "q", "v", "L", "tag", "sh", "ty", "mask")}
        self.torch.cuda.synchronize()
        return dict(n=n, arrays=keep)

    def restore_state(self, state):
        """Back to save_state()'s instant: owned rows, then (collectively, every rank calls it) migration + ghost plan +
        list + forces, as after the constructor.  The list is built under the context's CURRENT options (a changed
        "halo_overlap" takes effect here)."""
        self.sync()
        n = state["n"]
        for k, t in state["arrays"].items():
            getattr(self, k)[:n] = t
        self.a.nlocal = n
        self.torch.cuda.synchronize()
        self.rebuild()
        self.force()

    def owned(self):
        """(tag, x, v, quat, f, torque) of the owned atoms on the host, sorted by tag."""
        self.sync()
        n = self.a.nlocal
        t = self.tag[:n].cpu().numpy()
        o = np.argsort(t)
        return (t[o], self.x[:n].cpu().numpy()[o], self.v[:n].cpu().numpy()[o], self.q[:n].cpu().numpy()[o],
                self.f[:n].cpu().numpy()[o], self.tq[:n].cpu().numpy()[o])
