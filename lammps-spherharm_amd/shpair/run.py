"""Device-resident timestep loop over the C ABI (include/shpair.h + include/shstep.h).

The host-side mirror of what LAMMPS' Verlet::run does around PairSH::compute for ONE rank whose atoms
live in HBM: initial_integrate -> neighbour decide (borders + build when an atom moved skin/2) ->
forward ghosts -> clear -> pair compute -> reverse ghosts -> post_force -> final_integrate.  Every
array stays on the GPU; torch only owns the memory.  LAMMPS itself is out of scope;
this driver exists so that tests and bench.py can time and check whole steps.
"""
import numpy as np
import torch


class DeviceRun:
    def __init__(self, sp, x, quat, shtype, lo, hi, periodic, skin, type_=None, dt=1e-3, gravity=(0.0, 0.0, 0.0),
                 gamma_t=0.0, gamma_r=0.0, mask=None, groupbit=1, ghost_factor=None, device="cuda:0", check=True):
        self.sp, self.dt, self.groupbit, self.check = sp, float(dt), int(groupbit), check
        self.g = np.asarray(gravity, dtype=np.float64)
        self.gamma_t, self.gamma_r = float(gamma_t), float(gamma_r)
        self.body_forces = bool(np.any(self.g != 0.0) or gamma_t != 0.0 or gamma_r != 0.0)
        n = x.shape[0]
        self.n = n
        if ghost_factor is None:
            # ghosts live in a shell of one ghost cutoff around the periodic faces
            ext = np.asarray(hi, float) - np.asarray(lo, float)
            cm = 2.0 * max(sp.rmax(s) for s in range(sp.nshapes)) + skin
            shell = np.prod(ext + 2.0 * cm * np.asarray(periodic, float)) / np.prod(ext)
            ghost_factor = 1.3 * shell + 0.05
        self.nmax = int(n * ghost_factor) + 64
        dev = torch.device(device)
        self.dev = dev
        f64 = dict(dtype=torch.float64, device=dev)
        i32 = dict(dtype=torch.int32, device=dev)
        self.x = torch.zeros(self.nmax, 3, **f64)
        self.q = torch.zeros(self.nmax, 4, **f64)
        self.ty = torch.ones(self.nmax, **i32)
        self.sh = torch.zeros(self.nmax, **i32)
        self.x[:n] = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
        self.q[:n] = torch.from_numpy(np.ascontiguousarray(quat)).to(dev)
        self.sh[:n] = torch.from_numpy(np.ascontiguousarray(shtype, dtype=np.int32)).to(dev)
        if type_ is not None:
            self.ty[:n] = torch.from_numpy(np.ascontiguousarray(type_, dtype=np.int32)).to(dev)
        self.v = torch.zeros(n, 3, **f64)
        self.L = torch.zeros(n, 3, **f64)
        self.mask = (torch.ones(n, **i32) if mask is None
                     else torch.from_numpy(np.ascontiguousarray(mask, dtype=np.int32)).to(dev))
        self.f = torch.zeros(self.nmax, 3, **f64)
        self.tq = torch.zeros(self.nmax, 3, **f64)
        self.ev = torch.zeros(7, **f64)
        self.en = torch.zeros(3, **f64)
        self.nghost = 0
        self.npairs = 0
        self.builds = 0
        self.steps = 0
        sp.set_box(lo, hi, periodic, skin)
        self.rebuild()
        self.force()

    # -- pieces ---------------------------------------------------------------------------------
    def rebuild(self):
        sp = self.sp
        self.nghost = sp.borders_device(self.n, self.nmax, self.x.data_ptr(), self.q.data_ptr(), self.ty.data_ptr(),
                                        self.sh.data_ptr())
        self.npairs = sp.neighbor_build_device(self.n, self.nghost, self.x.data_ptr(), self.sh.data_ptr())
        self.builds += 1

    def force(self, eflag=False):
        sp, n = self.sp, self.n
        self.f.zero_()
        self.tq.zero_()
        if eflag:
            self.ev.zero_()
        sp.forward_device(self.x.data_ptr(), self.q.data_ptr())
        sp.compute_device(n, self.nghost, self.x.data_ptr(), self.q.data_ptr(), self.ty.data_ptr(), self.sh.data_ptr(),
                          self.f.data_ptr(), self.tq.data_ptr(), eflag=eflag, ev=self.ev.data_ptr() if eflag else None)
        sp.reverse_device(self.f.data_ptr(), self.tq.data_ptr())
        if self.body_forces:
            sp.post_force_device(n, self.g, self.gamma_t, self.gamma_r, self.v.data_ptr(), self.q.data_ptr(),
                                 self.L.data_ptr(), self.sh.data_ptr(), self.mask.data_ptr(), self.f.data_ptr(),
                                 self.tq.data_ptr(), groupbit=self.groupbit)

    def _nve(self, phase):
        self.sp.nve_device(phase, self.n, self.dt, self.x.data_ptr(), self.v.data_ptr(), self.q.data_ptr(),
                           self.L.data_ptr(), self.f.data_ptr(), self.tq.data_ptr(), self.sh.data_ptr(),
                           self.mask.data_ptr(), groupbit=self.groupbit)

    # -- Verlet::run ----------------------------------------------------------------------------
    def step(self, eflag=False):
        self._nve(0)
        if self.check and self.sp.neighbor_check_device(self.n, self.x.data_ptr()):
            self.rebuild()
        self.force(eflag)
        self._nve(1)
        self.steps += 1

    def run(self, nsteps, eflag_last=False):
        for k in range(nsteps):
            self.step(eflag=eflag_last and k == nsteps - 1)

    def run_native(self, nsteps, use_graph=False, check_every=1):
        """The same loop inside the library (shstep_run_device): C++ host code, optionally replayed from
        captured hipGraphs — for small, launch-bound systems.  Runs on the context's own stream."""
        from .capi import StepArrays
        a = StepArrays()
        a.nlocal, a.nmax = self.n, self.nmax
        a.x, a.v, a.quat, a.angmom = self.x.data_ptr(), self.v.data_ptr(), self.q.data_ptr(), self.L.data_ptr()
        a.f, a.torque = self.f.data_ptr(), self.tq.data_ptr()
        a.type, a.shtype, a.mask = self.ty.data_ptr(), self.sh.data_ptr(), self.mask.data_ptr()
        a.groupbit, a.dt = self.groupbit, self.dt
        a.gravity = (self.g[0], self.g[1], self.g[2])
        a.gamma_t, a.gamma_r, a.check_every = self.gamma_t, self.gamma_r, int(check_every)
        torch.cuda.synchronize()          # everything enqueued on torch's stream is visible to the context's stream
        self.nghost, nreb = self.sp.run_device(a, nsteps, self.nghost, use_graph=use_graph, stream=self.sp.own_stream())
        self.builds += nreb
        self.steps += nsteps

    def energies(self):
        """(contact energy of the last eflag force call, translational KE, rotational KE, gravitational PE)."""
        self.en.zero_()
        self.sp.energies_device(self.n, self.g, self.x.data_ptr(), self.v.data_ptr(), self.q.data_ptr(), self.L.data_ptr(),
                                self.sh.data_ptr(), self.mask.data_ptr(), self.en.data_ptr(), groupbit=self.groupbit)
        torch.cuda.synchronize()
        e = self.en.cpu().numpy()
        return float(self.ev[0].item()), float(e[0]), float(e[1]), float(e[2])
