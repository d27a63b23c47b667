"""Device-resident timestep loop over several ranks: LAMMPS' spatial decomposition with atom migration.

What Comm::exchange / Comm::borders / forward_comm / reverse_comm and Neighbor::build do around
PairSH::compute when atoms move (BASELINE configs[3]): one rank and one shpair context per
GPU, uniform bricks of a px x py x pz grid over an orthogonal box, every array resident in HBM.

At a rebuild (any rank saw an atom move skin/2):
  1. owned atoms are wrapped into the periodic box and those that left the brick migrate to their new owner
     (counts by all_to_all, then one variable-size all_to_all of packed rows);
  2. ghosts: for each of the 26 directions the atoms within the ghost cutoff of that face / edge / corner go
     to the neighbour in that direction, shifted by a box length where the hop crosses a periodic boundary
     (a grid dimension of 1 makes a rank its own neighbour: plain periodic images); the send lists are kept;
  3. the half list is built on the device from owned + ghost rows with global ids as tags, so each physical
     pair is evaluated by exactly one rank with the lower id as the integrated particle (docs/SPEC.md §7).
Every step: forward (x, quat of the send lists -> the peers' ghost rows), pair forces, reverse (ghost forces
added into their owners), integrate.  Per step and direction of travel there is ONE batched point-to-point
group with one message per peer and array: the rows for all directions that lead to the same peer are
contiguous in the gathered send buffer and in the ghost rows (both ordered by the sender's direction code).

torch.distributed is the transport: backend `nccl` (= RCCL over xGMI) with device tensors, or — for
rehearsals on a box with fewer GPUs than ranks and for tests — `gloo` with the buffers staged through the
host (`staged=True`).  torch only owns memory and moves bytes; all arithmetic is the C ABI's.
"""
import numpy as np
import torch

_PACK = 3 + 4 + 3 + 3 + 4  # x, quat, v, angmom, (tag, shtype, type, mask) as doubles


def _dir_code(dx, dy, dz):
    return (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)


class MultiRankRun:
    def __init__(self, sp, dist, rank, world, grid, lo, hi, periodic, skin, x, quat, shtype, tag, type_=None, v=None,
                 angmom=None, mask=None, groupbit=1, dt=1e-3, gravity=(0.0, 0.0, 0.0), gamma_t=0.0, gamma_r=0.0, device="cuda:0", staged=False,
                 capacity_factor=3.0, check_every=1):
        self.sp, self.dist, self.rank, self.world, self.staged = sp, dist, rank, world, staged
        self.grid = tuple(int(g) for g in grid)
        assert int(np.prod(self.grid)) == world
        self.lo = np.asarray(lo, float)
        self.hi = np.asarray(hi, float)
        self.per = np.asarray(periodic, int)
        self.len = self.hi - self.lo
        self.skin, self.dt = float(skin), float(dt)
        self.g = np.asarray(gravity, float)
        self.gamma_t, self.gamma_r = float(gamma_t), float(gamma_r)
        self.body = bool(np.any(self.g != 0) or gamma_t != 0 or gamma_r != 0)
        self.dev = torch.device(device)
        self.cut = 2.0 * max(sp.rmax(s) for s in range(sp.nshapes)) + self.skin
        px, py, pz = self.grid
        self.coord = np.array([rank // (py * pz), (rank // pz) % py, rank % pz])
        self.blen = self.len / np.array(self.grid)
        self.blo = self.lo + self.coord * self.blen
        self.bhi = self.blo + self.blen
        for d in range(3):
            if self.grid[d] > 1 and self.blen[d] < self.cut:
                raise ValueError(f"brick edge {self.blen[d]:g} in dimension {d} is shorter than the ghost cutoff {self.cut:g}")
            if self.grid[d] == 1 and self.per[d] and self.len[d] < 2 * self.cut:
                raise ValueError(f"periodic box edge {self.len[d]:g} in dimension {d} is shorter than twice the ghost cutoff")
        n = x.shape[0]
        self.nmax = int(capacity_factor * max(n, 64)) + 256
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
        self.groupbit = int(groupbit)
        self.check_every = max(1, int(check_every))   # rebuild test (a host sync and an all-reduce) every this many steps
        self.n = n
        self.nghost = 0
        self.builds = 0
        self.steps = 0
        self.migrated = 0
        self.plan = []       # per direction with a peer: dict(code, peer, send_idx, shift, recv=(a, b))
        # the C ABI bins owned + ghost rows of THIS brick: a non-periodic box around it
        sp.set_box(self.blo - self.cut, self.bhi + self.cut, (0, 0, 0), self.skin)
        self.rebuild()
        self.force()

    # ---- transport helpers ---------------------------------------------------------------------------------
    def _xfer(self, t):
        return t.cpu() if self.staged else t

    def _all_to_all_counts(self, counts):
        src = torch.tensor(counts, dtype=torch.int64, device="cpu" if self.staged else self.dev)
        dst = torch.zeros_like(src)
        self.dist.all_to_all_single(dst, src)
        return [int(c) for c in dst.cpu().tolist()]

    def _all_to_all_rows(self, rows, send_counts, recv_counts):
        """rows: [sum(send_counts), W] grouped by destination rank; returns the received rows."""
        W = rows.shape[1]
        src = self._xfer(rows.contiguous())
        dst = torch.empty(sum(recv_counts), W, dtype=rows.dtype, device=src.device)
        self.dist.all_to_all_single(dst, src, output_split_sizes=recv_counts, input_split_sizes=send_counts)
        return dst.to(self.dev)

    def _p2p(self, sends, recvs):
        """sends: [(peer, tensor)], recvs: [(peer, tensor)] in matching order per pair of ranks; device tensors."""
        dist = self.dist
        ops, staged = [], []
        for peer, t in recvs:
            buf = torch.empty(t.shape, dtype=t.dtype) if self.staged else t
            if self.staged:
                staged.append((t, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer))
        for peer, t in sends:
            ops.append(dist.P2POp(dist.isend, self._xfer(t.contiguous()), peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for t, buf in staged:
            t.copy_(buf)

    # ---- Comm::exchange + Comm::borders + Neighbor::build --------------------------------------------------
    def _owner_rank(self, x):
        c = []
        for d in range(3):
            k = torch.floor((x[:, d] - self.lo[d]) / self.blen[d]).to(torch.int64).clamp_(0, self.grid[d] - 1)
            c.append(k)
        return (c[0] * self.grid[1] + c[1]) * self.grid[2] + c[2]

    def rebuild(self):
        n = self.n
        x = self.x
        # 1. wrap (periodic dimensions) and migrate
        for d in range(3):
            if self.per[d] and n:
                xd = x[:n, d]
                xd -= torch.floor((xd - self.lo[d]) / self.len[d]) * self.len[d]
                xd.clamp_(min=self.lo[d])
                hi_fix = xd >= self.hi[d]
                xd[hi_fix] -= self.len[d]
        if self.world > 1:
            own = self._owner_rank(x[:n]) if n else torch.zeros(0, dtype=torch.int64, device=self.dev)
            stay = own == self.rank
            go = torch.nonzero(~stay).flatten()
            order = go[torch.argsort(own[go], stable=True)] if go.numel() else go
            send_counts = torch.bincount(own[order], minlength=self.world).cpu().tolist() if order.numel() else [0] * self.world
            recv_counts = self._all_to_all_counts(send_counts)
            rows = torch.cat([self.x[order], self.q[order], self.v[order], self.L[order],
                              torch.stack([self.tag[order], self.sh[order], self.ty[order], self.mask[order]], 1).to(torch.float64)], 1) \
                if order.numel() else torch.zeros(0, _PACK, dtype=torch.float64, device=self.dev)
            got = self._all_to_all_rows(rows, send_counts, recv_counts)
            keep = torch.nonzero(stay).flatten()
            nk = keep.numel()
            for a in (self.x, self.q, self.v, self.L, self.tag, self.sh, self.ty, self.mask):
                a[:nk] = a[keep]
            ng = got.shape[0]
            if nk + ng > self.nmax:
                raise MemoryError("capacity exceeded by migration")
            if ng:
                self.x[nk:nk + ng] = got[:, 0:3]
                self.q[nk:nk + ng] = got[:, 3:7]
                self.v[nk:nk + ng] = got[:, 7:10]
                self.L[nk:nk + ng] = got[:, 10:13]
                meta = got[:, 13:17].round().to(torch.int32)
                self.tag[nk:nk + ng] = meta[:, 0]
                self.sh[nk:nk + ng] = meta[:, 1]
                self.ty[nk:nk + ng] = meta[:, 2]
                self.mask[nk:nk + ng] = meta[:, 3]
            self.migrated += int(order.numel())
            n = self.n = nk + ng
        # 2. ghosts, direction by direction
        plan = []
        xl = self.x[:n]
        for dz in (-1, 0, 1):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    s = (dx, dy, dz)
                    if s == (0, 0, 0):
                        continue
                    nc = self.coord + np.array(s)
                    shift = np.zeros(3)
                    ok = True
                    for d in range(3):
                        if nc[d] < 0 or nc[d] >= self.grid[d]:
                            if not self.per[d]:
                                ok = False
                                break
                            shift[d] = -s[d] * self.len[d]
                            nc[d] %= self.grid[d]
                    if not ok:
                        continue
                    sel = torch.ones(n, dtype=torch.bool, device=self.dev)
                    for d in range(3):
                        if s[d] == 1:
                            sel &= xl[:, d] >= self.bhi[d] - self.cut
                        elif s[d] == -1:
                            sel &= xl[:, d] < self.blo[d] + self.cut
                    peer = int((nc[0] * self.grid[1] + nc[1]) * self.grid[2] + nc[2])
                    plan.append(dict(code=_dir_code(*s), peer=peer, send_idx=torch.nonzero(sel).flatten(),
                                     shift=torch.tensor(shift, dtype=torch.float64, device=self.dev)))
        # how many rows come from each direction: the sender's count of the opposite direction
        nsend = torch.tensor([p["send_idx"].numel() for p in plan], dtype=torch.int64, device=self.dev)
        nrecv = torch.zeros_like(nsend)
        remote = [k for k, p in enumerate(plan) if p["peer"] != self.rank]
        # a message sent with direction code c arrives at the peer as its direction 26 - c
        by_code = {p["code"]: k for k, p in enumerate(plan)}
        sends = sorted(remote, key=lambda k: (plan[k]["peer"], plan[k]["code"]))
        recvs = sorted(remote, key=lambda k: (plan[k]["peer"], 26 - plan[k]["code"]))
        self._p2p([(plan[k]["peer"], nsend[k:k + 1]) for k in sends], [(plan[k]["peer"], nrecv[k:k + 1]) for k in recvs])
        for k, p in enumerate(plan):
            if p["peer"] == self.rank:                       # own periodic image: arrives as the opposite direction
                nrecv[by_code[26 - p["code"]]] = nsend[k]
        nrecv = nrecv.cpu().tolist()
        # ghost rows: first what the remote peers send, peer by peer in the order they send it (so that the rows of one
        # peer are contiguous and travel as ONE message per array), then the rank's own periodic images
        off = n
        for k in recvs + [k for k, p in enumerate(plan) if p["peer"] == self.rank]:
            plan[k]["recv"] = (off, off + nrecv[k])
            off += nrecv[k]
        if off > self.nmax:
            raise MemoryError(f"rank {self.rank}: {n} owned + {off - n} ghost rows exceed the capacity {self.nmax}")
        self.nghost = off - n
        self.plan = plan
        self._send_order, self._recv_order, self._by_code = sends, recvs, by_code
        # fused gather / scatter lists: one index_select (+ one row-wise shift) and one index_add per array and step,
        # whatever the number of directions; the messages are slices of the gathered buffer
        z = torch.zeros(0, dtype=torch.int64, device=self.dev)
        self._send_all = torch.cat([plan[k]["send_idx"] for k in sends]) if sends else z
        self._send_shift = (torch.cat([plan[k]["shift"].expand(plan[k]["send_idx"].numel(), 3) for k in sends])
                            if sends else torch.zeros(0, 3, dtype=torch.float64, device=self.dev))
        self._send_slices, o = {}, 0
        for k in sends:
            m = plan[k]["send_idx"].numel()
            self._send_slices[k] = (o, o + m)
            o += m
        selfs = [k for k, p in enumerate(plan) if p["peer"] == self.rank]
        self._self_src = torch.cat([plan[k]["send_idx"] for k in selfs]) if selfs else z
        self._self_dst = (torch.cat([torch.arange(*plan[by_code[26 - plan[k]["code"]]]["recv"], device=self.dev) for k in selfs])
                          if selfs else z)
        self._self_shift = (torch.cat([plan[k]["shift"].expand(plan[k]["send_idx"].numel(), 3) for k in selfs])
                            if selfs else torch.zeros(0, 3, dtype=torch.float64, device=self.dev))
        self._rf = torch.empty(self._send_all.numel(), 3, dtype=torch.float64, device=self.dev)
        self._rt = torch.empty_like(self._rf)
        # one message per peer and array: [peer, (first, last) row of the gathered send buffer, (first, last) ghost row]
        self._peer_msgs = []
        for peer in sorted({plan[k]["peer"] for k in sends}):
            ks = [k for k in sends if plan[k]["peer"] == peer]
            kr = [k for k in recvs if plan[k]["peer"] == peer]
            self._peer_msgs.append((peer, (self._send_slices[ks[0]][0], self._send_slices[ks[-1]][1]),
                                    (plan[kr[0]]["recv"][0], plan[kr[-1]]["recv"][1])))
        # static per-ghost data, then positions
        self._exchange_rows([self.tag, self.sh, self.ty], shift=False)
        self.forward()
        # 3. half list with global ids
        self.npairs = self.sp.neighbor_build_device(n, self.nghost, self.x.data_ptr(), self.sh.data_ptr(), tag=self.tag.data_ptr())
        self.builds += 1

    def _exchange_rows(self, arrays, shift):
        """owners' rows of `arrays` -> the ghost rows: one gather per array, one batched group for all directions."""
        plan = self.plan
        bufs = []
        for a in arrays:
            g = a.index_select(0, self._send_all)
            if shift and a is self.x:
                g += self._send_shift
            bufs.append(g)
        sends = [(peer, b[so[0]:so[1]]) for peer, so, _ in self._peer_msgs for b in bufs]
        recvs = [(peer, a[ro[0]:ro[1]]) for peer, _, ro in self._peer_msgs for a in arrays]
        self._p2p(sends, recvs)
        if self._self_src.numel():                      # own periodic images
            for a in arrays:
                g = a.index_select(0, self._self_src)
                if shift and a is self.x:
                    g += self._self_shift
                a[self._self_dst] = g

    # ---- per step ------------------------------------------------------------------------------------------
    def forward(self):
        self._exchange_rows([self.x, self.q], shift=True)

    def reverse(self):
        plan = self.plan
        # ghost rows travel back: what came in through a direction returns to that peer, which adds it to the rows it
        # sent.  The peer sends in ITS receive order (peer, 26 - code ascending) = our send order seen from there.
        sends = [(peer, a[ro[0]:ro[1]]) for peer, _, ro in self._peer_msgs for a in (self.f, self.tq)]
        recvs = [(peer, b[so[0]:so[1]]) for peer, so, _ in self._peer_msgs for b in (self._rf, self._rt)]
        self._p2p(sends, recvs)
        if self._send_all.numel():
            self.f.index_add_(0, self._send_all, self._rf)
            self.tq.index_add_(0, self._send_all, self._rt)
        if self._self_src.numel():
            self.f.index_add_(0, self._self_src, self.f.index_select(0, self._self_dst))
            self.tq.index_add_(0, self._self_src, self.tq.index_select(0, self._self_dst))

    def force(self, eflag=False):
        sp, n = self.sp, self.n
        self.f.zero_()
        self.tq.zero_()
        if eflag:
            self.ev.zero_()
        self.forward()
        sp.compute_device(n, self.nghost, self.x.data_ptr(), self.q.data_ptr(), self.ty.data_ptr(), self.sh.data_ptr(),
                          self.f.data_ptr(), self.tq.data_ptr(), eflag=eflag, ev=self.ev.data_ptr() if eflag else None)
        self.reverse()
        if self.body and n:
            sp.post_force_device(n, self.g, self.gamma_t, self.gamma_r, self.v.data_ptr(), self.q.data_ptr(), self.L.data_ptr(),
                                 self.sh.data_ptr(), self.mask.data_ptr(), self.f.data_ptr(), self.tq.data_ptr(),
                                 groupbit=self.groupbit)

    def _nve(self, phase):
        if self.n:
            self.sp.nve_device(phase, self.n, self.dt, self.x.data_ptr(), self.v.data_ptr(), self.q.data_ptr(), self.L.data_ptr(),
                               self.f.data_ptr(), self.tq.data_ptr(), self.sh.data_ptr(), self.mask.data_ptr(),
                               groupbit=self.groupbit)

    def step(self, eflag=False):
        self._nve(0)
        if (self.steps + 1) % self.check_every == 0:
            moved = self.sp.neighbor_check_device(self.n, self.x.data_ptr()) if self.n else False
            flag = torch.tensor([1.0 if moved else 0.0], device="cpu" if self.staged else self.dev)
            if self.world > 1:
                self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            if flag.item() > 0:
                self.rebuild()
        self.force(eflag)
        self._nve(1)
        self.steps += 1

    def run(self, nsteps):
        for _ in range(nsteps):
            self.step()

    def owned(self):
        """(tag, x, v, quat) of the owned atoms on the host, sorted by tag."""
        n = self.n
        t = self.tag[:n].cpu().numpy()
        o = np.argsort(t)
        return t[o], self.x[:n].cpu().numpy()[o], self.v[:n].cpu().numpy()[o], self.q[:n].cpu().numpy()[o]
