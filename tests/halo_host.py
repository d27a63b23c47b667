"""Host rehearsal of the N > 1 path for the CPU tests: one rank's Comm::borders / forward / reverse built from the
library's PURE HOST planner (shhalo_plan_* of include/shhalo.h: geometry, ownership, ghost masks, message layout) with
numpy doing what the pack / unpack kernels do and a caller-supplied transport moving the per-peer messages; the
oracle stands in for the pair kernel (tests may call the oracle, the product path never does).

The device path (csrc/shhalo_api.hip) uses the same geometry and layout functions and kernels compiled from the
same inline decisions (csrc/halo_plan.hpp), so what passes here at world 2/4/8 is the plan the GPUs execute.
"""
import numpy as np

from shpair import mrank


def _dir_codes(geo):
    return [c for c in range(27) if c != 13 and geo.peer[c] >= 0]


class HostRank:
    """One rank: owned rows of a global bed, send lists, and (after exchange_counts + layout) its ghost rows."""

    def __init__(self, rank, grid, lo, hi, periodic, cut, x, quat, type_, shtype, tag):
        self.geo = mrank.plan_geometry(grid, lo, hi, periodic, cut, rank)
        self.rank = rank
        xw, owner = mrank.plan_owner(self.geo, x)
        mine = np.flatnonzero(owner == rank)                 # ascending global index = ascending tag
        self.n = mine.size
        self.x, self.q = xw[mine], np.ascontiguousarray(quat[mine])
        self.ty, self.sh, self.tag = type_[mine].copy(), shtype[mine].copy(), tag[mine].copy()
        mask = mrank.plan_ghost_mask(self.geo, self.x)
        self.send_list = {c: np.flatnonzero((mask >> np.uint32(c)) & np.uint32(1)) for c in _dir_codes(self.geo)}
        self.send_cnt = np.zeros(27, dtype=np.int64)
        for c, idx in self.send_list.items():
            self.send_cnt[c] = idx.size

    def count_message(self, peer):
        """The 27 counts a remote peer is sent: my count of every direction that leads to it."""
        return np.array([self.send_cnt[c] if self.geo.peer[c] == peer else 0 for c in range(27)], dtype=np.int64)

    def set_counts(self, msg_from_peer):
        """msg_from_peer: {remote peer rank: its count_message(me)}."""
        g = self.geo
        self.recv_cnt = np.zeros(27, dtype=np.int64)
        for c in _dir_codes(g):
            p = g.peer[c]
            self.recv_cnt[c] = self.send_cnt[26 - c] if p == self.rank else msg_from_peer[p][26 - c]
        self.lay = mrank.plan_layout(g, self.send_cnt, self.recv_cnt)
        self.nghost = self.lay.nghost
        # the concatenated send list in layout order
        order = sorted(_dir_codes(g), key=lambda c: self.lay.send_off[c])
        self.send_idx = np.concatenate([self.send_list[c] for c in order]) if order else np.zeros(0, np.int64)
        self.send_code = np.concatenate([np.full(self.send_list[c].size, c) for c in order]) if order else np.zeros(0, np.int64)
        assert self.send_idx.size == self.lay.nsend
        for c in order:
            if self.send_list[c].size:
                assert self.lay.send_off[c] == int(np.flatnonzero(self.send_code == c)[0])

    def peers(self):
        L = self.lay
        return [(L.peer_rank[k], L.peer_send_off[k], L.peer_send_cnt[k], L.peer_recv_off[k], L.peer_recv_cnt[k])
                for k in range(L.npeers)]

    # ---- what halo_pack_kernel / halo_unpack_kernel do --------------------------------------------------------
    def pack_forward(self, wide):
        g = self.geo
        shift = np.array([[g.shift[c][d] for d in range(3)] for c in range(27)])
        cols = [self.x[self.send_idx] + shift[self.send_code], self.q[self.send_idx]]
        if wide:
            cols += [self.tag[self.send_idx, None].astype(np.float64), self.ty[self.send_idx, None].astype(np.float64),
                     self.sh[self.send_idx, None].astype(np.float64)]
        sendbuf = np.concatenate(cols, axis=1) if self.send_idx.size else np.zeros((0, 10 if wide else 7))
        recvbuf = np.full((self.nghost, sendbuf.shape[1]), np.nan)
        for c in _dir_codes(g):                                # own periodic images: straight into the receive buffer
            if g.peer[c] == self.rank and self.send_cnt[c]:
                a = self.lay.send_off[c]
                b = self.lay.recv_off[26 - c]
                recvbuf[b:b + self.send_cnt[c]] = sendbuf[a:a + self.send_cnt[c]]
        return sendbuf, recvbuf

    def unpack_forward(self, recvbuf, wide):
        assert not np.isnan(recvbuf).any(), "a ghost row was never written"
        if wide:
            self.xa = np.concatenate([self.x, recvbuf[:, 0:3]])
            self.qa = np.concatenate([self.q, recvbuf[:, 3:7]])
            self.taga = np.concatenate([self.tag, recvbuf[:, 7].astype(np.int32)])
            self.tya = np.concatenate([self.ty, recvbuf[:, 8].astype(np.int32)])
            self.sha = np.concatenate([self.sh, recvbuf[:, 9].astype(np.int32)])
        else:
            self.xa[self.n:] = recvbuf[:, 0:3]
            self.qa[self.n:] = recvbuf[:, 3:7]

    # ---- halo_rpack_kernel / halo_runpack_kernel ------------------------------------------------------------------
    def pack_reverse(self, f, tq):
        g = self.geo
        rsend = np.concatenate([f[self.n:], tq[self.n:]], axis=1)
        rrecv = np.full((self.lay.nsend, 6), np.nan)
        for k in _dir_codes(g):                               # ghosts that are my own images
            if g.peer[k] == self.rank and self.recv_cnt[k]:
                a = self.lay.recv_off[k]
                b = self.lay.send_off[26 - k]
                rrecv[b:b + self.recv_cnt[k]] = rsend[a:a + self.recv_cnt[k]]
        return rsend, rrecv

    def unpack_reverse(self, rrecv, f, tq):
        assert not np.isnan(rrecv).any()
        np.add.at(f, self.send_idx, rrecv[:, 0:3])
        np.add.at(tq, self.send_idx, rrecv[:, 3:6])


def run_rank(hr, O, shapes, K, E, nq, skin, exchange):
    """Borders (wide forward), list, oracle compute, reverse for one rank.  exchange(sends, recvs): sends =
    [(peer, array)], recvs = [(peer, out_array)] with at most one of each per peer; blocks until the data is in."""
    rmax = [s[2] for s in shapes]
    sendbuf, recvbuf = hr.pack_forward(True)
    exchange([(p, sendbuf[so:so + sc]) for p, so, sc, ro, rc in hr.peers() if sc],
             [(p, recvbuf[ro:ro + rc]) for p, so, sc, ro, rc in hr.peers() if rc])
    hr.unpack_forward(recvbuf, True)
    offs, jl = O.half_list(hr.n, hr.xa, hr.sha, hr.taga, rmax, skin)
    o = O.compute(shapes, K, E, nq, hr.n, hr.xa, hr.qa, hr.tya, hr.sha, np.arange(hr.n, dtype=np.int32), offs, jl,
                  newton_pair=True, eflag=True)
    f, tq = o["f"].copy(), o["torque"].copy()
    rsend, rrecv = hr.pack_reverse(f, tq)
    exchange([(p, rsend[ro:ro + rc]) for p, so, sc, ro, rc in hr.peers() if rc],
             [(p, rrecv[so:so + sc]) for p, so, sc, ro, rc in hr.peers() if sc])
    hr.unpack_reverse(rrecv, f, tq)
    return f[:hr.n], tq[:hr.n], o["eng_virial"][0], int(o["counts"][0])


def single_domain_reference(O, shapes, K, E, nq, skin, lo, hi, periodic, cut, x, quat, type_, shtype, tag):
    """The same bed on one rank: periodic images by the oracle's SPEC §7 borders, brute-force half list."""
    rmax = [s[2] for s in shapes]
    xw = np.ascontiguousarray(x, dtype=np.float64).copy()
    gown, gshift = O.borders(xw, lo, hi, periodic, cut)
    n = xw.shape[0]
    ext = np.asarray(hi, float) - np.asarray(lo, float)
    xa = np.concatenate([xw, xw[gown] + gshift * ext])
    qa = np.concatenate([quat, quat[gown]])
    o_ = lambda a: np.concatenate([a, a[gown]])
    offs, jl = O.half_list(n, xa, o_(shtype), o_(tag), rmax, skin)
    o = O.compute(shapes, K, E, nq, n, xa, qa, o_(type_), o_(shtype), np.arange(n, dtype=np.int32), offs, jl, newton_pair=True,
                  eflag=True)
    f, tq = o["f"].copy(), o["torque"].copy()
    np.add.at(f, gown, f[n:])
    np.add.at(tq, gown, tq[n:])
    return f[:n], tq[:n], o["eng_virial"][0], int(o["counts"][0])
