// halo_plan.cpp — the pure host half of include/shhalo.h: processor grid, brick geometry, ownership, ghost masks
// and the message layout.  No HIP call in this file: these functions run (and are tested) without a GPU, and the
// device path uses the same geometry / layout code plus kernels built from the same inline decisions
// (halo_plan.hpp).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "halo_plan.hpp"

using namespace shp;

extern "C" {

int shhalo_proc_grid(int nranks, int grid[3])
{
  if (nranks < 1 || !grid) return SHPAIR_EINVAL;
  int best[3] = {nranks, 1, 1};
  long long best_key = 0;
  bool have = false;
  for (int px = 1; px <= nranks; ++px) {
    if (nranks % px) continue;
    for (int py = 1; py <= nranks / px; ++py) {
      if ((nranks / px) % py) continue;
      const int pz = nranks / px / py;
      int v[3] = {px, py, pz};
      std::sort(v, v + 3);
      // most cubic first (smallest spread), then the larger leading factors
      const long long key = (long long)(v[2] - v[0]) * 1000000LL - v[2] * 1000LL - v[1];
      if (!have || key < best_key) {
        have = true;
        best_key = key;
        best[0] = v[2]; best[1] = v[1]; best[2] = v[0];
      }
    }
  }
  grid[0] = best[0]; grid[1] = best[1]; grid[2] = best[2];
  return SHPAIR_OK;
}

int shhalo_plan_geometry(const int grid[3], const double lo[3], const double hi[3], const int periodic[3], double cut,
                         int rank, shhalo_geometry* out)
{
  if (!grid || !lo || !hi || !periodic || !out) return SHPAIR_EINVAL;
  if (!(cut > 0.0) || !std::isfinite(cut)) return SHPAIR_EINVAL;
  for (int d = 0; d < 3; ++d)
    if (grid[d] < 1 || !std::isfinite(lo[d]) || !std::isfinite(hi[d]) || !(hi[d] > lo[d])) return SHPAIR_EINVAL;
  const int nranks = grid[0] * grid[1] * grid[2];
  if (rank < 0 || rank >= nranks) return SHPAIR_EINVAL;
  shhalo_geometry g;
  std::memset(&g, 0, sizeof(g));
  g.rank = rank;
  g.nranks = nranks;
  g.cut = cut;
  g.coord[0] = rank / (grid[1] * grid[2]);
  g.coord[1] = (rank / grid[2]) % grid[1];
  g.coord[2] = rank % grid[2];
  for (int d = 0; d < 3; ++d) {
    g.grid[d] = grid[d];
    g.periodic[d] = periodic[d] ? 1 : 0;
    g.lo[d] = lo[d];
    g.hi[d] = hi[d];
    const double len = hi[d] - lo[d], blen = len / grid[d];
    g.blo[d] = (g.coord[d] == 0) ? lo[d] : lo[d] + g.coord[d] * blen;
    g.bhi[d] = (g.coord[d] == grid[d] - 1) ? hi[d] : lo[d] + (g.coord[d] + 1) * blen;
    // a ghost shell must come from the adjacent brick only / a periodic image must not meet its original
    if (grid[d] > 1 && blen < cut) return SHPAIR_EINVAL;
    if (grid[d] == 1 && g.periodic[d] && len < 2.0 * cut) return SHPAIR_EINVAL;
  }
  for (int code = 0; code < 27; ++code) {
    g.peer[code] = -1;
    if (code == 13) continue;
    int s[3], nc[3];
    halo_dir(code, s);
    bool ok = true;
    for (int d = 0; d < 3; ++d) {
      nc[d] = g.coord[d] + s[d];
      if (nc[d] < 0 || nc[d] >= grid[d]) {
        if (!g.periodic[d]) {
          ok = false;
          break;
        }
        g.shift[code][d] = -s[d] * (hi[d] - lo[d]);
        nc[d] = (nc[d] + grid[d]) % grid[d];
      }
    }
    if (!ok) {
      for (int d = 0; d < 3; ++d) g.shift[code][d] = 0.0;
      continue;
    }
    g.peer[code] = (nc[0] * grid[1] + nc[1]) * grid[2] + nc[2];
  }
  *out = g;
  return SHPAIR_OK;
}

int shhalo_plan_owner(const shhalo_geometry* g, int n, double* x, int* owner)
{
  if (!g || n < 0 || (n > 0 && (!x || !owner))) return SHPAIR_EINVAL;
  const HaloGeom h = halo_geom_of(*g);
  for (int i = 0; i < n; ++i) {
    for (int d = 0; d < 3; ++d) x[3 * i + d] = halo_wrap(h, d, x[3 * i + d]);
    owner[i] = halo_owner(h, x + 3 * i);
  }
  return SHPAIR_OK;
}

int shhalo_plan_ghost_mask(const shhalo_geometry* g, int n, const double* x, unsigned* mask)
{
  if (!g || n < 0 || (n > 0 && (!x || !mask))) return SHPAIR_EINVAL;
  const HaloGeom h = halo_geom_of(*g);
  for (int i = 0; i < n; ++i) mask[i] = halo_ghost_mask(h, x + 3 * i);
  return SHPAIR_OK;
}

int shhalo_plan_layout(const shhalo_geometry* g, const int send_cnt[27], const int recv_cnt[27], shhalo_layout* out)
{
  if (!g || !send_cnt || !recv_cnt || !out) return SHPAIR_EINVAL;
  shhalo_layout L;
  std::memset(&L, 0, sizeof(L));
  std::vector<int> dirs;
  for (int c = 0; c < 27; ++c)
    if (c != 13 && g->peer[c] >= 0) {
      if (send_cnt[c] < 0 || recv_cnt[c] < 0) return SHPAIR_EINVAL;
      dirs.push_back(c);
    }
  // send rows: by (peer, my code)
  std::vector<int> so = dirs, ro = dirs;
  std::sort(so.begin(), so.end(), [&](int a, int b) { return g->peer[a] != g->peer[b] ? g->peer[a] < g->peer[b] : a < b; });
  long long off = 0;
  for (int c : so) {
    L.send_off[c] = (int)off;
    L.send_cnt[c] = send_cnt[c];
    off += send_cnt[c];
  }
  if (off > 0x7fffffffLL) return SHPAIR_EINVAL;
  L.nsend = (int)off;
  // ghost rows: by (peer, the SENDER's code 26 - my code)
  std::sort(ro.begin(), ro.end(),
            [&](int a, int b) { return g->peer[a] != g->peer[b] ? g->peer[a] < g->peer[b] : (26 - a) < (26 - b); });
  off = 0;
  for (int c : ro) {
    L.recv_off[c] = (int)off;
    L.recv_cnt[c] = recv_cnt[c];
    off += recv_cnt[c];
  }
  if (off > 0x7fffffffLL) return SHPAIR_EINVAL;
  L.nghost = (int)off;
  // one message per distinct remote peer: its blocks are contiguous in both orders
  for (int c : so) {
    const int p = g->peer[c];
    if (p == g->rank) continue;
    int k = L.npeers - 1;
    if (k < 0 || L.peer_rank[k] != p) {
      k = L.npeers++;
      L.peer_rank[k] = p;
      L.peer_send_off[k] = L.send_off[c];
      L.peer_send_cnt[k] = 0;
      L.peer_recv_off[k] = -1;
      L.peer_recv_cnt[k] = 0;
    }
    L.peer_send_cnt[k] += send_cnt[c];
  }
  for (int c : ro) {
    const int p = g->peer[c];
    if (p == g->rank) continue;
    for (int k = 0; k < L.npeers; ++k)
      if (L.peer_rank[k] == p) {
        if (L.peer_recv_off[k] < 0) L.peer_recv_off[k] = L.recv_off[c];
        L.peer_recv_cnt[k] += recv_cnt[c];
      }
  }
  for (int k = 0; k < L.npeers; ++k)
    if (L.peer_recv_off[k] < 0) L.peer_recv_off[k] = 0;
  *out = L;
  return SHPAIR_OK;
}

}  // extern "C"
