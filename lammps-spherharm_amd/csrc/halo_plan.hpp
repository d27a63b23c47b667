// halo_plan.hpp — the decisions of the domain decomposition (include/shhalo.h), written once for host and device:
// which brick owns a position, which neighbours need a row as a ghost, how the messages are laid out.
// The host functions shhalo_plan_* (halo_plan.cpp, testable without a GPU) and the kernels of halo_kernels.hpp
// call the SAME inline functions, so the device plan equals the host plan by construction.
// Reference: LAMMPS Comm::exchange / Comm::borders of the fork are ABSENT FROM MOUNT.
#pragma once
#include "../../include/shhalo.h"
#include "sh_const.hpp"   // SHP_HD

namespace shp {

struct HaloGeom {
  double lo[3], hi[3], len[3], blo[3], bhi[3], blen[3], cut;
  int grid[3], coord[3], periodic[3];
  int rank;
  int peer[27];
  double shift[27][3];
};

inline HaloGeom halo_geom_of(const shhalo_geometry& g)
{
  HaloGeom h;
  for (int d = 0; d < 3; ++d) {
    h.lo[d] = g.lo[d]; h.hi[d] = g.hi[d]; h.len[d] = g.hi[d] - g.lo[d];
    h.blo[d] = g.blo[d]; h.bhi[d] = g.bhi[d]; h.blen[d] = (g.hi[d] - g.lo[d]) / g.grid[d];
    h.grid[d] = g.grid[d]; h.coord[d] = g.coord[d]; h.periodic[d] = g.periodic[d];
  }
  h.cut = g.cut;
  h.rank = g.rank;
  for (int c = 0; c < 27; ++c) {
    h.peer[c] = g.peer[c];
    for (int d = 0; d < 3; ++d) h.shift[c][d] = g.shift[c][d];
  }
  return h;
}

SHP_HD inline void halo_dir(int code, int s[3])
{
  s[0] = code % 3 - 1;
  s[1] = (code / 3) % 3 - 1;
  s[2] = code / 9 - 1;
}

// Domain::pbc for one coordinate.  Bounded: a coordinate further than 1e6 box lengths away (or NaN) is left alone.
SHP_HD inline double halo_wrap(const HaloGeom& g, int d, double p)
{
  if (!g.periodic[d] || (p >= g.lo[d] && p < g.hi[d])) return p;
  const double k = __builtin_floor((p - g.lo[d]) / g.len[d]);
  if (!(__builtin_fabs(k) < 1e6)) return p;
  p -= k * g.len[d];
  if (p < g.lo[d]) p += g.len[d];     // rounding at the faces
  if (p >= g.hi[d]) p -= g.len[d];
  if (p < g.lo[d]) p = g.lo[d];
  return p;
}

// brick coordinate along d of a (wrapped) position
SHP_HD inline int halo_brick_coord(const HaloGeom& g, int d, double p)
{
  const double t = (p - g.lo[d]) / g.blen[d];
  int k = (t > 0.0) ? (int)__builtin_fmin(t, 2.0e9) : 0;   // NaN -> 0
  if (k > g.grid[d] - 1) k = g.grid[d] - 1;
  return k;
}

SHP_HD inline int halo_rank_of(const HaloGeom& g, const int c[3]) { return (c[0] * g.grid[1] + c[1]) * g.grid[2] + c[2]; }

SHP_HD inline int halo_owner(const HaloGeom& g, const double x[3])
{
  int c[3];
  for (int d = 0; d < 3; ++d) c[d] = halo_brick_coord(g, d, x[d]);
  return halo_rank_of(g, c);
}

// Direction code (0..26, 13 = stays) that leads from this rank's brick to the brick owning x, or -1 if that brick
// is not one of the 26 neighbours.  Where a decomposed periodic dimension has only two bricks both steps lead to
// the same peer; the step is then +1.
SHP_HD inline int halo_dest_code(const HaloGeom& g, const double x[3])
{
  int code = 0, mul = 1;
  for (int d = 0; d < 3; ++d) {
    const int c = halo_brick_coord(g, d, x[d]);
    int s = c - g.coord[d];
    if (g.periodic[d] && g.grid[d] > 1) {
      if (s == g.grid[d] - 1) s = -1;
      else if (s == -(g.grid[d] - 1)) s = 1;
      if (g.grid[d] == 2 && s != 0) s = 1;
    }
    if (s < -1 || s > 1) return -1;
    code += (s + 1) * mul;
    mul *= 3;
  }
  return code;
}

SHP_HD inline unsigned halo_ghost_mask(const HaloGeom& g, const double x[3])
{
  unsigned m = 0u;
  // per dimension: may the row go up (+1) / down (-1)?
  bool up[3], dn[3];
  for (int d = 0; d < 3; ++d) {
    up[d] = x[d] >= g.bhi[d] - g.cut;
    dn[d] = x[d] < g.blo[d] + g.cut;
  }
  for (int code = 0; code < 27; ++code) {
    if (code == 13 || g.peer[code] < 0) continue;
    int s[3];
    halo_dir(code, s);
    bool ok = true;
    for (int d = 0; d < 3; ++d) {
      if (s[d] == 1 && !up[d]) ok = false;
      if (s[d] == -1 && !dn[d]) ok = false;
    }
    if (ok) m |= 1u << code;
  }
  return m;
}

}  // namespace shp
