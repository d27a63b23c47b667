/* c_halo_plan.c — include/shhalo.h from plain C99 (-pedantic), no GPU needed: the pure host planner of the N > 1
 * path.  Prints, for rank 5 of a 2 x 2 x 2 decomposition of a box periodic in x and y, its brick, its
 * distinct peers (7: one per xGMI link of an 8-GPU node) and the message layout for made-up per-direction counts.
 * Exit 0 on success; used by tests/test_c_example.py. */
#include <stdio.h>
#include <string.h>

#include "shhalo.h"

int main(void)
{
  int grid[3];
  const double lo[3] = {0.0, 0.0, 0.0}, hi[3] = {20.0, 20.0, 20.0};
  const int periodic[3] = {1, 1, 0};
  shhalo_geometry g;
  shhalo_layout L;
  int send[27], recv[27], c, k, seen[8], npeer = 0;
  double x[6] = {21.0, 3.0, 4.0, 9.99, 19.0, -1.0};
  int owner[2];

  if (shhalo_proc_grid(8, grid) != SHPAIR_OK || grid[0] != 2 || grid[1] != 2 || grid[2] != 2) return 1;
  if (shhalo_plan_geometry(grid, lo, hi, periodic, 2.5, 5, &g) != SHPAIR_OK) return 2;
  printf("rank %d of %d: brick [%g,%g) x [%g,%g) x [%g,%g), ghost cutoff %g\n", g.rank, g.nranks, g.blo[0], g.bhi[0], g.blo[1],
         g.bhi[1], g.blo[2], g.bhi[2], g.cut);
  memset(seen, 0, sizeof(seen));
  for (c = 0; c < 27; ++c)
    if (g.peer[c] >= 0 && g.peer[c] != g.rank && !seen[g.peer[c]]) {
      seen[g.peer[c]] = 1;
      ++npeer;
    }
  printf("distinct peers: %d\n", npeer);
  if (npeer != 7) return 3;
  /* a brick shorter than the cutoff must be refused */
  if (shhalo_plan_geometry(grid, lo, hi, periodic, 11.0, 5, &g) == SHPAIR_OK) return 4;
  if (shhalo_plan_geometry(grid, lo, hi, periodic, 2.5, 5, &g) != SHPAIR_OK) return 2;
  /* ownership: wrapped in x (periodic), clamped in z (open) */
  if (shhalo_plan_owner(&g, 2, x, owner) != SHPAIR_OK) return 5;
  printf("owners: %d %d (x wrapped to %g)\n", owner[0], owner[1], x[0]);
  if (owner[0] != 0 || owner[1] != 2 || x[0] != 1.0) return 6;
  for (c = 0; c < 27; ++c) {
    send[c] = (c != 13 && g.peer[c] >= 0) ? 10 + c : 0;
    recv[c] = (c != 13 && g.peer[c] >= 0) ? 40 - c : 0;
  }
  if (shhalo_plan_layout(&g, send, recv, &L) != SHPAIR_OK) return 7;
  printf("layout: %d send rows, %d ghost rows, %d messages each way\n", L.nsend, L.nghost, L.npeers);
  for (k = 0; k < L.npeers; ++k)
    printf("  peer %d: send rows [%d,%d)  ghost rows [%d,%d)\n", L.peer_rank[k], L.peer_send_off[k],
           L.peer_send_off[k] + L.peer_send_cnt[k], L.peer_recv_off[k], L.peer_recv_off[k] + L.peer_recv_cnt[k]);
  return L.npeers == 7 ? 0 : 8;
}
