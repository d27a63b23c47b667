/*
 * c_api_min.c — the C ABI used from plain C (C99), without LAMMPS: two overlapping spherical-harmonic
 * particles, one pair, forces and torques back on the host.  Build (see examples/Makefile):
 *   gcc -std=c99 -I../include c_api_min.c -L../lammps-spherharm_amd/shpair -lshpair -lm
 * Exit code 0 = ran on a GPU and Newton's third law holds; 77 = no HIP device (the library has no CPU fallback).
 */
#include <math.h>
#include <stdio.h>

#include "shpair.h"
#include "shstep.h"

int main(void)
{
  shpair_ctx *ctx = NULL;
  int rc = shpair_create(&ctx, 0);
  if (rc == SHPAIR_ENODEV) {
    printf("no HIP device: %s\n", shpair_strerror(rc));
    return 77;
  }
  if (rc) return 1;

  /* a slightly oblate shape of order 2: a_00 = sqrt(4 pi) (unit mean radius), a_20 = -0.15 */
  enum { LMAX = 2, NCOEF = (LMAX + 1) * (LMAX + 2) };
  double anm[NCOEF] = { 0 };
  anm[0] = sqrt(4.0 * 3.14159265358979323846);
  anm[2 * 3] = -0.15; /* (n, m) = (2, 0) -> k = n(n+1)/2 + m = 3 */
  double body[10];
  if (shstep_shape_mass_props(LMAX, anm, body)) return 1;

  if ((rc = shpair_settings(ctx, 12)) || (rc = shpair_set_ntypes(ctx, 1, 1)) || (rc = shpair_set_shape(ctx, 0, LMAX, anm, 0.0)) ||
      (rc = shpair_set_coeff(ctx, 1, 1, 1000.0, 1.25))) {
    printf("setup failed: %s (%s)\n", shpair_strerror(rc), shpair_last_error(ctx));
    return 1;
  }
  double rmax;
  shpair_get_rmax(ctx, 0, &rmax);

  const double x[6] = { 0, 0, 0, 1.8, 0.2, 0.1 };
  const double quat[8] = { 1, 0, 0, 0, 0.9238795325112867, 0, 0.3826834323650898, 0 }; /* j turned by 45 deg about y */
  const int type[2] = { 1, 1 }, shtype[2] = { 0, 0 };
  const int ilist[1] = { 0 }, offsets[2] = { 0, 1 }, jlist[1] = { 1 };
  double f[6] = { 0 }, torque[6] = { 0 }, eng = 0.0, virial[6] = { 0 };
  if ((rc = shpair_set_neighbors_csr(ctx, 1, ilist, offsets, jlist)) ||
      (rc = shpair_compute(ctx, 2, 0, x, quat, type, shtype, 1, 1, 1, f, torque, &eng, virial))) {
    printf("compute failed: %s (%s)\n", shpair_strerror(rc), shpair_last_error(ctx));
    return 1;
  }
  shpair_stats st;
  shpair_get_stats(ctx, &st);
  printf("%s: volume %.6f (unit density), bounding radius %.4f, energy %.6f\n", shpair_version(), body[0], rmax, eng);
  printf("F_i = (%.6f, %.6f, %.6f)  F_j = (%.6f, %.6f, %.6f)\n", f[0], f[1], f[2], f[3], f[4], f[5]);
  /* Newton's third law, and the torque balance tau_i + tau_j + d x F_j = 0 */
  const double d[3] = { x[3] - x[0], x[4] - x[1], x[5] - x[2] };
  const double bal[3] = { torque[0] + torque[3] + d[1] * f[5] - d[2] * f[4], torque[1] + torque[4] + d[2] * f[3] - d[0] * f[5],
                          torque[2] + torque[5] + d[0] * f[4] - d[1] * f[3] };
  double err = 0.0;
  for (int a = 0; a < 3; ++a) err += fabs(f[a] + f[3 + a]) + fabs(bal[a]);
  shpair_destroy(ctx);
  if (!(eng > 0.0) || err > 1e-9 * fabs(f[0])) return 1;
  return 0;
}
