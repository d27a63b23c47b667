/*
 * sanitize_main.c — drives every oracle entry point on small seeded inputs; built with
 * -fsanitize=address,undefined by `make sanitize` (tests/test_oracle_sanitize.py).  TEST INFRASTRUCTURE ONLY.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

double sho_sh_eval(int L, const double *anm, const double u[3], double *grad);
double sho_shape_rmax(int L, const double *anm);
int sho_pair(int Li, const double *anmi, double Ri, int Lj, const double *anmj, double Rj, const double xi[3],
             const double qi[4], const double xj[3], const double qj[4], int nq, int need_volume, double out[7],
             double diag[4]);
void sho_set_rule(int rule);
void sho_set_peratom(double *eatom, double *vatom);
int sho_compute(int nshape, const int *lmax, const int *anm_off, const double *anm_all, const double *rmax, int ntypes,
                const double *kn, const double *expo, int nq, int nlocal, const double *x, const double *quat,
                const int *type, const int *shtype, int inum, const int *ilist, const int *offs, const int *jlist,
                int newton_pair, int eflag, int vflag, int force_volume, double *f, double *torque, double *eng_virial,
                long long *counts, double *pair_out, int nthreads);
void sho_mass_props(int L, const double *anm, double out[10]);
void sho_nve(int phase, int n, double dt, const double *massprops, const double *density, double *x, double *v,
             double *quat, double *angmom, const double *f, const double *torque, const int *shtype, const int *mask,
             int groupbit);
void sho_post_force(int n, const double *massprops, const double *density, const double g[3], double gamma_t,
                    double gamma_r, const double *v, const double *quat, const double *angmom, const int *shtype,
                    const int *mask, int groupbit, double *f, double *torque);
void sho_energies(int n, const double *massprops, const double *density, const double g[3], const double *x,
                  const double *v, const double *quat, const double *angmom, const int *shtype, const int *mask,
                  int groupbit, double out[3]);
int sho_borders(int n, double *x, const double lo[3], const double hi[3], const int periodic[3], double cmax,
                int *ghost_owner, int *ghost_shift);
int sho_half_list(int nlocal, int nall, const double *x, const int *shtype, const int *tag, const double *rmax, double skin,
                  int *offsets, int *jlist);

static unsigned long long s = 88172645463325252ULL;
static double rnd(void)
{
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return (double)(s >> 11) / 9007199254740992.0;
}

int main(void)
{
  enum { L = 5, NT = (L + 1) * (L + 2), N = 64 };
  double anm[NT];
  for (int k = 0; k < NT; ++k) anm[k] = 0.03 * (rnd() - 0.5);
  anm[0] = sqrt(4.0 * 3.14159265358979323846);
  anm[1] = 0.0;
  const double R = sho_shape_rmax(L, anm);
  double x[4 * N * 3], q[4 * N * 4];
  int type[4 * N], sht[4 * N], tag[4 * N], mask[N];
  const int m = 4;
  for (int i = 0; i < N; ++i) {
    x[3 * i] = 1.8 * (i % m) + 0.2 * rnd();
    x[3 * i + 1] = 1.8 * ((i / m) % m) + 0.2 * rnd();
    x[3 * i + 2] = 1.8 * (i / (m * m)) + 0.2 * rnd();
    double nn = 0;
    for (int a = 0; a < 4; ++a) { q[4 * i + a] = rnd() - 0.5; nn += q[4 * i + a] * q[4 * i + a]; }
    for (int a = 0; a < 4; ++a) q[4 * i + a] /= sqrt(nn);
    mask[i] = 1;
  }
  for (int i = 0; i < 4 * N; ++i) { type[i] = 1; sht[i] = 0; }
  /* periodic ghosts + half list */
  const double lo[3] = { -0.1, -0.1, -0.1 }, hi[3] = { 7.3, 7.3, 7.3 };
  const int per[3] = { 1, 1, 0 };
  int *gown = malloc(sizeof(int) * 26 * N), *gsh = malloc(sizeof(int) * 78 * N);
  const int ng = sho_borders(N, x, lo, hi, per, 2 * R + 0.1, gown, gsh);
  if (N + ng > 4 * N) return 2;
  for (int g = 0; g < ng; ++g) {
    for (int d = 0; d < 3; ++d) x[3 * (N + g) + d] = x[3 * gown[g] + d] + gsh[3 * g + d] * (hi[d] - lo[d]);
    for (int a = 0; a < 4; ++a) q[4 * (N + g) + a] = q[4 * gown[g] + a];
  }
  for (int i = 0; i < N; ++i) tag[i] = i;
  for (int g = 0; g < ng; ++g) tag[N + g] = gown[g];
  int offs[N + 1];
  const int np = sho_half_list(N, N + ng, x, sht, tag, &R, 0.1, offs, NULL);
  int *jl = malloc(sizeof(int) * (np > 0 ? np : 1));
  sho_half_list(N, N + ng, x, sht, tag, &R, 0.1, offs, jl);
  int ilist[N];
  for (int i = 0; i < N; ++i) ilist[i] = i;
  /* the bed, both rules, with every optional output */
  const int lmax[1] = { L }, aoff[1] = { 0 };
  const double kn[4] = { 0, 0, 0, 800.0 }, ex[4] = { 1, 1, 1, 1.25 };
  double esum = 0.0;
  for (int rule = 0; rule < 2; ++rule) {
    double *f = calloc(3 * (N + ng), sizeof(double)), *t = calloc(3 * (N + ng), sizeof(double));
    double *ea = calloc(N + ng, sizeof(double)), *va = calloc(6 * (N + ng), sizeof(double));
    double *po = calloc(7 * (np > 0 ? np : 1), sizeof(double));
    double ev[7];
    long long counts[3];
    sho_set_rule(rule);
    sho_set_peratom(ea, va);
    sho_compute(1, lmax, aoff, anm, &R, 1, kn, ex, 6, N, x, q, type, sht, N, ilist, offs, jl, 1, 1, 1, 0, f, t, ev, counts, po, 2);
    sho_set_peratom(NULL, NULL);
    esum += ev[0];
    if (counts[0] != np) return 3;
    /* integrate one step with those forces */
    double mp[10], rho = 1.3, g[3] = { 0, 0, -1 }, en[3];
    sho_mass_props(L, anm, mp);
    double v[3 * N], Lm[3 * N];
    for (int k = 0; k < 3 * N; ++k) { v[k] = 0.1 * (rnd() - 0.5); Lm[k] = 0.1 * (rnd() - 0.5); }
    sho_post_force(N, mp, &rho, g, 0.1, 0.05, v, q, Lm, sht, mask, 1, f, t);
    sho_nve(0, N, 1e-3, mp, &rho, x, v, q, Lm, f, t, sht, mask, 1);
    sho_nve(1, N, 1e-3, mp, &rho, x, v, q, Lm, f, t, sht, mask, 1);
    sho_energies(N, mp, &rho, g, x, v, q, Lm, sht, mask, 1, en);
    esum += en[0] + en[1];
    free(f); free(t); free(ea); free(va); free(po);
  }
  sho_set_rule(0);
  /* isolated pairs across the cap branches, n_q 1 .. 9, both rules */
  for (int k = 0; k < 60; ++k) {
    double xi[3] = { 0, 0, 0 }, xj[3], o[7], dg[4];
    double nn = 0;
    for (int a = 0; a < 3; ++a) { xj[a] = rnd() - 0.5; nn += xj[a] * xj[a]; }
    const double rho = (0.1 + 2.0 * rnd()) * R;
    for (int a = 0; a < 3; ++a) xj[a] *= rho / sqrt(nn);
    sho_set_rule(k & 1);
    sho_pair(L, anm, R, L, anm, R, xi, q + 4 * (k % N), xj, q + 4 * ((k + 7) % N), 1 + k % 9, 1, o, dg);
    esum += o[0];
  }
  sho_set_rule(0);
  free(gown); free(gsh); free(jl);
  printf("sanitize ok %.6g\n", esum);
  return isfinite(esum) ? 0 : 4;
}
