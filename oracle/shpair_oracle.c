/*
 * shpair_oracle.c — CPU restatement of the `pair_style sh` contact path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * build, load or call it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED.  The reference mount holds a single line of README
 * (/root/reference/README.md:1) and no source, tests or fixtures, so no
 * function below can cite a reference file:line.  This file restates
 * docs/SPEC.md (section numbers in the comments) in plain scalar C and is
 * pinned by closed-form and library known answers in tests/test_oracle_*.py
 * (scipy sph_harm_y, numpy leggauss, sphere-sphere lens volume and cap area,
 * finite-difference gradients, brute-force ray marching, energy derivative).
 *
 * Plain loops, runtime L, no tables shared with the HIP path.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SHO_PI 3.14159265358979323846264338327950288
#define SHO_NEIGHMASK 0x1FFFFFFF
#define SHO_MAX_NQ 256

/* ---------------------------------------------------------------- SPEC §1 */

/* r(u) and, if grad != NULL, the Cartesian gradient of the polynomial F. */
double sho_sh_eval(int L, const double *anm, const double u[3], double *grad)
{
  const double x = u[0], y = u[1], z = u[2];
  double Cm = 1.0, Sm = 0.0, Cp = 0.0, Sp = 0.0;
  double r = 0.0, gx = 0.0, gy = 0.0, gz = 0.0;
  double pmm = sqrt(1.0 / (4.0 * SHO_PI));
  for (int m = 0; m <= L; ++m) {
    if (m > 0) pmm = -pmm * sqrt((2.0 * m + 1.0) / (2.0 * m));
    const double fac = (m == 0) ? 1.0 : 2.0;
    double p2 = 0.0, p1 = pmm, d2 = 0.0, d1 = 0.0;
    int k = m * (m + 1) / 2 + m;
    double Wr = fac * anm[2 * k] * p1, Wi = fac * anm[2 * k + 1] * p1;
    double Zr = 0.0, Zi = 0.0;
    for (int n = m + 1; n <= L; ++n) {
      const double a = sqrt((4.0 * n * n - 1.0) / ((double)n * n - (double)m * m));
      double b = 0.0;
      if (n - m >= 2)
        b = sqrt(((2.0 * n + 1.0) * (n + m - 1.0) * (n - m - 1.0)) /
                 ((double)(n - m) * (n + m) * (2.0 * n - 3.0)));
      const double p = a * z * p1 - b * p2;
      const double dp = a * (p1 + z * d1) - b * d2;
      k = n * (n + 1) / 2 + m;
      Wr += fac * anm[2 * k] * p;
      Wi += fac * anm[2 * k + 1] * p;
      Zr += fac * anm[2 * k] * dp;
      Zi += fac * anm[2 * k + 1] * dp;
      p2 = p1; p1 = p; d2 = d1; d1 = dp;
    }
    r += Wr * Cm - Wi * Sm;
    gz += Zr * Cm - Zi * Sm;
    if (m > 0) {
      gx += m * (Wr * Cp - Wi * Sp);
      gy += m * (-Wr * Sp - Wi * Cp);
    }
    Cp = Cm; Sp = Sm;
    Cm = Cp * x - Sp * y;
    Sm = Cp * y + Sp * x;
  }
  if (grad) { grad[0] = gx; grad[1] = gy; grad[2] = gz; }
  return r;
}

/* Gauss-Legendre nodes/weights on [-1,1], ascending. Newton on P_n. */
void sho_gauss_legendre(int n, double *t, double *w)
{
  for (int i = 0; i < (n + 1) / 2; ++i) {
    double xx = cos(SHO_PI * (i + 0.75) / (n + 0.5));
    double pp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p0 = 1.0, p1 = xx;
      for (int k = 2; k <= n; ++k) {
        const double pk = ((2.0 * k - 1.0) * xx * p1 - (k - 1.0) * p0) / k;
        p0 = p1; p1 = pk;
      }
      if (n == 1) { p0 = 1.0; p1 = xx; }
      pp = n * (xx * p1 - p0) / (xx * xx - 1.0);
      const double dx = p1 / pp;
      xx -= dx;
      if (fabs(dx) < 1e-16) break;
    }
    /* re-evaluate the derivative at the converged node */
    {
      double p0 = 1.0, p1 = xx;
      for (int k = 2; k <= n; ++k) {
        const double pk = ((2.0 * k - 1.0) * xx * p1 - (k - 1.0) * p0) / k;
        p0 = p1; p1 = pk;
      }
      pp = n * (xx * p1 - p0) / (xx * xx - 1.0);
    }
    t[i] = -xx; t[n - 1 - i] = xx;
    w[i] = w[n - 1 - i] = 2.0 / ((1.0 - xx * xx) * pp * pp);
  }
  if (n % 2 == 1) t[n / 2] = 0.0;
}

/* default bounding radius, SPEC §1 */
double sho_shape_rmax(int L, const double *anm)
{
  const int nt = 6 * (L + 1) + 2, np = 2 * nt;
  double *t = (double *)malloc(sizeof(double) * 2 * nt), *w = t + nt;
  sho_gauss_legendre(nt, t, w);
  double best = 0.0;
  for (int a = 0; a < nt; ++a) {
    const double ct = t[a], st = sqrt(1.0 - ct * ct);
    for (int b = 0; b < np; ++b) {
      const double ph = 2.0 * SHO_PI * b / np;
      const double u[3] = { st * cos(ph), st * sin(ph), ct };
      const double r = sho_sh_eval(L, anm, u, NULL);
      if (r > best) best = r;
    }
  }
  free(t);
  return 1.01 * best;
}

/* ---------------------------------------------------------------- helpers */

static void quat_to_mat(const double q[4], double R[3][3])
{
  const double w2 = q[0] * q[0], i2 = q[1] * q[1], j2 = q[2] * q[2], k2 = q[3] * q[3];
  const double twoij = 2.0 * q[1] * q[2], twoik = 2.0 * q[1] * q[3], twojk = 2.0 * q[2] * q[3];
  const double twoiw = 2.0 * q[1] * q[0], twojw = 2.0 * q[2] * q[0], twokw = 2.0 * q[3] * q[0];
  R[0][0] = w2 + i2 - j2 - k2; R[0][1] = twoij - twokw;     R[0][2] = twojw + twoik;
  R[1][0] = twoij + twokw;     R[1][1] = w2 - i2 + j2 - k2; R[1][2] = twojk - twoiw;
  R[2][0] = twoik - twojw;     R[2][1] = twojk + twoiw;     R[2][2] = w2 - i2 - j2 + k2;
}
static void matvec(const double R[3][3], const double v[3], double o[3])
{ for (int a = 0; a < 3; ++a) o[a] = R[a][0] * v[0] + R[a][1] * v[1] + R[a][2] * v[2]; }
static void tmatvec(const double R[3][3], const double v[3], double o[3])
{ for (int a = 0; a < 3; ++a) o[a] = R[0][a] * v[0] + R[1][a] * v[1] + R[2][a] * v[2]; }
static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double a[3], const double b[3], double o[3])
{ o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }

/* g(lambda) of SPEC §2.6 in j's body frame */
static double g_ray(int Lj, const double *anmj, double Rj, const double uj[3], const double dj[3], double lam)
{
  const double q[3] = { lam * uj[0] - dj[0], lam * uj[1] - dj[1], lam * uj[2] - dj[2] };
  const double s = sqrt(dot3(q, q));
  if (s == 0.0) return -Rj;
  const double qh[3] = { q[0] / s, q[1] / s, q[2] / s };
  return s - sho_sh_eval(Lj, anmj, qh, NULL);
}

/* ---------------------------------------------------------------- SPEC §2 */

/*
 * One pair. out[0]=V, out[1..3]=S_n, out[4..6]=T_n (space frame, T_n about x_i).
 * diag (nullable): [0]=#inside nodes, [1]=#nodes not culled by B_j, [2]=sum of
 * root-finder evaluations, [3]=cos(alpha).
 * Returns 1 if the bounding spheres overlap (a contact pair), else 0.
 */
/* SPEC §2.6: inner radius of the ray x_i + lambda u for a node whose surface point (lambda = ri) lies inside j. */
static double inner_radius(int Lj, const double *anmj, double Rj, const double Rmj[3][3], const double u[3],
                           const double d[3], const double dj[3], double rho, double rho2, double ri, double s,
                           double rj0, double *diag)
{
  double uj[3];
  tmatvec(Rmj, u, uj);
  const double bp = dot3(u, d);
  double lo = 0.0;
  if (!(rho < Rj)) lo = bp - sqrt(fmax(0.0, bp * bp - (rho2 - Rj * Rj)));
  double hi = ri;
  double lam = bp - sqrt(fmax(0.0, bp * bp - (rho2 - rj0 * rj0)));
  if (!(lam > lo && lam < hi)) lam = 0.5 * (lo + hi);
  /* three most recent points: (xa,ga) oldest, (xb,gb), (lam,gl) newest */
  double xa = ri, ga = s - rj0, xb = ri, gb = s - rj0;
  double rin = lam;
  for (int it = 0; it < 60; ++it) {
    const double gl = g_ray(Lj, anmj, Rj, uj, dj, lam);
    if (diag) diag[2] += 1.0;
    if (gl >= 0.0) lo = lam; else hi = lam;
    const int have3 = (it >= 1);
    const double sec = lam - gl * (lam - xb) / (gl - gb);
    double ext = sec;  /* extrapolation through all known points */
    if (have3)
      ext = xa * gb * gl / ((ga - gb) * (ga - gl)) + xb * ga * gl / ((gb - ga) * (gb - gl)) +
            lam * ga * gb / ((gl - ga) * (gl - gb));
    if (!(fabs(ext) <= 1e300)) ext = sec;
    if (fabs(gl) <= (have3 ? 1e-4 : 1e-7) * Rj) {  /* accept the extrapolated point */
      rin = (fabs(ext) <= 1e300) ? fmin(fmax(ext, lo), hi) : lam;
      break;
    }
    double nxt = ext;
    if (!(nxt > lo && nxt < hi)) nxt = sec;
    if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
    if (hi - lo <= 1e-14 * Rj) { rin = 0.5 * (lo + hi); break; }
    rin = nxt;
    xa = xb; ga = gb; xb = lam; gb = gl; lam = nxt;
  }
  return rin;
}

/* SPEC §2.8: 0 = sharp inside test (default), 1 = covered-fraction weights. Applies to the next sho_pair /
 * sho_compute calls. */
static int g_rule = 0;
void sho_set_rule(int rule) { g_rule = rule ? 1 : 0; }

static int sho_pair_weighted(int Li, const double *anmi, double Ri, int Lj, const double *anmj, double Rj,
                             const double xi[3], const double qi[4], const double xj[3], const double qj[4],
                             int nq, int need_volume, double out[7], double diag[4]);

int sho_pair(int Li, const double *anmi, double Ri, int Lj, const double *anmj, double Rj,
             const double xi[3], const double qi[4], const double xj[3], const double qj[4],
             int nq, int need_volume, double out[7], double diag[4])
{
  if (g_rule) return sho_pair_weighted(Li, anmi, Ri, Lj, anmj, Rj, xi, qi, xj, qj, nq, need_volume, out, diag);
  for (int a = 0; a < 7; ++a) out[a] = 0.0;
  if (diag) for (int a = 0; a < 4; ++a) diag[a] = 0.0;
  const double d[3] = { xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2] };
  const double rho2 = dot3(d, d), rho = sqrt(rho2);
  if (rho >= Ri + Rj) return 0;
  if (!(rho > 0.0)) return 0;   /* docs/SPEC.md §2 step 1: coincident centres (no line of centres): nothing */

  double cosa;
  if (rho <= Rj) cosa = -1.0;
  else if (rho2 - Rj * Rj <= Ri * Ri) cosa = sqrt(rho2 - Rj * Rj) / rho;
  else cosa = (rho2 + Ri * Ri - Rj * Rj) / (2.0 * rho * Ri);
  if (diag) diag[3] = cosa;

  const double c[3] = { d[0] / rho, d[1] / rho, d[2] / rho };
  const double sg = copysign(1.0, c[2]);
  const double aa = -1.0 / (sg + c[2]);
  const double bb = c[0] * c[1] * aa;
  const double e1[3] = { 1.0 + sg * c[0] * c[0] * aa, sg * bb, -sg * c[0] };
  const double e2[3] = { bb, sg + c[1] * c[1] * aa, -c[1] };

  double Rmi[3][3], Rmj[3][3];
  quat_to_mat(qi, Rmi);
  quat_to_mat(qj, Rmj);
  double dj[3];
  tmatvec(Rmj, d, dj);

  /* centre of i inside j? (only possible when rho < Rj) */
  int centre_inside = 0;
  if (rho < Rj) {
    const double zero[3] = { 0.0, 0.0, 0.0 };
    centre_inside = (g_ray(Lj, anmj, Rj, zero, dj, 0.0) <= 0.0);
  }

  double t[SHO_MAX_NQ], w[SHO_MAX_NQ];
  sho_gauss_legendre(nq, t, w);
  const int npsi = 2 * nq;
  const double hw = 0.5 * (1.0 - cosa), hm = 0.5 * (1.0 + cosa);

  for (int k = 0; k < nq; ++k) {
    const double mu = hm + hw * t[k];
    const double sig = sqrt(fmax(0.0, 1.0 - mu * mu));
    const double om = hw * w[k] * (2.0 * SHO_PI / npsi);
    for (int l = 0; l < npsi; ++l) {
      const double psi = 2.0 * SHO_PI * (l + 0.5) / npsi;
      const double cp = cos(psi), sp = sin(psi);
      double u[3];
      for (int a = 0; a < 3; ++a) u[a] = sig * (cp * e1[a] + sp * e2[a]) + mu * c[a];
      double ui[3], gi[3];
      tmatvec(Rmi, u, ui);
      const double ri = sho_sh_eval(Li, anmi, ui, gi);
      /* surface point relative to x_j, in j's frame */
      const double ps[3] = { ri * u[0] - d[0], ri * u[1] - d[1], ri * u[2] - d[2] };
      double q[3];
      tmatvec(Rmj, ps, q);
      const double s = sqrt(dot3(q, q));
      if (s >= Rj) continue;
      if (diag) diag[1] += 1.0;
      double rj0 = Rj;
      if (s > 0.0) {
        const double qh[3] = { q[0] / s, q[1] / s, q[2] / s };
        rj0 = sho_sh_eval(Lj, anmj, qh, NULL);
        if (!(s < rj0)) continue;
      }
      if (diag) diag[0] += 1.0;
      /* vector area element, body frame then space frame */
      const double ug = dot3(ui, gi);
      double Ab[3], A[3];
      for (int a = 0; a < 3; ++a) Ab[a] = ri * ri * ui[a] - ri * (gi[a] - ug * ui[a]);
      matvec(Rmi, Ab, A);
      const double pr[3] = { ri * u[0], ri * u[1], ri * u[2] };
      double pxA[3];
      cross3(pr, A, pxA);
      for (int a = 0; a < 3; ++a) { out[1 + a] += om * A[a]; out[4 + a] += om * pxA[a]; }
      if (!need_volume) continue;

      /* SPEC §2.6 inner radius */
      double rin = 0.0;
      if (!centre_inside) rin = inner_radius(Lj, anmj, Rj, Rmj, u, d, dj, rho, rho2, ri, s, rj0, diag);
      out[0] += om * (ri * ri * ri - rin * rin * rin) / 3.0;
    }
  }
  return 1;
}

/* SPEC §2.8: the same pair with covered-fraction weights. Two passes over the Q nodes: residuals, then
 * weights and contributions. */
static int sho_pair_weighted(int Li, const double *anmi, double Ri, int Lj, const double *anmj, double Rj,
                             const double xi[3], const double qi[4], const double xj[3], const double qj[4],
                             int nq, int need_volume, double out[7], double diag[4])
{
  for (int a = 0; a < 7; ++a) out[a] = 0.0;
  if (diag) for (int a = 0; a < 4; ++a) diag[a] = 0.0;
  const double d[3] = { xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2] };
  const double rho2 = dot3(d, d), rho = sqrt(rho2);
  if (rho >= Ri + Rj) return 0;
  if (!(rho > 0.0)) return 0;   /* docs/SPEC.md §2 step 1: coincident centres (no line of centres): nothing */
  double cosa;
  if (rho <= Rj) cosa = -1.0;
  else if (rho2 - Rj * Rj <= Ri * Ri) cosa = sqrt(rho2 - Rj * Rj) / rho;
  else cosa = (rho2 + Ri * Ri - Rj * Rj) / (2.0 * rho * Ri);
  if (diag) diag[3] = cosa;
  const double c[3] = { d[0] / rho, d[1] / rho, d[2] / rho };
  const double sg = copysign(1.0, c[2]);
  const double aa = -1.0 / (sg + c[2]);
  const double bb = c[0] * c[1] * aa;
  const double e1[3] = { 1.0 + sg * c[0] * c[0] * aa, sg * bb, -sg * c[0] };
  const double e2[3] = { bb, sg + c[1] * c[1] * aa, -c[1] };
  double Rmi[3][3], Rmj[3][3];
  quat_to_mat(qi, Rmi);
  quat_to_mat(qj, Rmj);
  double dj[3];
  tmatvec(Rmj, d, dj);
  int centre_inside = 0;
  if (rho < Rj) {
    const double zero[3] = { 0.0, 0.0, 0.0 };
    centre_inside = (g_ray(Lj, anmj, Rj, zero, dj, 0.0) <= 0.0);
  }
  double t[SHO_MAX_NQ], w[SHO_MAX_NQ];
  sho_gauss_legendre(nq, t, w);
  const int npsi = 2 * nq, Q = nq * npsi;
  const double hw = 0.5 * (1.0 - cosa), hm = 0.5 * (1.0 + cosa);
  /* per node: g~, s, rj0, ri, u[3], gradient gi[3] (body frame of i), inB */
  double *G = malloc(sizeof(double) * Q * 11);
  if (!G) return 0;
  double *S = G + Q, *RJ0 = G + 2 * Q, *RI = G + 3 * Q, *UU = G + 4 * Q, *GI = G + 7 * Q, *INB = G + 10 * Q;
  for (int k = 0; k < nq; ++k) {
    const double mu = hm + hw * t[k];
    const double sig = sqrt(fmax(0.0, 1.0 - mu * mu));
    for (int l = 0; l < npsi; ++l) {
      const int p = k * npsi + l;
      const double psi = 2.0 * SHO_PI * (l + 0.5) / npsi;
      const double cp = cos(psi), sp = sin(psi);
      double u[3], ui[3], gi[3];
      for (int a = 0; a < 3; ++a) u[a] = sig * (cp * e1[a] + sp * e2[a]) + mu * c[a];
      tmatvec(Rmi, u, ui);
      const double ri = sho_sh_eval(Li, anmi, ui, gi);
      const double ps[3] = { ri * u[0] - d[0], ri * u[1] - d[1], ri * u[2] - d[2] };
      double q[3];
      tmatvec(Rmj, ps, q);
      const double s = sqrt(dot3(q, q));
      double rj0 = Rj, g;
      if (s >= Rj) {
        g = s - Rj;
        INB[p] = 0.0;
      } else {
        if (diag) diag[1] += 1.0;
        if (s > 0.0) {
          const double qh[3] = { q[0] / s, q[1] / s, q[2] / s };
          rj0 = sho_sh_eval(Lj, anmj, qh, NULL);
        }
        g = s - rj0;
        INB[p] = 1.0;
      }
      G[p] = g; S[p] = s; RJ0[p] = rj0; RI[p] = ri;
      for (int a = 0; a < 3; ++a) { UU[3 * p + a] = u[a]; GI[3 * p + a] = gi[a]; }
    }
  }
  for (int k = 0; k < nq; ++k) {
    const double om = hw * w[k] * (2.0 * SHO_PI / npsi);
    for (int l = 0; l < npsi; ++l) {
      const int p = k * npsi + l;
      if (INB[p] == 0.0) continue;
      const double g = G[p];
      const double Dl = 0.5 * fabs(G[k * npsi + (l + 1) % npsi] - G[k * npsi + (l + npsi - 1) % npsi]);
      double Dk = 0.0;
      if (nq > 1) Dk = (k < nq - 1) ? fabs(G[p + npsi] - g) : fabs(g - G[p - npsi]);
      const double den = Dk + Dl;
      double wt;
      if (den > 0.0) wt = fmin(1.0, fmax(0.0, 0.5 - g / den));
      else wt = (g < 0.0) ? 1.0 : 0.0;
      if (!(wt > 0.0)) continue;
      if (diag) diag[0] += 1.0;
      const double ri = RI[p], *u = UU + 3 * p, *gi = GI + 3 * p;
      double ui[3];
      tmatvec(Rmi, u, ui);
      const double ug = dot3(ui, gi);
      double Ab[3], A[3];
      for (int a = 0; a < 3; ++a) Ab[a] = ri * ri * ui[a] - ri * (gi[a] - ug * ui[a]);
      matvec(Rmi, Ab, A);
      const double pr[3] = { ri * u[0], ri * u[1], ri * u[2] };
      double pxA[3];
      cross3(pr, A, pxA);
      const double ow = om * wt;
      for (int a = 0; a < 3; ++a) { out[1 + a] += ow * A[a]; out[4 + a] += ow * pxA[a]; }
      if (!need_volume || !(g < 0.0)) continue;
      double rin = 0.0;
      if (!centre_inside) rin = inner_radius(Lj, anmj, Rj, Rmj, u, d, dj, rho, rho2, ri, S[p], RJ0[p], diag);
      out[0] += om * (ri * ri * ri - rin * rin * rin) / 3.0;  /* unweighted: the integrand vanishes at the boundary */
    }
  }
  free(G);
  return 1;
}

/* ------------------------------------------------- SPEC §2.7 + §3: a bed */

/*
 * Shapes are packed: lmax[s], anm_off[s] (offset in doubles into anm_all),
 * rmax[s].  kn/expo are (ntypes+1)^2 row-major, 1-based types.
 * Neighbour list in CSR form: ilist[inum], offs[inum+1], jlist[] (bits above
 * NEIGHMASK are stripped).  f and torque are ADDED to.
 * eng_virial (nullable): [0]=energy, [1..6]=virial xx,yy,zz,xy,xz,yz.
 * counts (nullable): [0]=candidate pairs, [1]=contact pairs (bounding
 * spheres overlap), [2]=pairs with V>0 / any inside node.
 * pair_out (nullable): 7 doubles per CSR slot (V,S_n,T_n).
 */
/* Per-atom tallies of the NEXT sho_compute calls (LAMMPS ev_tally_xyz per-atom part, SPEC §2.7): eatom[nall],
 * vatom[nall][6], added to; NULL switches them off. */
static double *g_eatom = NULL, *g_vatom = NULL;
void sho_set_peratom(double *eatom, double *vatom)
{
  g_eatom = eatom;
  g_vatom = vatom;
}

int sho_compute(int nshape, const int *lmax, const int *anm_off, const double *anm_all,
                const double *rmax, int ntypes, const double *kn, const double *expo, int nq,
                int nlocal, const double *x, const double *quat, const int *type, const int *shtype,
                int inum, const int *ilist, const int *offs, const int *jlist, int newton_pair,
                int eflag, int vflag, int force_volume, double *f, double *torque, double *eng_virial,
                long long *counts, double *pair_out, int nthreads)
{
  (void)nshape;
  long long ncand = 0, ncontact = 0, ntouch = 0;
  double ev[7] = { 0, 0, 0, 0, 0, 0, 0 };
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#else
  (void)nthreads;
#endif
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : ncand, ncontact, ntouch, ev[:7])
  for (int ii = 0; ii < inum; ++ii) {
    const int i = ilist[ii];
    const int si = shtype[i];
    for (int p = offs[ii]; p < offs[ii + 1]; ++p) {
      const int j = jlist[p] & SHO_NEIGHMASK;
      const int sj = shtype[j];
      const double knij = kn[type[i] * (ntypes + 1) + type[j]];
      const double mij = expo[type[i] * (ntypes + 1) + type[j]];
      const int needV = force_volume || eflag || (mij != 1.0) || (g_eatom != NULL);
      double o[7];
      ++ncand;
      const int hit = sho_pair(lmax[si], anm_all + anm_off[si], rmax[si], lmax[sj], anm_all + anm_off[sj],
                               rmax[sj], x + 3 * i, quat + 4 * i, x + 3 * j, quat + 4 * j, nq, needV, o, NULL);
      if (pair_out) memcpy(pair_out + 7 * (size_t)p, o, sizeof(o));
      if (!hit) continue;
      ++ncontact;
      const double V = o[0];
      const int touched = needV ? (V > 0.0) : (o[1] != 0.0 || o[2] != 0.0 || o[3] != 0.0);
      if (!touched) continue;
      ++ntouch;
      const double pn = (mij == 1.0) ? knij : knij * mij * pow(V, mij - 1.0);
      const double Fi[3] = { -pn * o[1], -pn * o[2], -pn * o[3] };
      const double Ti[3] = { -pn * o[4], -pn * o[5], -pn * o[6] };
      const double d[3] = { x[3 * j] - x[3 * i], x[3 * j + 1] - x[3 * i + 1], x[3 * j + 2] - x[3 * i + 2] };
      for (int a = 0; a < 3; ++a) {
#pragma omp atomic
        f[3 * i + a] += Fi[a];
#pragma omp atomic
        torque[3 * i + a] += Ti[a];
      }
      const int applyj = newton_pair || j < nlocal;
      if (applyj) {
        const double Fj[3] = { -Fi[0], -Fi[1], -Fi[2] };
        double dxF[3];
        cross3(d, Fj, dxF);
        for (int a = 0; a < 3; ++a) {
#pragma omp atomic
          f[3 * j + a] += Fj[a];
#pragma omp atomic
          torque[3 * j + a] += -Ti[a] - dxF[a];
        }
      }
      if (g_eatom || g_vatom) {
        const int owni = newton_pair || i < nlocal;
        if (g_eatom) {
          const double eh = 0.5 * knij * pow(V, mij);
          if (owni) {
#pragma omp atomic
            g_eatom[i] += eh;
          }
          if (applyj) {
#pragma omp atomic
            g_eatom[j] += eh;
          }
        }
        if (g_vatom) {
          const double del[3] = { -d[0], -d[1], -d[2] };
          const double v[6] = { 0.5 * del[0] * Fi[0], 0.5 * del[1] * Fi[1], 0.5 * del[2] * Fi[2],
                                0.5 * del[0] * Fi[1], 0.5 * del[0] * Fi[2], 0.5 * del[1] * Fi[2] };
          for (int a = 0; a < 6; ++a) {
            if (owni) {
#pragma omp atomic
              g_vatom[6 * (size_t)i + a] += v[a];
            }
            if (applyj) {
#pragma omp atomic
              g_vatom[6 * (size_t)j + a] += v[a];
            }
          }
        }
      }
      if (eflag || vflag) {
        const double share = newton_pair ? 1.0 : (0.5 + (j < nlocal ? 0.5 : 0.0));
        if (eflag) ev[0] += share * knij * pow(V, mij);
        if (vflag) {
          const double del[3] = { -d[0], -d[1], -d[2] };
          ev[1] += share * del[0] * Fi[0]; ev[2] += share * del[1] * Fi[1]; ev[3] += share * del[2] * Fi[2];
          ev[4] += share * del[0] * Fi[1]; ev[5] += share * del[0] * Fi[2]; ev[6] += share * del[1] * Fi[2];
        }
      }
    }
  }
  if (eng_virial) for (int a = 0; a < 7; ++a) eng_virial[a] = ev[a];
  if (counts) { counts[0] = ncand; counts[1] = ncontact; counts[2] = ntouch; }
  return 0;
}

int sho_max_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
