/*
 * cpu_tuned.c — a TUNED CPU implementation of the sharp-rule `pair_style sh` contact path, as the CPU baseline of
 * bench.py (`cpu_baseline`, kind "port", variant "tuned").
 *
 * BASELINE ONLY: used by bench.py's cpu_baseline leg and by tests/test_cpu_tuned.py, never by the product path and
 * never as the checker (the checker is the plain oracle, oracle/shpair_oracle.c, which this file is itself held to,
 * <= 1e-12 relative, by tests/test_cpu_tuned.py).  It is NOT the reference's PairSH: the reference mount holds
 * no source (/root/reference/README.md:1), so this is the build's own CPU restatement of docs/SPEC.md §2, written
 * the way one would write it for speed on a CPU:
 *   - the recurrence constants alpha_nm, beta_nm, Pi_m^m and the scaled coefficients are tabulated once per shape
 *     (the oracle recomputes two square roots and a division per (n, m) term per evaluation);
 *   - the radius (and gradient) is evaluated for BLOCKS of directions, structure-of-arrays, the node index innermost
 *     (`#pragma omp simd`: AVX2 / AVX-512 lanes are cap nodes), for particle i over all Q nodes of the cap, for
 *     particle j over the compacted nodes inside B_j, and inside the root finder over the still-active nodes in
 *     lock step;
 *   - OpenMP over the rows of the half list.
 * Same algorithm and same evaluation points as the oracle: direct body-frame evaluation of both particles (no
 * cap-frame ring tables — that is the GPU kernel's reformulation).
 *
 * Build: gcc -O3 -march=native -fopenmp -fno-math-errno -shared -fPIC bench/cpu_tuned.c -o libcpu_tuned.so -lm
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define T_PI 3.14159265358979323846264338327950288
#define T_NEIGHMASK 0x1FFFFFFF
#define T_MAXL 20
#define T_BLK 64 /* directions per block: every work array of a block stays in L1 */

typedef struct {
  int L;
  double rmax;
  /* m-major: entry (m, n) at off[m] + (n - m) */
  int off[T_MAXL + 2];
  double a[(T_MAXL + 1) * (T_MAXL + 2) / 2]; /* alpha_nm (n > m) */
  double b[(T_MAXL + 1) * (T_MAXL + 2) / 2]; /* beta_nm  (n >= m + 2) */
  double cr[(T_MAXL + 1) * (T_MAXL + 2) / 2]; /* (2 - delta_m0) Re a_nm, times Pi_m^m for n == m */
  double ci[(T_MAXL + 1) * (T_MAXL + 2) / 2];
  double pmm[T_MAXL + 1];
} tshape;

static void tshape_build(tshape *s, int L, const double *anm, double rmax)
{
  s->L = L;
  s->rmax = rmax;
  int o = 0;
  double pmm = sqrt(1.0 / (4.0 * T_PI));
  for (int m = 0; m <= L; ++m) {
    if (m > 0) pmm = -pmm * sqrt((2.0 * m + 1.0) / (2.0 * m));
    s->pmm[m] = pmm;
    s->off[m] = o;
    const double fac = (m == 0) ? 1.0 : 2.0;
    for (int n = m; n <= L; ++n, ++o) {
      const int k = n * (n + 1) / 2 + m;
      s->a[o] = (n > m) ? sqrt((4.0 * n * n - 1.0) / ((double)n * n - (double)m * m)) : 0.0;
      s->b[o] = (n - m >= 2) ? sqrt(((2.0 * n + 1.0) * (n + m - 1.0) * (n - m - 1.0)) /
                                    ((double)(n - m) * (n + m) * (2.0 * n - 3.0)))
                             : 0.0;
      s->cr[o] = fac * anm[2 * k];
      s->ci[o] = fac * anm[2 * k + 1];
    }
  }
  s->off[L + 1] = o;
}

/* r (and, if gx != NULL, the Cartesian gradient of the polynomial F, docs/SPEC.md §1) for n <= T_BLK unit directions. */
static void tsh_eval_block(const tshape *s, int n, const double *restrict x, const double *restrict y,
                           const double *restrict z, double *restrict r, double *restrict gx, double *restrict gy,
                           double *restrict gz)
{
  double Cm[T_BLK], Sm[T_BLK], Cp[T_BLK], Sp[T_BLK], p1[T_BLK], p2[T_BLK], d1[T_BLK], d2[T_BLK];
  double Wr[T_BLK], Wi[T_BLK], Zr[T_BLK], Zi[T_BLK];
  const int grad = gx != NULL;
  for (int i = 0; i < n; ++i) {
    Cm[i] = 1.0; Sm[i] = 0.0; Cp[i] = 0.0; Sp[i] = 0.0; r[i] = 0.0;
  }
  if (grad)
    for (int i = 0; i < n; ++i) gx[i] = gy[i] = gz[i] = 0.0;
  for (int m = 0; m <= s->L; ++m) {
    const int o = s->off[m];
    const double pmm = s->pmm[m], c0r = s->cr[o] * pmm, c0i = s->ci[o] * pmm;
    for (int i = 0; i < n; ++i) {
      p2[i] = 0.0; p1[i] = pmm; d2[i] = 0.0; d1[i] = 0.0;
      Wr[i] = c0r; Wi[i] = c0i; Zr[i] = 0.0; Zi[i] = 0.0;
    }
    for (int nn = m + 1; nn <= s->L; ++nn) {
      const double a = s->a[o + nn - m], b = s->b[o + nn - m], cr = s->cr[o + nn - m], ci = s->ci[o + nn - m];
      if (grad) {
#pragma omp simd
        for (int i = 0; i < n; ++i) {
          const double p = a * z[i] * p1[i] - b * p2[i];
          const double dp = a * (p1[i] + z[i] * d1[i]) - b * d2[i];
          Wr[i] += cr * p; Wi[i] += ci * p; Zr[i] += cr * dp; Zi[i] += ci * dp;
          p2[i] = p1[i]; p1[i] = p; d2[i] = d1[i]; d1[i] = dp;
        }
      } else {
#pragma omp simd
        for (int i = 0; i < n; ++i) {
          const double p = a * z[i] * p1[i] - b * p2[i];
          Wr[i] += cr * p; Wi[i] += ci * p;
          p2[i] = p1[i]; p1[i] = p;
        }
      }
    }
    const double dm = (double)m;
#pragma omp simd
    for (int i = 0; i < n; ++i) {
      r[i] += Wr[i] * Cm[i] - Wi[i] * Sm[i];
      if (grad) {
        gz[i] += Zr[i] * Cm[i] - Zi[i] * Sm[i];
        gx[i] += dm * (Wr[i] * Cp[i] - Wi[i] * Sp[i]);
        gy[i] += dm * (-Wr[i] * Sp[i] - Wi[i] * Cp[i]);
      }
      const double c = Cm[i] * x[i] - Sm[i] * y[i], sn = Cm[i] * y[i] + Sm[i] * x[i];
      Cp[i] = Cm[i]; Sp[i] = Sm[i];
      Cm[i] = c; Sm[i] = sn;
    }
  }
}

static void tsh_eval(const tshape *s, int n, const double *x, const double *y, const double *z, double *r, double *gx,
                     double *gy, double *gz)
{
  for (int b = 0; b < n; b += T_BLK) {
    const int m = (n - b < T_BLK) ? n - b : T_BLK;
    tsh_eval_block(s, m, x + b, y + b, z + b, r + b, gx ? gx + b : NULL, gy ? gy + b : NULL, gz ? gz + b : NULL);
  }
}

static void gauss_legendre(int n, double *t, double *w)
{
  for (int i = 0; i < (n + 1) / 2; ++i) {
    double xx = cos(T_PI * (i + 0.75) / (n + 0.5)), pp = 1.0;
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

static void quat_to_mat(const double q[4], double R[3][3])
{
  const double w2 = q[0] * q[0], i2 = q[1] * q[1], j2 = q[2] * q[2], k2 = q[3] * q[3];
  const double twoij = 2.0 * q[1] * q[2], twoik = 2.0 * q[1] * q[3], twojk = 2.0 * q[2] * q[3];
  const double twoiw = 2.0 * q[1] * q[0], twojw = 2.0 * q[2] * q[0], twokw = 2.0 * q[3] * q[0];
  R[0][0] = w2 + i2 - j2 - k2; R[0][1] = twoij - twokw;     R[0][2] = twojw + twoik;
  R[1][0] = twoij + twokw;     R[1][1] = w2 - i2 + j2 - k2; R[1][2] = twojk - twoiw;
  R[2][0] = twoik - twojw;     R[2][1] = twojk + twoiw;     R[2][2] = w2 - i2 - j2 + k2;
}

/* per-thread work space for one pair: Q nodes */
typedef struct {
  int cap;
  double *buf;
} twork;

static double *wk(twork *w, int Q, int k) { return w->buf + (size_t)k * Q; }

/* One pair (docs/SPEC.md §2, sharp rule). out: V, S_n[3], T_n[3]. Returns 1 for a contact pair. */
static int tpair(const tshape *si, const tshape *sj, const double xi[3], const double qi[4], const double xj[3],
                 const double qj[4], int nq, const double *glt, const double *glw, const double *cps, const double *sps,
                 int need_volume, twork *W, double out[7])
{
  for (int a = 0; a < 7; ++a) out[a] = 0.0;
  const double Ri = si->rmax, Rj = sj->rmax;
  const double d[3] = {xj[0] - xi[0], xj[1] - xi[1], xj[2] - xi[2]};
  const double rho2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2], rho = sqrt(rho2);
  if (rho >= Ri + Rj) return 0;
  if (!(rho > 0.0)) return 0;   /* docs/SPEC.md §2 step 1: coincident centres (no line of centres): nothing */
  double cosa;
  if (rho <= Rj) cosa = -1.0;
  else if (rho2 - Rj * Rj <= Ri * Ri) cosa = sqrt(rho2 - Rj * Rj) / rho;
  else cosa = (rho2 + Ri * Ri - Rj * Rj) / (2.0 * rho * Ri);
  const double c[3] = {d[0] / rho, d[1] / rho, d[2] / rho};
  const double sg = copysign(1.0, c[2]);
  const double aa = -1.0 / (sg + c[2]);
  const double bb = c[0] * c[1] * aa;
  const double e1[3] = {1.0 + sg * c[0] * c[0] * aa, sg * bb, -sg * c[0]};
  const double e2[3] = {bb, sg + c[1] * c[1] * aa, -c[1]};
  double Rmi[3][3], Rmj[3][3];
  quat_to_mat(qi, Rmi);
  quat_to_mat(qj, Rmj);
  double dj[3];
  for (int a = 0; a < 3; ++a) dj[a] = Rmj[0][a] * d[0] + Rmj[1][a] * d[1] + Rmj[2][a] * d[2];

  const int npsi = 2 * nq, Q = nq * npsi;
  double *ux = wk(W, Q, 0), *uy = wk(W, Q, 1), *uz = wk(W, Q, 2);          /* node directions, space frame */
  double *bx = wk(W, Q, 3), *by = wk(W, Q, 4), *bz = wk(W, Q, 5);          /* ... in i's body frame */
  double *ri = wk(W, Q, 6), *gx = wk(W, Q, 7), *gy = wk(W, Q, 8), *gz = wk(W, Q, 9);
  double *qx = wk(W, Q, 10), *qy = wk(W, Q, 11), *qz = wk(W, Q, 12), *ss = wk(W, Q, 13), *rj = wk(W, Q, 14);
  double *om = wk(W, Q, 15);
  int *idx = (int *)wk(W, Q, 16); /* compaction lists (two ints per double slot) */
  const double hw = 0.5 * (1.0 - cosa), hm = 0.5 * (1.0 + cosa);
  for (int k = 0; k < nq; ++k) {
    const double mu = hm + hw * glt[k];
    const double sig = sqrt(fmax(0.0, 1.0 - mu * mu));
    const double omk = hw * glw[k] * (2.0 * T_PI / npsi);
#pragma omp simd
    for (int l = 0; l < npsi; ++l) {
      const int p = k * npsi + l;
      const double a1 = sig * cps[l], a2 = sig * sps[l];
      const double u0 = a1 * e1[0] + a2 * e2[0] + mu * c[0], u1 = a1 * e1[1] + a2 * e2[1] + mu * c[1],
                   u2 = a1 * e1[2] + a2 * e2[2] + mu * c[2];
      ux[p] = u0; uy[p] = u1; uz[p] = u2;
      bx[p] = Rmi[0][0] * u0 + Rmi[1][0] * u1 + Rmi[2][0] * u2;
      by[p] = Rmi[0][1] * u0 + Rmi[1][1] * u1 + Rmi[2][1] * u2;
      bz[p] = Rmi[0][2] * u0 + Rmi[1][2] * u1 + Rmi[2][2] * u2;
      om[p] = omk;
    }
  }
  /* particle i: radius and gradient at every node */
  tsh_eval(si, Q, bx, by, bz, ri, gx, gy, gz);
  /* surface point seen from x_j in j's frame; the nodes inside B_j are compacted */
  int nc = 0;
  for (int p = 0; p < Q; ++p) {
    const double ps0 = ri[p] * ux[p] - d[0], ps1 = ri[p] * uy[p] - d[1], ps2 = ri[p] * uz[p] - d[2];
    const double q0 = Rmj[0][0] * ps0 + Rmj[1][0] * ps1 + Rmj[2][0] * ps2;
    const double q1 = Rmj[0][1] * ps0 + Rmj[1][1] * ps1 + Rmj[2][1] * ps2;
    const double q2 = Rmj[0][2] * ps0 + Rmj[1][2] * ps1 + Rmj[2][2] * ps2;
    const double s = sqrt(q0 * q0 + q1 * q1 + q2 * q2);
    if (s >= Rj) continue;
    idx[nc] = p;
    ss[nc] = s;
    if (s > 0.0) { qx[nc] = q0 / s; qy[nc] = q1 / s; qz[nc] = q2 / s; }
    else { qx[nc] = 0.0; qy[nc] = 0.0; qz[nc] = 1.0; }
    ++nc;
  }
  if (nc == 0) return 1;
  tsh_eval(sj, nc, qx, qy, qz, rj, NULL, NULL, NULL);
  /* inside nodes: surface integrals, and the list for the volume */
  int *ins = idx + Q;
  int ni = 0;
  for (int k = 0; k < nc; ++k) {
    double rj0 = rj[k];
    if (ss[k] > 0.0) {
      if (!(ss[k] < rj0)) continue;
    } else rj0 = Rj;
    const int p = idx[k];
    const double r = ri[p];
    const double ug = bx[p] * gx[p] + by[p] * gy[p] + bz[p] * gz[p];
    const double Ab0 = r * r * bx[p] - r * (gx[p] - ug * bx[p]), Ab1 = r * r * by[p] - r * (gy[p] - ug * by[p]),
                 Ab2 = r * r * bz[p] - r * (gz[p] - ug * bz[p]);
    const double A0 = Rmi[0][0] * Ab0 + Rmi[0][1] * Ab1 + Rmi[0][2] * Ab2;
    const double A1 = Rmi[1][0] * Ab0 + Rmi[1][1] * Ab1 + Rmi[1][2] * Ab2;
    const double A2 = Rmi[2][0] * Ab0 + Rmi[2][1] * Ab1 + Rmi[2][2] * Ab2;
    const double p0 = r * ux[p], p1 = r * uy[p], p2 = r * uz[p];
    const double w = om[p];
    out[1] += w * A0; out[2] += w * A1; out[3] += w * A2;
    out[4] += w * (p1 * A2 - p2 * A1); out[5] += w * (p2 * A0 - p0 * A2); out[6] += w * (p0 * A1 - p1 * A0);
    /* keep what the root finder needs, compacted in place (ni <= k) */
    ins[ni] = p;
    ss[ni] = ss[k];
    rj[ni] = rj0;
    ++ni;
  }
  if (!need_volume || ni == 0) return 1;

  /* SPEC §2.6: inner radius of every inside node, the active ones in lock step */
  int centre_inside = 0;
  if (rho < Rj) {
    double zx = -dj[0] / rho, zy = -dj[1] / rho, zz = -dj[2] / rho, rc;
    tsh_eval_block(sj, 1, &zx, &zy, &zz, &rc, NULL, NULL, NULL);
    centre_inside = (rho - rc <= 0.0);
  }
  if (centre_inside) {
    for (int k = 0; k < ni; ++k) {
      const int p = ins[k];
      out[0] += om[p] * ri[p] * ri[p] * ri[p] / 3.0;
    }
    return 1;
  }
  /* per inside node: direction in j's frame (re-using the i-frame arrays, which are done with), bracket and the
   * three most recent points */
  double *vjx = bx, *vjy = by, *vjz = bz;              /* u in j's frame */
  double *lo = gx, *hi = gy, *lam = gz, *xa = wk(W, Q, 17), *ga = wk(W, Q, 18), *xb = wk(W, Q, 19), *gb = wk(W, Q, 20);
  double *rin = wk(W, Q, 21), *ex = wk(W, Q, 22), *ey = wk(W, Q, 23), *ez = wk(W, Q, 24), *er = wk(W, Q, 25), *es = wk(W, Q, 26);
  int *act = (int *)wk(W, Q, 27);
  for (int k = 0; k < ni; ++k) {
    const int p = ins[k];
    const double u0 = ux[p], u1 = uy[p], u2 = uz[p];
    vjx[k] = Rmj[0][0] * u0 + Rmj[1][0] * u1 + Rmj[2][0] * u2;
    vjy[k] = Rmj[0][1] * u0 + Rmj[1][1] * u1 + Rmj[2][1] * u2;
    vjz[k] = Rmj[0][2] * u0 + Rmj[1][2] * u1 + Rmj[2][2] * u2;
  }
  /* NOTE: vj* alias bx.., whose entries at ins[k] >= k were read above for index p = ins[k] only through ux/uy/uz
   * (space frame), so the in-place fill is safe. */
  int nact = 0;
  for (int k = 0; k < ni; ++k) {
    const int p = ins[k];
    const double bp = ux[p] * d[0] + uy[p] * d[1] + uz[p] * d[2];
    const double rj0 = rj[k], r = ri[p];
    double l0 = 0.0;
    if (!(rho < Rj)) l0 = bp - sqrt(fmax(0.0, bp * bp - (rho2 - Rj * Rj)));
    double lm = bp - sqrt(fmax(0.0, bp * bp - (rho2 - rj0 * rj0)));
    if (!(lm > l0 && lm < r)) lm = 0.5 * (l0 + r);
    lo[k] = l0; hi[k] = r; lam[k] = lm;
    xa[k] = r; ga[k] = ss[k] - rj0; xb[k] = r; gb[k] = ss[k] - rj0;
    rin[k] = lm;
    act[nact++] = k;
  }
  for (int it = 0; it < 60 && nact > 0; ++it) {
    for (int a = 0; a < nact; ++a) {
      const int k = act[a];
      const double y0 = lam[k] * vjx[k] - dj[0], y1 = lam[k] * vjy[k] - dj[1], y2 = lam[k] * vjz[k] - dj[2];
      const double s = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
      es[a] = s;
      if (s > 0.0) { ex[a] = y0 / s; ey[a] = y1 / s; ez[a] = y2 / s; }
      else { ex[a] = 0.0; ey[a] = 0.0; ez[a] = 1.0; }
    }
    tsh_eval(sj, nact, ex, ey, ez, er, NULL, NULL, NULL);
    int nn = 0;
    const int have3 = (it >= 1);
    for (int a = 0; a < nact; ++a) {
      const int k = act[a];
      const double gl = (es[a] == 0.0) ? -Rj : es[a] - er[a];
      const double lm = lam[k];
      if (gl >= 0.0) lo[k] = lm; else hi[k] = lm;
      const double sec = lm - gl * (lm - xb[k]) / (gl - gb[k]);
      double ext = sec;
      if (have3)
        ext = xa[k] * gb[k] * gl / ((ga[k] - gb[k]) * (ga[k] - gl)) + xb[k] * ga[k] * gl / ((gb[k] - ga[k]) * (gb[k] - gl)) +
              lm * ga[k] * gb[k] / ((gl - ga[k]) * (gl - gb[k]));
      if (!(fabs(ext) <= 1e300)) ext = sec;
      if (fabs(gl) <= (have3 ? 1e-4 : 1e-7) * Rj) {
        rin[k] = (fabs(ext) <= 1e300) ? fmin(fmax(ext, lo[k]), hi[k]) : lm;
        continue;
      }
      double nxt = ext;
      if (!(nxt > lo[k] && nxt < hi[k])) nxt = sec;
      if (!(nxt > lo[k] && nxt < hi[k])) nxt = 0.5 * (lo[k] + hi[k]);
      if (hi[k] - lo[k] <= 1e-14 * Rj) {
        rin[k] = 0.5 * (lo[k] + hi[k]);
        continue;
      }
      rin[k] = nxt;
      xa[k] = xb[k]; ga[k] = gb[k]; xb[k] = lm; gb[k] = gl; lam[k] = nxt;
      act[nn++] = k;
    }
    nact = nn;
  }
  for (int k = 0; k < ni; ++k) {
    const int p = ins[k];
    const double r = ri[p];
    out[0] += om[p] * (r * r * r - rin[k] * rin[k] * rin[k]) / 3.0;
  }
  return 1;
}

/*
 * A bed: same argument layout as the oracle's sho_compute (shapes packed, kn / expo (ntypes+1)^2 row-major, CSR half
 * list), sharp rule, forces and torques ADDED to f / torque; energy into *energy (nullable) when eflag.
 * counts (nullable): [0] candidate pairs, [1] contact pairs, [2] touching pairs.
 */
void sht_compute(int nshapes, const int *lmax, const int *anm_off, const double *anm_all, const double *rmax, int ntypes,
                 const double *kn, const double *expo, int nq, int nlocal, const double *x, const double *quat,
                 const int *type, const int *shtype, int inum, const int *ilist, const int *offs, const int *jlist,
                 int newton_pair, int eflag, int force_volume, double *f, double *torque, double *energy, long long *counts,
                 int nthreads)
{
  tshape *S = (tshape *)malloc(sizeof(tshape) * (size_t)nshapes);
  for (int s = 0; s < nshapes; ++s) tshape_build(&S[s], lmax[s], anm_all + anm_off[s], rmax[s]);
  double *glt = (double *)malloc(sizeof(double) * (size_t)(2 * nq + 4 * nq));
  double *glw = glt + nq, *cps = glw + nq, *sps = cps + 2 * nq;
  gauss_legendre(nq, glt, glw);
  for (int l = 0; l < 2 * nq; ++l) {
    const double psi = 2.0 * T_PI * (l + 0.5) / (2 * nq);
    cps[l] = cos(psi);
    sps[l] = sin(psi);
  }
  int any_nonunit = 0;
  for (int a = 1; a <= ntypes; ++a)
    for (int b = 1; b <= ntypes; ++b)
      if (expo[a * (ntypes + 1) + b] != 1.0) any_nonunit = 1;
  const int need_volume = force_volume || eflag || any_nonunit;
  const int Q = 2 * nq * nq;
  long long c0 = 0, c1 = 0, c2 = 0;
  double etot = 0.0;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : c0, c1, c2, etot)
  {
    twork W;
    W.cap = Q;
    W.buf = (double *)malloc(sizeof(double) * (size_t)Q * 28);
#pragma omp for schedule(dynamic, 16)
    for (int ii = 0; ii < inum; ++ii) {
      const int i = ilist[ii];
      for (int p = offs[ii]; p < offs[ii + 1]; ++p) {
        const int j = jlist[p] & T_NEIGHMASK;
        ++c0;
        double out[7];
        if (!tpair(&S[shtype[i]], &S[shtype[j]], x + 3 * i, quat + 4 * i, x + 3 * j, quat + 4 * j, nq, glt, glw, cps, sps,
                   need_volume, &W, out))
          continue;
        ++c1;
        const double V = out[0];
        const int touched = need_volume ? (V > 0.0) : (out[1] != 0.0 || out[2] != 0.0 || out[3] != 0.0);
        if (!touched) continue;
        ++c2;
        const double k_ = kn[type[i] * (ntypes + 1) + type[j]], m_ = expo[type[i] * (ntypes + 1) + type[j]];
        const double vm1 = (m_ == 1.0) ? 1.0 : pow(V, m_ - 1.0);
        const double pn = k_ * m_ * vm1;
        const double F[3] = {-pn * out[1], -pn * out[2], -pn * out[3]};
        const double M[3] = {-pn * out[4], -pn * out[5], -pn * out[6]};
        const double d[3] = {x[3 * j] - x[3 * i], x[3 * j + 1] - x[3 * i + 1], x[3 * j + 2] - x[3 * i + 2]};
        for (int a = 0; a < 3; ++a) {
#pragma omp atomic
          f[3 * i + a] += F[a];
#pragma omp atomic
          torque[3 * i + a] += M[a];
        }
        if (newton_pair || j < nlocal) {
          const double G[3] = {-F[0], -F[1], -F[2]};
          const double tj[3] = {-M[0] - (d[1] * G[2] - d[2] * G[1]), -M[1] - (d[2] * G[0] - d[0] * G[2]),
                                -M[2] - (d[0] * G[1] - d[1] * G[0])};
          for (int a = 0; a < 3; ++a) {
#pragma omp atomic
            f[3 * j + a] += G[a];
#pragma omp atomic
            torque[3 * j + a] += tj[a];
          }
        }
        if (eflag) {
          const double share = newton_pair ? 1.0 : (0.5 + (j < nlocal ? 0.5 : 0.0));
          etot += share * k_ * (vm1 * V);
        }
      }
    }
    free(W.buf);
  }
  if (energy && eflag) *energy += etot;
  if (counts) { counts[0] = c0; counts[1] = c1; counts[2] = c2; }
  free(glt);
  free(S);
}
