// sh_tables.cpp — host-side table construction (docs/SPEC.md §1).
#include "sh_tables.hpp"

#include "sh_const.hpp"

#include <cmath>

namespace shp {

static const long double kPi = 3.14159265358979323846264338327950288L;

static inline int idx(int n, int m) { return n * (n + 1) / 2 + m; }

void gauss_legendre(int n, std::vector<double>& t, std::vector<double>& w)
{
  t.assign(n, 0.0);
  w.assign(n, 0.0);
  for (int i = 0; i < (n + 1) / 2; ++i) {
    long double xx = cosl(kPi * (i + 0.75L) / (n + 0.5L));
    long double pp = 1.0L;
    for (int it = 0; it < 100; ++it) {
      long double p0 = 1.0L, p1 = xx;
      for (int k = 2; k <= n; ++k) {
        const long double pk = ((2.0L * k - 1.0L) * xx * p1 - (k - 1.0L) * p0) / k;
        p0 = p1;
        p1 = pk;
      }
      pp = n * (xx * p1 - p0) / (xx * xx - 1.0L);
      const long double dx = p1 / pp;
      xx -= dx;
      if (fabsl(dx) < 1e-19L) break;
    }
    long double p0 = 1.0L, p1 = xx;
    for (int k = 2; k <= n; ++k) {
      const long double pk = ((2.0L * k - 1.0L) * xx * p1 - (k - 1.0L) * p0) / k;
      p0 = p1;
      p1 = pk;
    }
    pp = n * (xx * p1 - p0) / (xx * xx - 1.0L);
    t[i] = (double)(-xx);
    t[n - 1 - i] = (double)xx;
    w[i] = w[n - 1 - i] = (double)(2.0L / ((1.0L - xx * xx) * pp * pp));
  }
  if (n % 2 == 1) t[n / 2] = 0.0;
}

void build_recurrence(int L, std::vector<double>& rc, std::vector<double>& scale)
{
  // n-major (k = n(n+1)/2+m); values from sh_const.hpp so that the host-folded
  // scale and the constants compiled into the kernels agree bit for bit.
  // rc: n == m -> 1 (Q_m), n > m -> a'_nm.  scale: s_nm Pi_m^m.
  const int T = (L + 1) * (L + 2) / 2;
  rc.assign(T, 0.0);
  scale.assign(T, 1.0);
  for (int m = 0; m <= L; ++m) {
    rc[idx(m, m)] = 1.0;
    scale[idx(m, m)] = sh_const::coef_scale(m, m);
    for (int n = m + 1; n <= L; ++n) {
      rc[idx(n, m)] = sh_const::aprime(n, m);
      scale[idx(n, m)] = sh_const::coef_scale(n, m);
    }
  }
}

void build_coefficients(int L, int lmax, const double* anm, const std::vector<double>& rc,
                        const std::vector<double>& scale, std::vector<double>& cw)
{
  (void)rc;
  const int T = (L + 1) * (L + 2) / 2;
  cw.assign(2 * T, 0.0);
  for (int n = 0; n <= lmax && n <= L; ++n)
    for (int m = 0; m <= n; ++m) {
      const int k = idx(n, m);
      const double fac = (m == 0) ? 1.0 : 2.0;
      cw[2 * k] = fac * anm[2 * k] * scale[k];
      cw[2 * k + 1] = (m == 0) ? 0.0 : fac * anm[2 * k + 1] * scale[k];
    }
}

void build_monomial(int L, int lmax, const double* anm, std::vector<double>& wm)
{
  const int T = (L + 1) * (L + 2) / 2;
  wm.assign(2 * T, 0.0);
  for (int m = 0; m <= L; ++m) {
    const int d = L - m;
    // Pi_n^m(z), n = m..L, ascending monomial coefficients
    std::vector<long double> p2(d + 1, 0.0L), p1(d + 1, 0.0L), p(d + 1, 0.0L), wr(d + 1, 0.0L), wi(d + 1, 0.0L);
    const long double fac = (m == 0) ? 1.0L : 2.0L;
    for (int n = m; n <= L; ++n) {
      if (n == m) {
        p.assign(d + 1, 0.0L);
        p[0] = (long double)sh_const::pmm(m);
      } else {
        const long double a = sqrtl((4.0L * n * n - 1.0L) / ((long double)n * n - (long double)m * m));
        const long double b = (n - m >= 2) ? sqrtl(((2.0L * n + 1.0L) * (n + m - 1.0L) * (n - m - 1.0L)) /
                                                   ((long double)(n - m) * (n + m) * (2.0L * n - 3.0L)))
                                           : 0.0L;
        for (int k = 0; k <= d; ++k) p[k] = ((k > 0) ? a * p1[k - 1] : 0.0L) - b * p2[k];
      }
      if (n <= lmax) {
        const int kk = idx(n, m);
        for (int k = 0; k <= d; ++k) {
          wr[k] += fac * (long double)anm[2 * kk] * p[k];
          if (m > 0) wi[k] += fac * (long double)anm[2 * kk + 1] * p[k];
        }
      }
      p2 = p1;
      p1 = p;
    }
    const int base = m * (L + 1) - m * (m - 1) / 2;  // sh_moff
    for (int k = 0; k <= d; ++k) {                   // position k holds the coefficient of z^(d-k)
      wm[2 * (base + k)] = (double)wr[d - k];
      wm[2 * (base + k) + 1] = (double)wi[d - k];
    }
  }
}

// Particle j in the pair's common frame (pair_kernel.hpp, jpoly_build): with the rotated, scaled coefficients v0 (the
// same vector cap_frame_rotate leaves for particle i),
//   r_j(mu, psi) = sum_m sigma^m [cos(m psi) sum_n v0[n^2+n+m] Q_n^m(mu) + sin(m psi) sum_n v0[n^2+n-m] Q_n^m(mu)],
// and every point the kernel evaluates r_j at lies on one of the 2 n_q azimuths psi_l of the quadrature.  With
//   E_nm(mu) = (1 - mu^2)^floor(m/2) Q_n^m(mu)        (a polynomial of degree n - (m & 1), parity n - m)
// the even orders sum to a polynomial G_l(mu) of degree L and the odd ones to sigma H_l(mu), H_l of degree L - 1.
// This table is the azimuth-independent first stage as a sparse matrix in ELL form (2L + 4 polynomials of L + 1
// coefficients; the last two, of the non-existent order L + 1, are zero): row o = (2 m + part) (L + 1) + k
// (part 0: cos, 1: sin) holds the <= L/2 + 1 products E_nm[k] that make up the coefficient of mu^k of
//   PJ[2 m + part](mu) = sum_n v0[n^2 + n +- m] E_nm(mu).
// Built in long double from the exact Pi_n^m and divided by the very double coef_scale(n, m) that is folded into v0.
void build_jpoly_ell(int L, std::vector<double>& val, std::vector<int>& col)
{
  const int W = L / 2 + 1, K = L + 1, NR = (2 * L + 4) * K;   // the two rows of the order L + 1 stay empty
  val.assign((size_t)NR * W, 0.0);
  col.assign((size_t)NR * W, 0);
  std::vector<int> fill(NR, 0);
  for (int m = 0; m <= L; ++m) {
    std::vector<long double> p2(K + 2, 0.0L), p1(K + 2, 0.0L), p(K + 2, 0.0L);
    long double pmm = sqrtl(1.0L / (4.0L * kPi));
    for (int k = 1; k <= m; ++k) pmm = -pmm * sqrtl((2.0L * k + 1.0L) / (2.0L * k));
    for (int n = m; n <= L; ++n) {
      if (n == m) {
        p.assign(K + 2, 0.0L);
        p[0] = pmm;
      } else {
        const long double a = sqrtl((4.0L * n * n - 1.0L) / ((long double)n * n - (long double)m * m));
        const long double b = (n - m >= 2) ? sqrtl(((2.0L * n + 1.0L) * (n + m - 1.0L) * (n - m - 1.0L)) /
                                                   ((long double)(n - m) * (n + m) * (2.0L * n - 3.0L)))
                                           : 0.0L;
        for (int k = 0; k < K + 2; ++k) p[k] = ((k > 0) ? a * p1[k - 1] : 0.0L) - b * p2[k];
      }
      std::vector<long double> e(p);
      for (int j = 0; j < m / 2; ++j) {  // times (1 - mu^2)
        std::vector<long double> t(e.size(), 0.0L);
        for (size_t k = 0; k < e.size(); ++k) {
          t[k] += e[k];
          if (k + 2 < e.size()) t[k + 2] -= e[k];
        }
        e = t;
      }
      const long double cs = (long double)sh_const::coef_scale(n, m);
      for (int k = 0; k <= L; ++k) {
        if (e[k] == 0.0L) continue;  // the other parity: never written
        for (int part = 0; part < (m ? 2 : 1); ++part) {
          const int o = (2 * m + part) * K + k;
          const int t = fill[o]++;
          if (t >= W) {  // cannot happen (n runs over one parity class above max(m, k)); never truncate silently
            val.clear();
            return;
          }
          val[(size_t)o * W + t] = (double)(e[k] / cs);
          col[(size_t)o * W + t] = n * n + n + (part ? -m : m);
        }
      }
      p2 = p1;
      p1 = p;
    }
  }
}

void to_m_major(int L, int width, const std::vector<double>& src, std::vector<double>& dst)
{
  dst.assign(src.size(), 0.0);
  for (int m = 0; m <= L; ++m)
    for (int n = m; n <= L; ++n) {
      const int kd = m * (L + 1) - m * (m - 1) / 2 + (n - m);
      for (int a = 0; a < width; ++a) dst[(size_t)width * kd + a] = src[(size_t)width * idx(n, m) + a];
    }
}

void real_sh_all(int L, const double u[3], double* out)
{
  const double x = u[0], y = u[1], z = u[2];
  const double s2 = std::sqrt(2.0);
  double Cm = 1.0, Sm = 0.0;
  for (int m = 0; m <= L; ++m) {
    const double pmm = sh_const::pmm(m);
    double p2 = 0.0, p1 = pmm;
    const double sgn = (m % 2) ? -1.0 : 1.0;
    for (int n = m; n <= L; ++n) {
      double p;
      if (n == m) p = pmm;
      else {
        const double b = (n - m >= 2) ? sh_const::beta(n, m) : 0.0;
        p = sh_const::alpha(n, m) * z * p1 - b * p2;
      }
      if (m == 0) out[n * n + n] = p;
      else {
        out[n * n + n + m] = s2 * sgn * p * Cm;
        out[n * n + n - m] = s2 * sgn * p * Sm;
      }
      if (n > m) { p2 = p1; p1 = p; }
    }
    const double c = Cm * x - Sm * y, s = Cm * y + Sm * x;
    Cm = c;
    Sm = s;
  }
}

void real_coefficients(int L, int lmax, const double* anm, std::vector<double>& c)
{
  c.assign((size_t)(L + 1) * (L + 1), 0.0);
  const double s2 = std::sqrt(2.0);
  for (int l = 0; l <= lmax && l <= L; ++l)
    for (int m = 0; m <= l; ++m) {
      const int k = idx(l, m);
      if (m == 0) c[l * l + l] = anm[2 * k];
      else {
        const double sg = s2 * ((m % 2) ? -1.0 : 1.0);
        c[l * l + l + m] = sg * anm[2 * k];
        c[l * l + l - m] = -sg * anm[2 * k + 1];
      }
    }
}

void build_xmats(int L, std::vector<double>& xp, std::vector<double>& xpt)
{
  const int nt = L + 2, np = 2 * nt;
  std::vector<double> t, w;
  gauss_legendre(nt, t, w);
  size_t tot = 0;
  for (int l = 0; l <= L; ++l) tot += (size_t)(2 * l + 1) * (2 * l + 1);
  xp.assign(tot, 0.0);
  xpt.assign(tot, 0.0);
  const int NS = (L + 1) * (L + 1);
  std::vector<double> su(NS), sv(NS);
  for (int a = 0; a < nt; ++a) {
    const double ct = t[a], st = std::sqrt(1.0 - ct * ct);
    for (int b = 0; b < np; ++b) {
      const double ph = 2.0 * (double)kPi * b / np;
      const double u[3] = {st * std::cos(ph), st * std::sin(ph), ct};
      const double v[3] = {u[0], -u[2], u[1]};  // Rx(+90) u
      const double wq = w[a] * 2.0 * (double)kPi / np;
      real_sh_all(L, u, su.data());
      real_sh_all(L, v, sv.data());
      size_t off = 0;
      for (int l = 0; l <= L; ++l) {
        const int n = 2 * l + 1;
        for (int r = 0; r < n; ++r)
          for (int c = 0; c < n; ++c) xp[off + (size_t)r * n + c] += wq * su[l * l + r] * sv[l * l + c];
        off += (size_t)n * n;
      }
    }
  }
  size_t off = 0;
  for (int l = 0; l <= L; ++l) {
    const int n = 2 * l + 1;
    for (int r = 0; r < n; ++r)
      for (int c = 0; c < n; ++c) {
        double& e = xp[off + (size_t)r * n + c];
        if (std::fabs(e) < 1e-14) e = 0.0;  // structural zeros of the parity pattern
        xpt[off + (size_t)c * n + r] = e;
      }
    off += (size_t)n * n;
  }
}

void build_xmats_ell(int L, std::vector<double>& val, std::vector<int>& col, std::vector<int>& info)
{
  // Row (l, m) of X (and of X^T) couples to the columns m' = first, first + 2, ... of ONE parity class
  // (sh_const::xpat_first / xpat_count): slot t of the row holds the entry of column first + 2 t whether or not it
  // happens to vanish, so that the rotation kernel of the compiled orders can address the columns at compile time.
  // Everything outside the pattern must be a structural zero: checked here, an empty `val` reports the failure.
  std::vector<double> xp, xpt;
  build_xmats(L, xp, xpt);
  const int ns = (L + 1) * (L + 1), W = L / 2 + 1;
  val.assign((size_t)2 * ns * W, 0.0);
  col.assign((size_t)2 * ns * W, 0);
  info.assign(ns, 0);
  for (int which = 0; which < 2; ++which) {
    const std::vector<double>& x = which ? xpt : xp;
    size_t off = 0;
    for (int l = 0; l <= L; ++l) {
      const int n = 2 * l + 1;
      for (int r = 0; r < n; ++r) {
        const int e = l * l + r;
        info[e] = l | (r << 8);
        const int first = sh_const::xpat_first(l, r - l), count = sh_const::xpat_count(l, r - l);
        for (int c = 0; c < n; ++c) {
          const double v = x[off + (size_t)r * n + c];
          const int d = (c - l) - first;
          const bool inpat = d >= 0 && (d % 2) == 0 && d / 2 < count;
          if (inpat) {
            val[((size_t)which * ns + e) * W + d / 2] = v;
          } else if (v != 0.0) {  // cannot happen for Rx(90): never truncate silently
            val.clear();
            return;
          }
        }
        for (int t = 0; t < W; ++t) col[((size_t)which * ns + e) * W + t] = (t < count) ? l * l + l + first + 2 * t : l * l;
      }
      off += (size_t)n * n;
    }
  }
}

void build_ring_scale(int L, std::vector<double>& g)
{
  g.assign((size_t)(L + 1) * (L + 1), 0.0);
  const double s2 = std::sqrt(2.0);
  for (int l = 0; l <= L; ++l)
    for (int m = 0; m <= l; ++m) {
      const double cs = sh_const::coef_scale(l, m);
      if (m == 0) g[l * l + l] = cs;
      else {
        const double v = s2 * ((m % 2) ? -1.0 : 1.0) * cs;
        g[l * l + l + m] = v;
        g[l * l + l - m] = v;
      }
    }
}

double host_radius(int lmax, const double* anm, const double u[3])
{
  // r = sum_m Re[W_m (x+iy)^m] with the plain normalised recurrence
  const double x = u[0], y = u[1], z = u[2];
  double Cm = 1.0, Sm = 0.0, r = 0.0;
  double pmm = std::sqrt(1.0 / (4.0 * (double)kPi));
  for (int m = 0; m <= lmax; ++m) {
    if (m > 0) pmm = -pmm * std::sqrt((2.0 * m + 1.0) / (2.0 * m));
    const double fac = (m == 0) ? 1.0 : 2.0;
    double p2 = 0.0, p1 = pmm;
    double Wr = anm[2 * idx(m, m)] * p1, Wi = anm[2 * idx(m, m) + 1] * p1;
    for (int n = m + 1; n <= lmax; ++n) {
      const double a = std::sqrt((4.0 * n * n - 1.0) / ((double)n * n - (double)m * m));
      const double b = (n - m < 2) ? 0.0
                                   : std::sqrt(((2.0 * n + 1.0) * (n + m - 1.0) * (n - m - 1.0)) /
                                               ((double)(n - m) * (n + m) * (2.0 * n - 3.0)));
      const double p = a * z * p1 - b * p2;
      Wr += anm[2 * idx(n, m)] * p;
      Wi += anm[2 * idx(n, m) + 1] * p;
      p2 = p1;
      p1 = p;
    }
    r += fac * (Wr * Cm - (m == 0 ? 0.0 : Wi * Sm));
    const double c = Cm * x - Sm * y, s = Cm * y + Sm * x;
    Cm = c;
    Sm = s;
  }
  return r;
}

double default_rmax(int lmax, const double* anm)
{
  const int nt = 6 * (lmax + 1) + 2, np = 2 * nt;
  std::vector<double> t, w;
  gauss_legendre(nt, t, w);
  double best = 0.0;
  for (int a = 0; a < nt; ++a) {
    const double ct = t[a], st = std::sqrt(1.0 - ct * ct);
    for (int b = 0; b < np; ++b) {
      const double ph = 2.0 * (double)kPi * b / np;
      const double u[3] = {st * std::cos(ph), st * std::sin(ph), ct};
      const double r = host_radius(lmax, anm, u);
      if (r > best) best = r;
    }
  }
  return 1.01 * best;
}

// The largest radius of the shape: the sample grid of default_rmax(), then a derivative-free local search on the
// sphere (shrinking tangent-plane pattern) from the best nodes.  Used to REFUSE a bounding radius that is too small;
// the default itself stays 1.01 x the sampled maximum (docs/SPEC.md §1).
double refined_max_radius(int lmax, const double* anm)
{
  const int nt = 6 * (lmax + 1) + 2, np = 2 * nt;
  std::vector<double> t, w;
  gauss_legendre(nt, t, w);
  struct Node {
    double r, u[3];
  };
  std::vector<Node> best;
  const size_t keep = 12;
  for (int a = 0; a < nt; ++a) {
    const double ct = t[a], st = std::sqrt(1.0 - ct * ct);
    for (int b = 0; b < np; ++b) {
      const double ph = 2.0 * (double)kPi * b / np;
      Node n;
      n.u[0] = st * std::cos(ph);
      n.u[1] = st * std::sin(ph);
      n.u[2] = ct;
      n.r = host_radius(lmax, anm, n.u);
      if (best.size() < keep) {
        best.push_back(n);
      } else {
        size_t lo = 0;
        for (size_t k = 1; k < keep; ++k)
          if (best[k].r < best[lo].r) lo = k;
        if (n.r > best[lo].r) best[lo] = n;
      }
    }
  }
  double rmax = 0.0;
  for (Node n : best) {
    double h = 2.0 * (double)kPi / np;  // one grid cell
    for (int it = 0; it < 60 && h > 1e-9; ++it) {
      // tangent basis at n.u
      const int k = std::fabs(n.u[0]) < 0.6 ? 0 : 1;
      double e[3] = {0.0, 0.0, 0.0};
      e[k] = 1.0;
      const double d = e[0] * n.u[0] + e[1] * n.u[1] + e[2] * n.u[2];
      double a1[3], a2[3];
      double na = 0.0;
      for (int q = 0; q < 3; ++q) {
        a1[q] = e[q] - d * n.u[q];
        na += a1[q] * a1[q];
      }
      na = std::sqrt(na);
      for (int q = 0; q < 3; ++q) a1[q] /= na;
      a2[0] = n.u[1] * a1[2] - n.u[2] * a1[1];
      a2[1] = n.u[2] * a1[0] - n.u[0] * a1[2];
      a2[2] = n.u[0] * a1[1] - n.u[1] * a1[0];
      bool moved = false;
      for (int dir = 0; dir < 8; ++dir) {
        const double c = std::cos(dir * (double)kPi / 4.0), sn = std::sin(dir * (double)kPi / 4.0);
        double v[3], nv = 0.0;
        for (int q = 0; q < 3; ++q) {
          v[q] = n.u[q] + h * (c * a1[q] + sn * a2[q]);
          nv += v[q] * v[q];
        }
        nv = std::sqrt(nv);
        for (int q = 0; q < 3; ++q) v[q] /= nv;
        const double r = host_radius(lmax, anm, v);
        if (r > n.r) {
          n.r = r;
          for (int q = 0; q < 3; ++q) n.u[q] = v[q];
          moved = true;
        }
      }
      if (!moved) h *= 0.5;
    }
    if (n.r > rmax) rmax = n.r;
  }
  return rmax;
}

// docs/SPEC.md §5. Azimuth nodes start at phi = 0 (any offset is exact for these integrands).
void mass_props(int lmax, const double* anm, double out[10])
{
  const int nt = (5 * lmax) / 2 + 3, np = 5 * lmax + 4;
  std::vector<double> t, w;
  gauss_legendre(nt, t, w);
  long double V = 0, c[3] = {0, 0, 0}, J[6] = {0, 0, 0, 0, 0, 0};
  for (int a = 0; a < nt; ++a) {
    const double ct = t[a], st = std::sqrt(1.0 - ct * ct);
    for (int b = 0; b < np; ++b) {
      const double ph = 2.0 * (double)kPi * b / np;
      const double u[3] = {st * std::cos(ph), st * std::sin(ph), ct};
      const long double r = host_radius(lmax, anm, u);
      const long double dw = (long double)w[a] * 2.0L * kPi / np;
      const long double r3 = r * r * r, r4 = r3 * r, r5 = r4 * r / 5.0L;
      V += dw * r3 / 3.0L;
      for (int k = 0; k < 3; ++k) c[k] += dw * r4 / 4.0L * u[k];
      J[0] += dw * r5 * (1.0L - u[0] * u[0]);
      J[1] += dw * r5 * (1.0L - u[1] * u[1]);
      J[2] += dw * r5 * (1.0L - u[2] * u[2]);
      J[3] -= dw * r5 * u[0] * u[1];
      J[4] -= dw * r5 * u[0] * u[2];
      J[5] -= dw * r5 * u[1] * u[2];
    }
  }
  for (int k = 0; k < 3; ++k) c[k] /= V;
  const long double c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
  out[0] = (double)V;
  for (int k = 0; k < 3; ++k) out[1 + k] = (double)c[k];
  for (int k = 0; k < 3; ++k) out[4 + k] = (double)(J[k] - V * (c2 - c[k] * c[k]));
  out[7] = (double)(J[3] + V * c[0] * c[1]);
  out[8] = (double)(J[4] + V * c[0] * c[2]);
  out[9] = (double)(J[5] + V * c[1] * c[2]);
}

// Inverse of rho * J (symmetric, xx,yy,zz,xy,xz,yz); false if not positive definite.
bool inertia_inverse(const double mp[10], double rho, double inv[6])
{
  const double a = rho * mp[4], b = rho * mp[5], c = rho * mp[6], d = rho * mp[7], e = rho * mp[8], f = rho * mp[9];
  const double det = a * (b * c - f * f) - d * (d * c - f * e) + e * (d * f - b * e);
  if (!(a > 0.0) || !(a * b - d * d > 0.0) || !(det > 0.0)) return false;
  inv[0] = (b * c - f * f) / det;
  inv[1] = (a * c - e * e) / det;
  inv[2] = (a * b - d * d) / det;
  inv[3] = (e * f - d * c) / det;
  inv[4] = (d * f - b * e) / det;
  inv[5] = (d * e - a * f) / det;
  return true;
}

}  // namespace shp
