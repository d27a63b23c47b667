// CPU check of the host-built tables of libshpair (lammps-spherharm_amd/csrc/sh_tables.cpp):
// compiled and run by tests/test_host_tables.py, prints "name value" lines.
//  1. X^l = T(Rx(90)) is orthogonal, its ELL form reproduces it, rows have <= l/2+1 non-zeros.
//  2. The cap-frame pipeline on the host — c' = Z(g) X Z(b) X^T Z(a) c, ring scale, ring recurrence,
//     r = sum_m (A_m cos m psi + B_m sin m psi) — reproduces r_body(M u') for random rotations M,
//     including the pole-degenerate ones, using exactly the tables the kernel reads.
//  3. The monomial (Horner) table reproduces the recurrence evaluation.
//  4. Particle j's per-azimuth polynomials (build_jpoly_ell + the azimuth stage of pair_kernel.hpp jpoly_build):
//     r = G_l(mu) + sigma H_l(mu) on the quadrature's azimuths reproduces r_body(M u'), and the second half of the
//     azimuths is the first with the sign of H flipped.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "sh_const.hpp"
#include "sh_tables.hpp"

using namespace shp;

static double urand() { return rand() / (double)RAND_MAX; }

static void quat_to_mat(const double q[4], double R[9])
{
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
  R[3] = 2 * (x * y + w * z); R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
  R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = w * w - x * x - y * y + z * z;
}

int main(int argc, char** argv)
{
  const int L = argc > 1 ? atoi(argv[1]) : 6;
  srand(1234 + L);
  const int ns = (L + 1) * (L + 1), T = (L + 1) * (L + 2) / 2, XW = L / 2 + 1;
  // a random shape
  std::vector<double> anm(2 * T, 0.0);
  anm[0] = std::sqrt(4 * M_PI);
  for (int n = 1; n <= L; ++n)
    for (int m = 0; m <= n; ++m) {
      anm[2 * (n * (n + 1) / 2 + m)] = 0.3 * (urand() - 0.5) / (n + 1);
      if (m > 0) anm[2 * (n * (n + 1) / 2 + m) + 1] = 0.3 * (urand() - 0.5) / (n + 1);
    }

  // ---- 1. X matrices
  std::vector<double> xp, xpt, xval;
  std::vector<int> xcol, xinfo;
  build_xmats(L, xp, xpt);
  build_xmats_ell(L, xval, xcol, xinfo);
  double orth = 0.0, ell = 0.0;
  int maxnz_excess = 0;
  size_t off = 0;
  for (int l = 0; l <= L; ++l) {
    const int n = 2 * l + 1;
    for (int r = 0; r < n; ++r) {
      int nz = 0;
      for (int c = 0; c < n; ++c) {
        double s = 0.0;
        for (int k = 0; k < n; ++k) s += xp[off + r * n + k] * xp[off + c * n + k];
        orth = fmax(orth, fabs(s - (r == c)));
        if (xp[off + r * n + c] != 0.0) nz++;
        double e = 0.0;  // ELL row r of X applied to unit vector c
        for (int t = 0; t < XW; ++t)
          if (xcol[(size_t)(l * l + r) * XW + t] == l * l + c) e += xval[(size_t)(l * l + r) * XW + t];
        ell = fmax(ell, fabs(e - xp[off + r * n + c]));
        double et = 0.0;
        for (int t = 0; t < XW; ++t)
          if (xcol[((size_t)ns + l * l + r) * XW + t] == l * l + c) et += xval[((size_t)ns + l * l + r) * XW + t];
        ell = fmax(ell, fabs(et - xpt[off + r * n + c]));
      }
      // the layout the rotation kernel of the compiled orders addresses at compile time (sh_const::xpat_*): slot t of
      // the row is column first + 2 t for t < count, empty beyond — in both tables
      const int first = shp::sh_const::xpat_first(l, r - l), count = shp::sh_const::xpat_count(l, r - l);
      for (int which = 0; which < 2; ++which)
        for (int t = 0; t < XW; ++t) {
          const size_t k = ((size_t)which * ns + l * l + r) * XW + t;
          if (t < count ? (xcol[k] != l * l + l + first + 2 * t) : (xval[k] != 0.0)) ell = 2.0;
        }
      if (count > l / 2 + 1 || count < 1) ell = 3.0;
      if (nz - (l / 2 + 1) > maxnz_excess) maxnz_excess = nz - (l / 2 + 1);
      if (xinfo[l * l + r] != (l | (r << 8))) ell = 1.0;
    }
    off += (size_t)n * n;
  }
  printf("x_orthogonality %.3e\nx_ell_mismatch %.3e\nx_row_excess %d\n", orth, ell, maxnz_excess);

  // ---- 2. the cap-frame pipeline
  std::vector<double> creal, g, rc_n, scale, rc;
  real_coefficients(L, L, anm.data(), creal);
  build_ring_scale(L, g);
  build_recurrence(L, rc_n, scale);
  to_m_major(L, 1, rc_n, rc);
  double worst = 0.0, jworst = 0.0;
  std::vector<double> jval;
  std::vector<int> jcol;
  build_jpoly_ell(L, jval, jcol);
  if (jval.empty()) jworst = 1.0;
  for (int trial = 0; trial < 40; ++trial) {
    double q[4] = {urand() - 0.5, urand() - 0.5, urand() - 0.5, urand() - 0.5};
    if (trial == 0) { q[0] = 1; q[1] = q[2] = q[3] = 0; }                 // identity: sin(beta) = 0
    if (trial == 1) { q[0] = 0; q[1] = 1; q[2] = q[3] = 0; }                 // flip: cos(beta) = -1
    if (trial == 2) { q[0] = 1; q[1] = 1e-9; q[2] = -2e-9; q[3] = 0.3; }     // nearly polar
    if (trial == 3) { q[0] = 1e-7; q[1] = 1; q[2] = 0.2; q[3] = 1e-8; }
    const double nq = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (double& v : q) v /= nq;
    double M[9];
    quat_to_mat(q, M);  // columns = cap axes in the body frame: b1 = M[:,0], b2 = M[:,1], bc = M[:,2]
    const double b1[3] = {M[0], M[3], M[6]}, b2[3] = {M[1], M[4], M[7]}, bc[3] = {M[2], M[5], M[8]};
    // Euler angles exactly as pair_kernel.hpp: cap_frame_rotate
    const double cb = bc[2], sb2 = bc[0] * bc[0] + bc[1] * bc[1];
    double sb = 0.0, ca = 1.0, sa = 0.0;
    if (sb2 > 0.0) {
      const double n2 = 1.0 / std::sqrt(sb2);
      sb = sb2 * n2;
      ca = bc[0] * n2;
      sa = bc[1] * n2;
    }
    double cg, sg;
    if (cb >= 0.0) {
      const double iv = 1.0 / (1.0 + cb), cs = (b1[0] + b2[1]) * iv, ss = (b1[1] - b2[0]) * iv;
      cg = cs * ca + ss * sa;
      sg = ss * ca - cs * sa;
    } else {
      const double iv = 1.0 / (1.0 - cb), cd = -(b1[0] - b2[1]) * iv, sd = -(b1[1] + b2[0]) * iv;
      cg = ca * cd + sa * sd;
      sg = sa * cd - ca * sd;
    }
    const double c1[3] = {ca, cb, cg}, s1[3] = {sa, sb, sg};
    std::vector<double> v0(creal), v1(ns);
    for (int step = 0; step < 5; ++step) {
      const std::vector<double>& src = (step & 1) ? v0 : ((step == 0) ? creal : v1);
      std::vector<double>& dst = (step & 1) ? v1 : v0;
      std::vector<double> tmp(ns);
      for (int e = 0; e < ns; ++e) {
        const int l = xinfo[e] & 255, r = xinfo[e] >> 8, mm = r - l;
        if ((step & 1) == 0) {
          const int m = abs(mm);
          double cm = 1.0, sm = 0.0;
          for (int t = 0; t < m; ++t) {
            const double c = cm * c1[step >> 1] - sm * s1[step >> 1], s = cm * s1[step >> 1] + sm * c1[step >> 1];
            cm = c;
            sm = s;
          }
          const double self = src[e], other = src[l * l + l - mm];
          tmp[e] = (mm == 0) ? self : cm * self + (mm > 0 ? sm : -sm) * other;
          if (step == 4) tmp[e] *= g[e];
        } else {
          const size_t ro = ((size_t)(step == 1 ? ns : 0) + e) * XW;
          double o = 0.0;
          for (int t = 0; t < XW; ++t) o += xval[ro + t] * src[xcol[ro + t]];
          tmp[e] = o;
        }
      }
      dst = tmp;
    }
    const std::vector<double>& ch = v0;
    // compare on random cap-frame directions (mu, psi)
    for (int s = 0; s < 50; ++s) {
      const double mu = 2 * urand() - 1, psi = 2 * M_PI * urand(), sig = std::sqrt(1 - mu * mu);
      double r = 0.0;
      for (int m = 0; m <= L; ++m) {
        const double* rcm = rc.data() + sh_moff(L, m);
        double q2 = 0.0, q1 = 1.0, wa = ch[m * m + 2 * m], wb = (m > 0) ? ch[m * m] : 0.0;
        for (int n = m + 1; n <= L; ++n) {
          const double qq = rcm[n - m] * mu * q1 - q2;
          wa += ch[n * n + n + m] * qq;
          if (m > 0) wb += ch[n * n + n - m] * qq;
          q2 = q1;
          q1 = qq;
        }
        const double sp = std::pow(sig, m);
        r += sp * (wa * std::cos(m * psi) + wb * std::sin(m * psi));
      }
      const double up[3] = {sig * std::cos(psi), sig * std::sin(psi), mu};
      const double ub[3] = {M[0] * up[0] + M[1] * up[1] + M[2] * up[2], M[3] * up[0] + M[4] * up[1] + M[5] * up[2],
                            M[6] * up[0] + M[7] * up[1] + M[8] * up[2]};
      worst = fmax(worst, fabs(r - host_radius(L, anm.data(), ub)));
    }
    // ---- 4. per-azimuth polynomials of the same rotated vector, n_q azimuth pairs
    if (!jval.empty()) {
      const int K = L + 1, NR = (2 * L + 4) * K, nqa = 3 + trial % 6, npsi = 2 * nqa;
      std::vector<double> pj(NR, 0.0);
      for (int o = 0; o < NR; ++o)
        for (int t = 0; t < XW; ++t) pj[o] += jval[(size_t)o * XW + t] * ch[jcol[(size_t)o * XW + t]];
      for (int l = 0; l < npsi; ++l) {
        const int lrow = l >= nqa ? l - nqa : l;
        const double psi_row = 2.0 * M_PI * (lrow + 0.5) / npsi, psi = 2.0 * M_PI * (l + 0.5) / npsi;
        std::vector<double> G(K, 0.0), H(K, 0.0);
        for (int k = 0; k <= L; ++k)
          for (int m = 0; m <= L; ++m) {
            const double v = std::cos(m * psi_row) * pj[(2 * m) * K + k] + std::sin(m * psi_row) * pj[(2 * m + 1) * K + k];
            ((m & 1) ? H : G)[k] += v;
          }
        if (L >= 1 && H[L] != 0.0) jworst = 1.0;  // H has degree L - 1
        for (int k = 0; k <= L; ++k)               // the rows of the order L + 1 are empty
          if (pj[(2 * L + 2) * K + k] != 0.0 || pj[(2 * L + 3) * K + k] != 0.0) jworst = 1.0;
        for (int s = 0; s < 8; ++s) {
          const double mu = 2 * urand() - 1, sig = std::sqrt(1 - mu * mu);
          double g = G[L], h = (L >= 1) ? H[L - 1] : 0.0;
          for (int k = L - 1; k >= 0; --k) g = g * mu + G[k];
          for (int k = L - 2; k >= 0; --k) h = h * mu + H[k];
          const double r = g + (l >= nqa ? -sig : sig) * h;
          const double up[3] = {sig * std::cos(psi), sig * std::sin(psi), mu};
          const double ub[3] = {M[0] * up[0] + M[1] * up[1] + M[2] * up[2], M[3] * up[0] + M[4] * up[1] + M[5] * up[2],
                                M[6] * up[0] + M[7] * up[1] + M[8] * up[2]};
          jworst = fmax(jworst, fabs(r - host_radius(L, anm.data(), ub)));
        }
      }
    }
  }
  printf("cap_frame_error %.3e\n", worst);
  printf("jpoly_error %.3e\n", jworst);

  // ---- 3. monomial table vs recurrence
  std::vector<double> wm;
  build_monomial(L, L, anm.data(), wm);
  double herr = 0.0;
  for (int s = 0; s < 2000; ++s) {
    double u[3] = {urand() - 0.5, urand() - 0.5, urand() - 0.5};
    const double nn = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    for (double& v : u) v /= nn;
    double Cm = 1.0, Sm = 0.0, r = 0.0;
    for (int m = 0; m <= L; ++m) {
      const int base = sh_moff(L, m), d = L - m;
      double wr = wm[2 * base], wi = wm[2 * base + 1];
      for (int k = 1; k <= d; ++k) {
        wr = wr * u[2] + wm[2 * (base + k)];
        wi = wi * u[2] + wm[2 * (base + k) + 1];
      }
      r += wr * Cm - wi * Sm;
      const double c = Cm * u[0] - Sm * u[1], sn = Cm * u[1] + Sm * u[0];
      Cm = c;
      Sm = sn;
    }
    herr = fmax(herr, fabs(r - host_radius(L, anm.data(), u)));
  }
  printf("horner_error %.3e\n", herr);
  return 0;
}
