// sh_const.hpp — recurrence constants of docs/SPEC.md §1 as constant expressions.
//
// Shared by the device templates (where they fold into the instruction stream)
// and by the host table builder (sh_tables.cpp), so both sides use bit-identical
// values: every function is plain IEEE double arithmetic, evaluated either by
// the compiler's constant evaluator or by the host CPU.
#pragma once

#if defined(__HIPCC__) || defined(__HIP__)
#define SHP_HD __host__ __device__
#else
#define SHP_HD
#endif

namespace shp {

// m-major table index: all n of one m are contiguous (wide scalar loads).
SHP_HD constexpr int sh_moff(int L, int m) { return m * (L + 1) - m * (m - 1) / 2; }
SHP_HD constexpr int sh_index(int L, int n, int m) { return sh_moff(L, m) + (n - m); }

// doubles per shape in the device coefficient table: 2T plus room for the last
// 4-term chunk to run past the end, rounded to a 64-byte multiple
SHP_HD constexpr int sh_chunk_stride(int L) { return ((L + 1) * (L + 2) + 6 + 7) / 8 * 8; }

namespace sh_const {

// Newton square root, monotone from above; exact to the last ulp or two.
SHP_HD constexpr double csqrt(double v)
{
  if (!(v > 0.0)) return 0.0;
  double x = (v > 1.0) ? v : 1.0;
  for (int i = 0; i < 200; ++i) {
    const double nx = 0.5 * (x + v / x);
    if (!(nx < x)) break;
    x = nx;
  }
  return x;
}

// alpha_nm = sqrt((4n^2-1)/(n^2-m^2)), n > m
SHP_HD constexpr double alpha(int n, int m)
{
  return csqrt((4.0 * n * n - 1.0) / ((double)n * n - (double)m * m));
}
// beta_nm = sqrt((2n+1)(n+m-1)(n-m-1)/((n-m)(n+m)(2n-3))), n >= m+2
SHP_HD constexpr double beta(int n, int m)
{
  return csqrt(((2.0 * n + 1.0) * (n + m - 1.0) * (n - m - 1.0)) / ((double)(n - m) * (n + m) * (2.0 * n - 3.0)));
}
// Pi_m^m = (-1)^m N_mm (2m-1)!!
SHP_HD constexpr double pmm(int m)
{
  double p = csqrt(1.0 / (4.0 * 3.14159265358979323846264338327950288));
  for (int k = 1; k <= m; ++k) p = -p * csqrt((2.0 * k + 1.0) / (2.0 * k));
  return p;
}
// s_nm with Pi_n^m = s_nm Q_n^m:  s_m = s_{m+1} = 1, s_n = beta_nm s_{n-2}
SHP_HD constexpr double scale(int n, int m)
{
  double s = 1.0;
  for (int k = n; k >= m + 2; k -= 2) s *= beta(k, m);
  return s;
}
// The kernels run the recurrence on Q_n = Pi_n^m / (s_nm Pi_m^m):
//   Q_m = 1,  Q_{m+1} = a'_{m+1} z,  Q_n = a'_n z Q_{n-1} - Q_{n-2}.
// Both subtrahends (Q_{n-2} in general, Q_m = 1.0 for n = m+2) then cost no
// register: 1.0 is an inline constant of v_fma_f64, a general constant is not
// (one SGPR operand per VALU instruction on gfx950) and had to be copied to a
// VGPR pair per m-block.  Pi_m^m and s_nm are folded into the coefficients.
SHP_HD constexpr double aprime(int n, int m)
{
  if (n == m + 1) return alpha(n, m);
  return alpha(n, m) * scale(n - 1, m) / scale(n, m);
}
// what the host multiplies (2 - delta_m0) a_nm with
SHP_HD constexpr double coef_scale(int n, int m) { return scale(n, m) * pmm(m); }


// Sparsity of X = the real-harmonic matrix of Rx(+90) (and of its transpose), degree l: row m (m < 0 the sin, m >= 0
// the cos functions) couples to the columns m' = xpat_first, xpat_first + 2, ... (xpat_count of them, <= l / 2 + 1) —
// one sign class and one parity, by the y- and z-parities of the two functions.  The host's ELL tables are laid out by
// this rule and check it against the computed matrices (sh_tables.cpp build_xmats_ell).
SHP_HD constexpr int xpat_first(int l, int m)
{
  if (m >= 0) return ((l - m) % 2 == 0) ? (l & 1) : -l;
  return ((l - m) % 2 == 0) ? 1 - (l & 1) : -(l - 1);
}
SHP_HD constexpr int xpat_count(int l, int m)
{
  if (m >= 0) return ((l - m) % 2 == 0) ? l / 2 + 1 : (l + 1) / 2;
  return ((l - m) % 2 == 0) ? (l + 1) / 2 : l / 2;
}

}  // namespace sh_const
}  // namespace shp
