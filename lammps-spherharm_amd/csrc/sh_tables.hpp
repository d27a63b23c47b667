// sh_tables.hpp — host-side construction of the tables the kernels read.
#pragma once
#include <vector>

namespace shp {

// Gauss-Legendre nodes (ascending) and weights on [-1,1].
void gauss_legendre(int n, std::vector<double>& t, std::vector<double>& w);

// Recurrence constants rc[(L+1)(L+2)/2] in the layout of sh_device.hpp, and
// the scale s_nm such that Pi_n^m = s_nm Q_n^m.
void build_recurrence(int L, std::vector<double>& rc, std::vector<double>& scale);

// Kernel coefficients cw[(L+1)(L+2)] of a shape of order lmax <= L
// (zero padded), from the user's a_nm (docs/SPEC.md §1 storage).
void build_coefficients(int L, int lmax, const double* anm, const std::vector<double>& rc,
                        const std::vector<double>& scale, std::vector<double>& cw);

// Monomial form for the compiled-order kernels (L <= 12): W_m(z) = sum_n (2-delta_m0) a_nm Pi_n^m(z)
// as a polynomial of degree L-m in z, coefficients in DESCENDING powers (Horner order), complex,
// m-major in the same layout and stride as the recurrence table: wm[2 sh_index(L, m+k, m)] is the
// real part of the coefficient of z^(L-m-k).  Polynomial arithmetic in long double.
void build_monomial(int L, int lmax, const double* anm, std::vector<double>& wm);
// First stage of particle j's per-azimuth polynomials in the pair's common frame (sh_tables.cpp), ELL rows of width L/2+1.
void build_jpoly_ell(int L, std::vector<double>& val, std::vector<int>& col);

// Permutes an n-major table (k = n(n+1)/2+m, `width` doubles per term) into
// the m-major device layout of sh_device.hpp.
void to_m_major(int L, int width, const std::vector<double>& src, std::vector<double>& dst);

// ---- cap-frame evaluation of particle i (pair_kernel.hpp, rotation step) ----
// Real spherical harmonics S_lm, m = -l..l, index l*l + (m + l):
//   S_l0 = Y_l0,  S_lm = sqrt2 (-1)^m Re Y_lm,  S_l,-m = sqrt2 (-1)^m Im Y_lm  (m > 0).
// All (L+1)^2 values at unit vector u.
void real_sh_all(int L, const double u[3], double* out);
// Real-basis coefficients c_lm of a shape from its a_nm: r = sum c_lm S_lm.
void real_coefficients(int L, int lmax, const double* anm, std::vector<double>& c);
// X^l = representation of the fixed rotation Rx(+90 deg) on the real SH of order l:
// X^l[m'][m] = integral S_lm'(u) S_lm(Rx(90) u) dOmega, by exact Gauss x trapezoid
// quadrature.  xp: all l packed row-major, block l at offset l(4l^2-1)/3; xpt: the transposes.
void build_xmats(int L, std::vector<double>& xp, std::vector<double>& xpt);
// The same two matrices in ELL form for the kernel: a row of X^l has at most l/2+1
// non-zeros (parity structure of Rx(90)), so each of the (L+1)^2 rows gets W = L/2+1
// (value, absolute column index l^2 + c) slots, zero padded.  First X (rows 0..ns-1), then X^T.
// info[e] = l | (m + l) << 8 for row e.
void build_xmats_ell(int L, std::vector<double>& val, std::vector<int>& col, std::vector<int>& info);
// g_lm: what a rotated real coefficient is multiplied with to enter the ring recurrence
// (Q basis of sh_device.hpp): g_l0 = s_l0 Pi_0^0, g_l,+-m = sqrt2 (-1)^m s_lm Pi_m^m.
void build_ring_scale(int L, std::vector<double>& g);

// Host evaluation of r(u) straight from a_nm (setup only: bounding radii).
double host_radius(int lmax, const double* anm, const double u[3]);

// Default bounding radius: 1.01 x max over the (6(L+1)+2) x 2(6(L+1)+2) grid.
double default_rmax(int lmax, const double* anm);
double refined_max_radius(int lmax, const double* anm);  // largest radius incl. between the samples (local search)

// Rigid-body properties at unit density (docs/SPEC.md §5): V, c[3], J_c (xx,yy,zz,xy,xz,yz).
void mass_props(int lmax, const double* anm, double out[10]);
// inv[6] (xx,yy,zz,xy,xz,yz) = (rho J_c)^-1; false if the tensor is not positive definite.
bool inertia_inverse(const double mp[10], double rho, double inv[6]);

}  // namespace shp
