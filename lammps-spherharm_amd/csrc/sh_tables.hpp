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

// Permutes an n-major table (k = n(n+1)/2+m, `width` doubles per term) into
// the m-major device layout of sh_device.hpp.
void to_m_major(int L, int width, const std::vector<double>& src, std::vector<double>& dst);

// Host evaluation of r(u) straight from a_nm (setup only: bounding radii).
double host_radius(int lmax, const double* anm, const double u[3]);

// Default bounding radius: 1.01 x max over the (6(L+1)+2) x 2(6(L+1)+2) grid.
double default_rmax(int lmax, const double* anm);

}  // namespace shp
