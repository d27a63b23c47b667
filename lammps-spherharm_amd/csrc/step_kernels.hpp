// step_kernels.hpp — gfx950 kernels of docs/SPEC.md Part II: the nve integrator and body forces (§6),
// periodic ghosts and the binned half neighbour list (§7).  All of them are streaming FP64 / integer
// passes bound by HBM: one lane per particle, 256-lane workgroups, no LDS except in the scans.
#pragma once
#include <hip/hip_runtime.h>

namespace shp {

constexpr int kStepBlock = 256;
constexpr int kMassStride = 16;  // doubles per shape row: m, 1/m, c[3], Iinv (xx,yy,zz,xy,xz,yz), rmax, pad

// device error flags (shstep_state::d_flags[0]), read back at the blocking calls
constexpr int kErrShape = 1;  // shape index outside the table

struct Mat3 {
  double m[3][3];
};

__device__ inline Mat3 rot_of(const double w, const double x, const double y, const double z)
{
  Mat3 R;
  R.m[0][0] = w * w + x * x - y * y - z * z; R.m[0][1] = 2 * (x * y - w * z); R.m[0][2] = 2 * (x * z + w * y);
  R.m[1][0] = 2 * (x * y + w * z); R.m[1][1] = w * w - x * x + y * y - z * z; R.m[1][2] = 2 * (y * z - w * x);
  R.m[2][0] = 2 * (x * z - w * y); R.m[2][1] = 2 * (y * z + w * x); R.m[2][2] = w * w - x * x - y * y + z * z;
  return R;
}

// omega = R Iinv R^T L
__device__ inline void omega_of(const Mat3& R, const double* __restrict__ mr, const double L[3], double w[3])
{
  double lb[3], wb[3];
  for (int k = 0; k < 3; ++k) lb[k] = R.m[0][k] * L[0] + R.m[1][k] * L[1] + R.m[2][k] * L[2];
  wb[0] = mr[5] * lb[0] + mr[8] * lb[1] + mr[9] * lb[2];
  wb[1] = mr[8] * lb[0] + mr[6] * lb[1] + mr[10] * lb[2];
  wb[2] = mr[9] * lb[0] + mr[10] * lb[1] + mr[7] * lb[2];
  for (int k = 0; k < 3; ++k) w[k] = R.m[k][0] * wb[0] + R.m[k][1] * wb[1] + R.m[k][2] * wb[2];
}

__device__ inline void qdot_of(const double q[4], const double* __restrict__ mr, const double L[3], double qd[4])
{
  const Mat3 R = rot_of(q[0], q[1], q[2], q[3]);
  double w[3];
  omega_of(R, mr, L, w);
  qd[0] = 0.5 * (-w[0] * q[1] - w[1] * q[2] - w[2] * q[3]);
  qd[1] = 0.5 * (w[0] * q[0] + w[1] * q[3] - w[2] * q[2]);
  qd[2] = 0.5 * (w[1] * q[0] + w[2] * q[1] - w[0] * q[3]);
  qd[3] = 0.5 * (w[2] * q[0] + w[0] * q[2] - w[1] * q[1]);
}

__device__ inline void qnormalize(double q[4])
{
  const double n = 1.0 / __builtin_sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  for (int k = 0; k < 4; ++k) q[k] *= n;
}

// SPEC §6. PHASE 0 = initial_integrate, 1 = final_integrate.
template <int PHASE>
__global__ __launch_bounds__(kStepBlock) void nve_kernel(int n, double dt, const double* __restrict__ mass, int nshapes,
                                                         double* __restrict__ x, double* __restrict__ v,
                                                         double* __restrict__ quat, double* __restrict__ angmom,
                                                         const double* __restrict__ f, const double* __restrict__ tq,
                                                         const int* __restrict__ shtype, const int* __restrict__ mask,
                                                         int groupbit, int* __restrict__ flags)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  if (i >= n) return;
  if (!(mask[i] & groupbit)) return;
  const int st = shtype[i];
  if ((unsigned)st >= (unsigned)nshapes) {
    atomicOr(flags, kErrShape);
    return;
  }
  const double* __restrict__ mr = mass + (size_t)kMassStride * st;
  double q[4] = {quat[4 * i], quat[4 * i + 1], quat[4 * i + 2], quat[4 * i + 3]};
  const double fi[3] = {f[3 * i], f[3 * i + 1], f[3 * i + 2]};
  Mat3 R = rot_of(q[0], q[1], q[2], q[3]);
  double s[3], L[3], vv[3];
  for (int k = 0; k < 3; ++k) s[k] = R.m[k][0] * mr[2] + R.m[k][1] * mr[3] + R.m[k][2] * mr[4];
  const double sf[3] = {s[1] * fi[2] - s[2] * fi[1], s[2] * fi[0] - s[0] * fi[2], s[0] * fi[1] - s[1] * fi[0]};
  const double hm = 0.5 * dt * mr[1], h = 0.5 * dt;
  for (int k = 0; k < 3; ++k) {
    vv[k] = v[3 * i + k] + hm * fi[k];
    L[k] = angmom[3 * i + k] + h * (tq[3 * i + k] - sf[k]);
    v[3 * i + k] = vv[k];
    angmom[3 * i + k] = L[k];
  }
  if (PHASE == 1) return;
  double X[3];
  for (int k = 0; k < 3; ++k) X[k] = x[3 * i + k] + s[k] + dt * vv[k];
  // richardson
  double qd[4], qf[4], qh[4];
  qdot_of(q, mr, L, qd);
  for (int k = 0; k < 4; ++k) {
    qf[k] = q[k] + dt * qd[k];
    qh[k] = q[k] + h * qd[k];
  }
  qnormalize(qf);
  qnormalize(qh);
  qdot_of(qh, mr, L, qd);
  for (int k = 0; k < 4; ++k) qh[k] += h * qd[k];
  qnormalize(qh);
  for (int k = 0; k < 4; ++k) q[k] = 2.0 * qh[k] - qf[k];
  qnormalize(q);
  R = rot_of(q[0], q[1], q[2], q[3]);
  for (int k = 0; k < 3; ++k) {
    s[k] = R.m[k][0] * mr[2] + R.m[k][1] * mr[3] + R.m[k][2] * mr[4];
    x[3 * i + k] = X[k] - s[k];
  }
  for (int k = 0; k < 4; ++k) quat[4 * i + k] = q[k];
}

__global__ __launch_bounds__(kStepBlock) void post_force_kernel(int n, const double* __restrict__ mass, int nshapes,
                                                                double gx, double gy, double gz, double gamma_t,
                                                                double gamma_r, const double* __restrict__ v,
                                                                const double* __restrict__ quat,
                                                                const double* __restrict__ angmom,
                                                                const int* __restrict__ shtype,
                                                                const int* __restrict__ mask, int groupbit,
                                                                double* __restrict__ f, double* __restrict__ tq,
                                                                int* __restrict__ flags)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  if (i >= n) return;
  if (!(mask[i] & groupbit)) return;
  const int st = shtype[i];
  if ((unsigned)st >= (unsigned)nshapes) {
    atomicOr(flags, kErrShape);
    return;
  }
  const double* __restrict__ mr = mass + (size_t)kMassStride * st;
  const Mat3 R = rot_of(quat[4 * i], quat[4 * i + 1], quat[4 * i + 2], quat[4 * i + 3]);
  double s[3], w[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k) s[k] = R.m[k][0] * mr[2] + R.m[k][1] * mr[3] + R.m[k][2] * mr[4];
  const double g[3] = {gx, gy, gz};
  double Fb[3];
  for (int k = 0; k < 3; ++k) Fb[k] = mr[0] * g[k] - gamma_t * v[3 * i + k];
  if (gamma_r != 0.0) {
    const double L[3] = {angmom[3 * i], angmom[3 * i + 1], angmom[3 * i + 2]};
    omega_of(R, mr, L, w);
  }
  const double sF[3] = {s[1] * Fb[2] - s[2] * Fb[1], s[2] * Fb[0] - s[0] * Fb[2], s[0] * Fb[1] - s[1] * Fb[0]};
  for (int k = 0; k < 3; ++k) {
    f[3 * i + k] += Fb[k];
    tq[3 * i + k] += sF[k] - gamma_r * w[k];
  }
}

__device__ inline double wave_sum_f64(double v)
{
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__global__ __launch_bounds__(kStepBlock) void energies_kernel(int n, const double* __restrict__ mass, int nshapes,
                                                              double gx, double gy, double gz,
                                                              const double* __restrict__ x, const double* __restrict__ v,
                                                              const double* __restrict__ quat,
                                                              const double* __restrict__ angmom,
                                                              const int* __restrict__ shtype,
                                                              const int* __restrict__ mask, int groupbit,
                                                              double* __restrict__ out, int* __restrict__ flags)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  double e[3] = {0, 0, 0};
  if (i < n && (mask[i] & groupbit)) {
    const int st = shtype[i];
    if ((unsigned)st >= (unsigned)nshapes) {
      atomicOr(flags, kErrShape);
    } else {
      const double* __restrict__ mr = mass + (size_t)kMassStride * st;
      const Mat3 R = rot_of(quat[4 * i], quat[4 * i + 1], quat[4 * i + 2], quat[4 * i + 3]);
      const double L[3] = {angmom[3 * i], angmom[3 * i + 1], angmom[3 * i + 2]};
      double w[3];
      omega_of(R, mr, L, w);
      const double g[3] = {gx, gy, gz};
      for (int k = 0; k < 3; ++k) {
        const double s = R.m[k][0] * mr[2] + R.m[k][1] * mr[3] + R.m[k][2] * mr[4];
        e[0] += 0.5 * mr[0] * v[3 * i + k] * v[3 * i + k];
        e[1] += 0.5 * w[k] * L[k];
        e[2] -= mr[0] * g[k] * (x[3 * i + k] + s);
      }
    }
  }
  for (int k = 0; k < 3; ++k) {
    const double t = wave_sum_f64(e[k]);
    if ((threadIdx.x & 63) == 0 && t != 0.0) atomicAdd(&out[k], t);
  }
}

// ---------------------------------------------------------------- scans
// Exclusive scan of n ints in three passes; the middle pass walks the block sums in one workgroup.
constexpr int kScanBlock = 1024;

__device__ inline int block_exclusive_scan(int val, int* lds, int* total)
{
  // Hillis-Steele over 1024 lanes via wave scans + one scan of the 16 wave sums
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int inc = val;
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(inc, off, 64);
    if (lane >= off) inc += t;
  }
  if (lane == 63) lds[wv] = inc;
  __syncthreads();
  if (wv == 0) {
    int s = (lane < kScanBlock / 64) ? lds[lane] : 0;
    int sinc = s;
    for (int off = 1; off < 16; off <<= 1) {
      const int t = __shfl_up(sinc, off, 64);
      if (lane >= off) sinc += t;
    }
    if (lane < kScanBlock / 64) lds[lane] = sinc - s;
    if (lane == kScanBlock / 64 - 1) lds[16] = sinc;
  }
  __syncthreads();
  const int res = lds[wv] + inc - val;
  *total = lds[16];
  __syncthreads();
  return res;
}

__global__ __launch_bounds__(kScanBlock) void scan_local_kernel(const int* __restrict__ in, int n, int* __restrict__ out,
                                                                int* __restrict__ block_sums)
{
  __shared__ int lds[17];
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  const int v = (i < n) ? in[i] : 0;
  int total;
  const int ex = block_exclusive_scan(v, lds, &total);
  if (i < n) out[i] = ex;
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// One workgroup: exclusive scan of the block sums in place, chunk by chunk; sums[nblocks] = grand total.
__global__ __launch_bounds__(kScanBlock) void scan_sums_kernel(int* __restrict__ sums, int nblocks)
{
  __shared__ int lds[17];
  int carry = 0;
  for (int base = 0; base < nblocks; base += kScanBlock) {
    const int i = base + threadIdx.x;
    const int v = (i < nblocks) ? sums[i] : 0;
    int total;
    const int ex = block_exclusive_scan(v, lds, &total);
    if (i < nblocks) sums[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) sums[nblocks] = carry;
}

// out[i] += sums[block]; out[n] = grand total (so that out has n + 1 entries: CSR offsets).
__global__ __launch_bounds__(kScanBlock) void scan_apply_kernel(int* __restrict__ out, int n, const int* __restrict__ sums,
                                                                int nblocks)
{
  const int i = blockIdx.x * kScanBlock + threadIdx.x;
  if (i < n) out[i] += sums[blockIdx.x];
  if (i == 0) out[n] = sums[nblocks];
}

// ---------------------------------------------------------------- SPEC §7: borders
struct BoxParams {
  double lo[3], hi[3], len[3];
  int periodic[3];
  double cmax;   // ghost cutoff = bin size
  double glo[3];  // origin of the bin grid
  double binv[3];
  int nc[3];
};

__device__ inline bool image_wanted(const BoxParams& b, const double xi[3], int code)
{
  const int s[3] = {code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1};
  bool ok = (code != 13);
  for (int d = 0; d < 3; ++d) {
    if (s[d] && !b.periodic[d]) ok = false;
    if (s[d] == 1 && !(xi[d] < b.lo[d] + b.cmax)) ok = false;
    if (s[d] == -1 && !(xi[d] >= b.hi[d] - b.cmax)) ok = false;
  }
  return ok;
}

// Domain::pbc + image count. Bounded wrap: a particle further than 1e6 box lengths away (or NaN) is left alone.
__global__ __launch_bounds__(kStepBlock) void wrap_count_kernel(int n, BoxParams b, double* __restrict__ x,
                                                                int* __restrict__ cnt)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  if (i >= n) return;
  double xi[3];
  for (int d = 0; d < 3; ++d) {
    double p = x[3 * i + d];
    if (b.periodic[d] && (p < b.lo[d] || p >= b.hi[d])) {
      const double k = __builtin_floor((p - b.lo[d]) / b.len[d]);
      if (__builtin_fabs(k) < 1e6) {
        p -= k * b.len[d];
        // rounding at the faces
        if (p < b.lo[d]) p += b.len[d];
        if (p >= b.hi[d]) p -= b.len[d];
        if (p < b.lo[d]) p = b.lo[d];
        x[3 * i + d] = p;
      }
    }
    xi[d] = p;
  }
  int c = 0;
  for (int code = 0; code < 27; ++code) c += image_wanted(b, xi, code) ? 1 : 0;
  cnt[i] = c;
}

// ghost g of owner i: rows nlocal + g of the caller's arrays; ctx-owned owner / shift-code tables.
__global__ __launch_bounds__(kStepBlock) void fill_ghosts_kernel(int n, int nmax, BoxParams b, const int* __restrict__ goff,
                                                                 double* __restrict__ x, double* __restrict__ quat,
                                                                 int* __restrict__ type, int* __restrict__ shtype,
                                                                 int* __restrict__ tag, int* __restrict__ gowner,
                                                                 int* __restrict__ gcode)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  if (i >= n) return;
  int g = goff[i];
  if (g == goff[i + 1]) return;
  const double xi[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
  for (int code = 0; code < 27; ++code) {
    if (!image_wanted(b, xi, code)) continue;
    const int row = n + g;
    if (row < nmax) {
      const int s[3] = {code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1};
      for (int d = 0; d < 3; ++d) x[3 * row + d] = xi[d] + s[d] * b.len[d];
      for (int k = 0; k < 4; ++k) quat[4 * row + k] = quat[4 * i + k];
      type[row] = type[i];
      shtype[row] = shtype[i];
      if (tag) tag[row] = tag[i];
      gowner[g] = i;
      gcode[g] = code;
    }
    ++g;
  }
}

__global__ __launch_bounds__(kStepBlock) void forward_kernel(int nlocal, int nghost, BoxParams b,
                                                             const int* __restrict__ gowner, const int* __restrict__ gcode,
                                                             double* __restrict__ x, double* __restrict__ quat)
{
  const int g = blockIdx.x * kStepBlock + threadIdx.x;
  if (g >= nghost) return;
  const int i = gowner[g], code = gcode[g], row = nlocal + g;
  const int s[3] = {code % 3 - 1, (code / 3) % 3 - 1, code / 9 - 1};
  for (int d = 0; d < 3; ++d) x[3 * row + d] = x[3 * i + d] + s[d] * b.len[d];
  for (int k = 0; k < 4; ++k) quat[4 * row + k] = quat[4 * i + k];
}

// Comm::reverse_comm for the periodic images of one rank.  The images of an owner are consecutive ghost rows (they are
// created owner by owner, fill_ghosts_kernel): the lane of an owner's FIRST image adds the whole run, in order, with
// plain adds — one writer per owner row, no atomics, the same bits every run (the deterministic mode of the pair
// kernel relies on it for reproducible trajectories; an owner has at most 7 images).
__global__ __launch_bounds__(kStepBlock) void reverse_kernel(int nlocal, int nghost, const int* __restrict__ gowner,
                                                             double* __restrict__ f, double* __restrict__ tq)
{
  const int g = blockIdx.x * kStepBlock + threadIdx.x;
  if (g >= nghost) return;
  const int i = gowner[g];
  if (g > 0 && gowner[g - 1] == i) return;   // not the first image of its owner
  double a[3] = {0.0, 0.0, 0.0}, t[3] = {0.0, 0.0, 0.0};
  for (int h = g; h < nghost && gowner[h] == i; ++h) {
    const int row = nlocal + h;
    for (int k = 0; k < 3; ++k) {
      a[k] += f[3 * row + k];
      t[k] += tq[3 * row + k];
    }
  }
  for (int k = 0; k < 3; ++k) {
    f[3 * i + k] += a[k];
    tq[3 * i + k] += t[k];
  }
}

// ---------------------------------------------------------------- SPEC §7: bins and half list
__device__ inline int cell_of(const BoxParams& b, const double* __restrict__ x, int i)
{
  int c[3];
  for (int d = 0; d < 3; ++d) {
    const double t = (x[3 * i + d] - b.glo[d]) * b.binv[d];
    int k = (t > 0.0) ? (int)__builtin_fmin(t, 2.0e9) : 0;  // NaN -> 0
    c[d] = (k < b.nc[d]) ? k : b.nc[d] - 1;
  }
  return (c[2] * b.nc[1] + c[1]) * b.nc[0] + c[0];
}

__global__ __launch_bounds__(kStepBlock) void bin_count_kernel(int nall, BoxParams b, const double* __restrict__ x,
                                                               int* __restrict__ cell, int* __restrict__ count)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  if (i >= nall) return;
  const int c = cell_of(b, x, i);
  cell[i] = c;
  atomicAdd(&count[c], 1);
}

__global__ __launch_bounds__(kStepBlock) void bin_fill_kernel(int nall, const int* __restrict__ cell,
                                                              const int* __restrict__ start, int* __restrict__ cursor,
                                                              int* __restrict__ atoms)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  if (i >= nall) return;
  const int c = cell[i];
  atoms[start[c] + atomicAdd(&cursor[c], 1)] = i;
}

// FILL == false: nn[i] = number of listed j. FILL == true: writes the row (sorted by j) at offs[i].
template <bool FILL>
__global__ __launch_bounds__(kStepBlock) void half_list_kernel(int nlocal, int nall, BoxParams b, double skin,
                                                               const double* __restrict__ x, const int* __restrict__ shtype,
                                                               const int* __restrict__ tag, const int* __restrict__ gowner,
                                                               const double* __restrict__ mass, int nshapes,
                                                               const int* __restrict__ cell, const int* __restrict__ start,
                                                               const int* __restrict__ atoms, int* __restrict__ nn,
                                                               const int* __restrict__ offs, int* __restrict__ pair_i,
                                                               int* __restrict__ pair_j, int* __restrict__ flags)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  if (i >= nlocal) return;
  const int sti = shtype[i];
  if ((unsigned)sti >= (unsigned)nshapes) {
    atomicOr(flags, kErrShape);
    if (!FILL) nn[i] = 0;
    return;
  }
  const double xi[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
  const double ri = mass[(size_t)kMassStride * sti + 11] + skin;  // column 11: bounding radius
  const int ti = tag ? tag[i] : i;
  const int c = cell[i];
  const int cx = c % b.nc[0], cy = (c / b.nc[0]) % b.nc[1], cz = c / (b.nc[0] * b.nc[1]);
  int cntr = 0;
  const int base = FILL ? offs[i] : 0;
  for (int dz = -1; dz <= 1; ++dz) {
    const int z = cz + dz;
    if (z < 0 || z >= b.nc[2]) continue;
    for (int dy = -1; dy <= 1; ++dy) {
      const int y = cy + dy;
      if (y < 0 || y >= b.nc[1]) continue;
      for (int dx = -1; dx <= 1; ++dx) {
        const int xx = cx + dx;
        if (xx < 0 || xx >= b.nc[0]) continue;
        const int cc = (z * b.nc[1] + y) * b.nc[0] + xx;
        const int e = start[cc + 1];
        for (int p = start[cc]; p < e; ++p) {
          const int j = atoms[p];
          const int tj = tag ? tag[j] : (j < nlocal ? j : gowner[j - nlocal]);
          if (!(ti < tj)) continue;
          const int stj = shtype[j];
          if ((unsigned)stj >= (unsigned)nshapes) {
            atomicOr(flags, kErrShape);
            continue;
          }
          const double ddx = xi[0] - x[3 * j], ddy = xi[1] - x[3 * j + 1], ddz = xi[2] - x[3 * j + 2];
          const double cut = ri + mass[(size_t)kMassStride * stj + 11];
          if (ddx * ddx + ddy * ddy + ddz * ddz < cut * cut) {
            if (FILL) {
              // insertion into the sorted row
              int q = base + cntr;
              while (q > base && pair_j[q - 1] > j) {
                pair_j[q] = pair_j[q - 1];
                --q;
              }
              pair_j[q] = j;
              pair_i[base + cntr] = i;
            }
            ++cntr;
          }
        }
      }
    }
  }
  if (!FILL) nn[i] = cntr;
}

// Interior / boundary partition of the expanded half list (option "halo_overlap"): slots whose j is an owned atom
// first, slots with a ghost j behind them, both in their list order (a stable partition: flag -> exclusive scan ->
// scatter).  The halo loop runs the interior slots while the forward exchange is in flight (shhalo_api.hip).
__global__ __launch_bounds__(kStepBlock) void part_flag_kernel(int np, int nlocal, const int* __restrict__ pair_j, int* __restrict__ flag)
{
  const int w = blockIdx.x * kStepBlock + threadIdx.x;
  if (w < np) flag[w] = pair_j[w] < nlocal ? 1 : 0;
}
__global__ __launch_bounds__(kStepBlock) void part_scatter_kernel(int np, int nlocal, const int* __restrict__ pair_i,
                                                                  const int* __restrict__ pair_j, const int* __restrict__ scan,
                                                                  int* __restrict__ out_i, int* __restrict__ out_j)
{
  const int w = blockIdx.x * kStepBlock + threadIdx.x;
  if (w >= np) return;
  const int j = pair_j[w], before = scan[w];   // interior slots in front of w
  const int dst = (j < nlocal) ? before : scan[np] + (w - before);
  out_i[dst] = pair_i[w];
  out_j[dst] = j;
}

// Verlet::force_clear: f and torque of n atoms (3 n doubles each, 16-byte aligned arrays of an even number of doubles or
// not: handled per double) in ONE launch — two hipMemsetAsync are four fill kernels of ~5 us each in front of every
// pair compute (rocprofv3 trace of a one-rank timestep, profiles/r05_r_step_trace.txt).
__global__ __launch_bounds__(kStepBlock) void force_clear_kernel(const long long n3, double* __restrict__ f, double* __restrict__ torque)
{
  const long long stride = (long long)gridDim.x * kStepBlock;
  for (long long k = (long long)blockIdx.x * kStepBlock + threadIdx.x; k < n3; k += stride) {
    f[k] = 0.0;
    torque[k] = 0.0;
  }
}

__global__ __launch_bounds__(kStepBlock) void copy_x_kernel(int n, const double* __restrict__ x, double* __restrict__ xhold)
{
  const int k = blockIdx.x * kStepBlock + threadIdx.x;
  if (k < 3 * n) xhold[k] = x[k];
}

__global__ __launch_bounds__(kStepBlock) void check_distance_kernel(int n, const double* __restrict__ x,
                                                                    const double* __restrict__ xhold, double trigger2,
                                                                    int* __restrict__ flag)
{
  const int i = blockIdx.x * kStepBlock + threadIdx.x;
  bool moved = false;
  if (i < n) {
    const double dx = x[3 * i] - xhold[3 * i], dy = x[3 * i + 1] - xhold[3 * i + 1], dz = x[3 * i + 2] - xhold[3 * i + 2];
    moved = !(dx * dx + dy * dy + dz * dz <= trigger2);  // NaN counts as moved
  }
  if (__ballot(moved) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

}  // namespace shp
