// pair_kernel.hpp — the contact kernel: ONE WAVEFRONT PER HALF-LIST PAIR.
//
// docs/SPEC.md §2.  The 64 lanes stride the Q = 2 nq^2 cap-quadrature nodes;
// everything that is per pair (centres, rotation matrices, cap frame, shape
// coefficients) is wave-uniform and lives in SGPRs / scalar-cache loads, the
// seven integrals (V, S_n, T_n) are reduced across the wave with cross-lane
// shuffles and lane 0 applies the force law and issues the FP64 atomics.
// No MFMA: the work is a polynomial recurrence per node, FP64 VALU bound.
//
// Reference: PairSH::compute() of the reference is ABSENT FROM MOUNT
// (/root/reference/README.md:1 is the whole mount).
#pragma once
#include "sh_device.hpp"

namespace shp {

struct PairParams {
  // atoms (device)
  const double* x;
  const double* quat;
  const int* type;
  const int* shtype;
  double* f;
  double* torque;
  // half list, expanded: one (i, j) per slot
  const int* pair_i;
  const int* pair_j;
  int npairs;
  int nlocal;
  int newton_pair;
  // shape tables
  const double* rc;     // recurrence constants for lmax (sh_device.hpp)
  const double* coef;   // nshapes x cstride doubles (cw)
  const double* rmax;   // nshapes
  int cstride;
  int lmax;
  // pair coefficients, (ntypes+1)^2 row-major
  const double* kn;
  const double* expo;
  int ntypes;
  // quadrature tables
  const double* glt;    // nq Gauss-Legendre nodes on [-1,1]
  const double* glw;    // nq weights
  const double* cpsi;   // 2nq cos(psi_l)
  const double* spsi;   // 2nq sin(psi_l)
  int nq;
  // outputs / flags
  double* ev;           // 7 doubles or null
  double* pair_out;     // 7 doubles per slot or null
  unsigned char* flags;  // per slot: 1 = contact pair, 2 = touching pair; or null (stats only)
  int eflag;
  int vflag;
  int force_volume;
};

constexpr int kWavesPerBlock = 4;
#ifndef SHP_MIN_WAVES
#define SHP_MIN_WAVES 2  // waves per SIMD the register allocator must leave room for (<= 256 VGPRs)
#endif

template <int L, bool NEEDV>
__global__ void __launch_bounds__(64 * kWavesPerBlock, SHP_MIN_WAVES) pair_contact_kernel(const PairParams P)
{
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)));
  if (w >= P.npairs) return;

  const int i = P.pair_i[w];
  const int j = P.pair_j[w];
  const int si = P.shtype[i], sj = P.shtype[j];
  const double Ri = P.rmax[si], Rj = P.rmax[sj];
  const double xi0 = P.x[3 * i], xi1 = P.x[3 * i + 1], xi2 = P.x[3 * i + 2];
  const double d0 = P.x[3 * j] - xi0, d1 = P.x[3 * j + 1] - xi1, d2 = P.x[3 * j + 2] - xi2;
  const double rho2 = d0 * d0 + d1 * d1 + d2 * d2;
  const double rho = sqrt(rho2);
  if (rho >= Ri + Rj) return;  // SPEC §2.1, wave-uniform

  // SPEC §2.2 cap
  double cosa;
  if (rho <= Rj) cosa = -1.0;
  else if (rho2 - Rj * Rj <= Ri * Ri) cosa = sqrt(rho2 - Rj * Rj) / rho;
  else cosa = (rho2 + Ri * Ri - Rj * Rj) / (2.0 * rho * Ri);

  // SPEC §2.3 frame (space)
  const double c0 = d0 / rho, c1 = d1 / rho, c2 = d2 / rho;
  const double sg = copysign(1.0, c2);
  const double aa = -1.0 / (sg + c2);
  const double bb = c0 * c1 * aa;
  const double e10 = 1.0 + sg * c0 * c0 * aa, e11 = sg * bb, e12 = -sg * c0;
  const double e20 = bb, e21 = sg + c1 * c1 * aa, e22 = -c1;

  double Rmi[9], Rmj[9];
  quat_to_mat(P.quat[4 * i], P.quat[4 * i + 1], P.quat[4 * i + 2], P.quat[4 * i + 3], Rmi);
  quat_to_mat(P.quat[4 * j], P.quat[4 * j + 1], P.quat[4 * j + 2], P.quat[4 * j + 3], Rmj);

  // the cap frame in both body frames:  b?1 = R^T e1, b?2 = R^T e2, b?c = R^T c
  double bi1[3], bi2[3], bic[3], bj1[3], bj2[3], bjc[3], dj[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    bi1[a] = Rmi[a] * e10 + Rmi[3 + a] * e11 + Rmi[6 + a] * e12;
    bi2[a] = Rmi[a] * e20 + Rmi[3 + a] * e21 + Rmi[6 + a] * e22;
    bic[a] = Rmi[a] * c0 + Rmi[3 + a] * c1 + Rmi[6 + a] * c2;
    bj1[a] = Rmj[a] * e10 + Rmj[3 + a] * e11 + Rmj[6 + a] * e12;
    bj2[a] = Rmj[a] * e20 + Rmj[3 + a] * e21 + Rmj[6 + a] * e22;
    bjc[a] = Rmj[a] * c0 + Rmj[3 + a] * c1 + Rmj[6 + a] * c2;
    dj[a] = Rmj[a] * d0 + Rmj[3 + a] * d1 + Rmj[6 + a] * d2;
  }

  const double* __restrict__ rc = P.rc;
  const double* __restrict__ cwi = P.coef + (size_t)si * P.cstride;
  const double* __restrict__ cwj = P.coef + (size_t)sj * P.cstride;
  const int lrt = P.lmax;
  const double Rj2 = Rj * Rj;

  // SPEC §2.6: centre of i inside j (only possible when rho < Rj)
  bool centre_inside = false;
  if (NEEDV && rho < Rj) {
    double rj, t0, t1, t2;
    sh_eval<L, false>(rc, cwj, lrt, -dj[0] / rho, -dj[1] / rho, -dj[2] / rho, rj, t0, t1, t2);
    centre_inside = (rho - rj <= 0.0);
  }

  const int nq = P.nq;
  const int npsi = 2 * nq;
  const int Q = nq * npsi;
  const double hw = 0.5 * (1.0 - cosa), hm = 0.5 * (1.0 + cosa);
  const double dpsi = 6.283185307179586476925286766559 / (double)npsi;

  double aV = 0.0, aS0 = 0.0, aS1 = 0.0, aS2 = 0.0, aT0 = 0.0, aT1 = 0.0, aT2 = 0.0;

  for (int p0 = 0; p0 < Q; p0 += 64) {
    const int p = p0 + lane;
    const bool valid = p < Q;
    const int k = valid ? p / npsi : 0;
    const int l = valid ? p - k * npsi : 0;
    const double mu = fma(hw, P.glt[k], hm);
    const double sig = sqrt(fmax(0.0, fma(-mu, mu, 1.0)));
    const double om = valid ? hw * P.glw[k] * dpsi : 0.0;
    const double a1 = sig * P.cpsi[l], a2 = sig * P.spsi[l];

    // node direction in i's body frame, radius and gradient of i there
    const double ui0 = fma(a1, bi1[0], fma(a2, bi2[0], mu * bic[0]));
    const double ui1 = fma(a1, bi1[1], fma(a2, bi2[1], mu * bic[1]));
    const double ui2 = fma(a1, bi1[2], fma(a2, bi2[2], mu * bic[2]));
    double ri, g0, g1, g2;
    sh_eval<L, false>(rc, cwi, lrt, ui0, ui1, ui2, ri, g0, g1, g2);

    // the surface point seen from x_j, in j's body frame
    const double uj0 = fma(a1, bj1[0], fma(a2, bj2[0], mu * bjc[0]));
    const double uj1 = fma(a1, bj1[1], fma(a2, bj2[1], mu * bjc[1]));
    const double uj2 = fma(a1, bj1[2], fma(a2, bj2[2], mu * bjc[2]));
    const double q0 = fma(ri, uj0, -dj[0]), q1 = fma(ri, uj1, -dj[1]), q2 = fma(ri, uj2, -dj[2]);
    const double s2 = q0 * q0 + q1 * q1 + q2 * q2;
    const bool cand = valid && (s2 < Rj2);
    if (!__any(cand)) continue;  // wave-uniform: the whole 64-node slab misses B_j

    const double s = sqrt(s2);
    const bool szero = !(s > 0.0);
    const double inv = szero ? 0.0 : 1.0 / s;
    double rj0, t0, t1, t2;
    sh_eval<L, false>(rc, cwj, lrt, szero ? 0.0 : q0 * inv, szero ? 0.0 : q1 * inv, szero ? 1.0 : q2 * inv, rj0,
                      t0, t1, t2);
    if (szero) rj0 = Rj;
    const bool inside = cand && (szero || s < rj0);
    if (!__any(inside)) continue;

    double rin = 0.0;
    if (NEEDV) {
      // SPEC §2.6 inner radius by safeguarded secant, all lanes in lock step
      bool act = inside && !centre_inside;
      const double bp = uj0 * dj[0] + uj1 * dj[1] + uj2 * dj[2];
      double lo = 0.0;
      if (!(rho < Rj)) lo = bp - sqrt(fmax(0.0, fma(bp, bp, -(rho2 - Rj2))));
      double hi = ri;
      double lam = bp - sqrt(fmax(0.0, fma(bp, bp, -(rho2 - rj0 * rj0))));
      if (!(lam > lo && lam < hi)) lam = 0.5 * (lo + hi);
      double lprev = ri, gprev = s - rj0;
      const double tolg = 1e-13 * Rj, tolx = 1e-14 * Rj;
      if (!act) lam = ri;
      for (int it = 0; it < 60; ++it) {
        if (!__any(act)) break;
        const double y0 = fma(lam, uj0, -dj[0]), y1 = fma(lam, uj1, -dj[1]), y2 = fma(lam, uj2, -dj[2]);
        const double ss = sqrt(y0 * y0 + y1 * y1 + y2 * y2);
        const bool z0 = !(ss > 0.0);
        const double iv = z0 ? 0.0 : 1.0 / ss;
        double rj;
        sh_eval<L, false>(rc, cwj, lrt, z0 ? 0.0 : y0 * iv, z0 ? 0.0 : y1 * iv, z0 ? 1.0 : y2 * iv, rj, t0, t1, t2);
        const double gl = z0 ? -Rj : ss - rj;
        if (act) {
          if (gl >= 0.0) lo = lam; else hi = lam;
          rin = lam;
          if (fabs(gl) <= tolg) {
            act = false;
          } else {
            double nxt = lam - gl * (lam - lprev) / (gl - gprev);
            if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
            if (hi - lo <= tolx) {
              rin = 0.5 * (lo + hi);
              act = false;
            } else {
              lprev = lam; gprev = gl; lam = nxt;
            }
          }
        }
      }
    }
    // Only now the surface gradient of i: most 64-node slabs of the bounding
    // cap miss particle j, so the 2x dearer gradient pass is taken on demand
    // (the value it recomputes is bit-identical to ri).  It comes LAST in the
    // iteration: placed before the root-finder loop, the compiler sinks its VALU
    // work below the loop and keeps every SGPR constant alive in VGPR lanes.
    {
      double ri2;
      sh_eval<L, true>(rc, cwi, lrt, ui0, ui1, ui2, ri2, g0, g1, g2);
    }
    const double omi = inside ? om : 0.0;
    if (NEEDV) aV = fma(omi * (1.0 / 3.0), ri * ri * ri - rin * rin * rin, aV);
    // vector area element A = r^2 u - r t, t = grad - (u.grad) u   (body frame of i)
    const double ug = ui0 * g0 + ui1 * g1 + ui2 * g2;
    const double rr = ri * (ri + ug);
    const double A0 = fma(rr, ui0, -ri * g0), A1 = fma(rr, ui1, -ri * g1), A2 = fma(rr, ui2, -ri * g2);
    aS0 = fma(omi, A0, aS0);
    aS1 = fma(omi, A1, aS1);
    aS2 = fma(omi, A2, aS2);
    // (r u) x A, body frame
    const double wr = omi * ri;
    aT0 = fma(wr, ui1 * A2 - ui2 * A1, aT0);
    aT1 = fma(wr, ui2 * A0 - ui0 * A2, aT1);
    aT2 = fma(wr, ui0 * A1 - ui1 * A0, aT2);

  }

  aS0 = wave_sum(aS0); aS1 = wave_sum(aS1); aS2 = wave_sum(aS2);
  aT0 = wave_sum(aT0); aT1 = wave_sum(aT1); aT2 = wave_sum(aT2);
  if (NEEDV) aV = wave_sum(aV);
  if (lane != 0) return;

  // rotate the body-frame integrals of i to the space frame
  const double S0 = Rmi[0] * aS0 + Rmi[1] * aS1 + Rmi[2] * aS2;
  const double S1 = Rmi[3] * aS0 + Rmi[4] * aS1 + Rmi[5] * aS2;
  const double S2 = Rmi[6] * aS0 + Rmi[7] * aS1 + Rmi[8] * aS2;
  const double T0 = Rmi[0] * aT0 + Rmi[1] * aT1 + Rmi[2] * aT2;
  const double T1 = Rmi[3] * aT0 + Rmi[4] * aT1 + Rmi[5] * aT2;
  const double T2 = Rmi[6] * aT0 + Rmi[7] * aT1 + Rmi[8] * aT2;

  if (P.pair_out) {
    double* o = P.pair_out + 7 * (size_t)w;
    o[0] = aV; o[1] = S0; o[2] = S1; o[3] = S2; o[4] = T0; o[5] = T1; o[6] = T2;
  }
  const bool touched = NEEDV ? (aV > 0.0) : (S0 != 0.0 || S1 != 0.0 || S2 != 0.0);
  // statistics go through a byte per slot, summed by count_flags_kernel: one
  // atomic per pair on a shared counter costs more than the whole kernel
  if (P.flags) P.flags[w] = touched ? 2 : 1;
  if (!touched) return;

  // SPEC §2.7 force law
  const int ti = P.type[i], tj = P.type[j];
  const double knij = P.kn[ti * (P.ntypes + 1) + tj];
  const double mij = P.expo[ti * (P.ntypes + 1) + tj];
  const double pn = (mij == 1.0) ? knij : knij * mij * pow(aV, mij - 1.0);
  const double F0 = -pn * S0, F1 = -pn * S1, F2 = -pn * S2;
  const double M0 = -pn * T0, M1 = -pn * T1, M2 = -pn * T2;
  atomicAdd(&P.f[3 * i], F0);
  atomicAdd(&P.f[3 * i + 1], F1);
  atomicAdd(&P.f[3 * i + 2], F2);
  atomicAdd(&P.torque[3 * i], M0);
  atomicAdd(&P.torque[3 * i + 1], M1);
  atomicAdd(&P.torque[3 * i + 2], M2);
  const bool applyj = P.newton_pair || j < P.nlocal;
  if (applyj) {
    // F_j = -F_i ;  tau_j = -tau_i - d x F_j
    const double G0 = -F0, G1 = -F1, G2 = -F2;
    atomicAdd(&P.f[3 * j], G0);
    atomicAdd(&P.f[3 * j + 1], G1);
    atomicAdd(&P.f[3 * j + 2], G2);
    atomicAdd(&P.torque[3 * j], -M0 - (d1 * G2 - d2 * G1));
    atomicAdd(&P.torque[3 * j + 1], -M1 - (d2 * G0 - d0 * G2));
    atomicAdd(&P.torque[3 * j + 2], -M2 - (d0 * G1 - d1 * G0));
  }
  if ((P.eflag || P.vflag) && P.ev) {
    const double share = P.newton_pair ? 1.0 : (0.5 + (j < P.nlocal ? 0.5 : 0.0));
    if (P.eflag) atomicAdd(&P.ev[0], share * knij * pow(aV, mij));
    if (P.vflag) {
      // ev_tally_xyz with del = x_i - x_j = -d and the force on i
      atomicAdd(&P.ev[1], share * (-d0) * F0);
      atomicAdd(&P.ev[2], share * (-d1) * F1);
      atomicAdd(&P.ev[3], share * (-d2) * F2);
      atomicAdd(&P.ev[4], share * (-d0) * F1);
      atomicAdd(&P.ev[5], share * (-d0) * F2);
      atomicAdd(&P.ev[6], share * (-d1) * F2);
    }
  }
}

// Host-callable launcher, one per compiled order (pair_kernels_L*.hip).
typedef void (*pair_launch_fn)(const PairParams&, bool needv, hipStream_t);

template <int L>
void launch_pair_contact(const PairParams& P, bool needv, hipStream_t st)
{
  if (P.npairs <= 0) return;
  const dim3 grid((P.npairs + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
  if (needv) hipLaunchKernelGGL((pair_contact_kernel<L, true>), grid, block, 0, st, P);
  else hipLaunchKernelGGL((pair_contact_kernel<L, false>), grid, block, 0, st, P);
}

}  // namespace shp
