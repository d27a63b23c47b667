// pair_setup.hpp — per-pair scalar set-up of the contact kernel, ONE LANE PER PAIR.
//
// Everything the contact kernel needs that depends on the pair as a whole (docs/SPEC.md §2.1-2.3: bounding-sphere
// reject, cap angle, the frame (e1, e2, c), the cap axes and the separation in j's body frame, the Euler angles of
// the cap frame in i's body frame) is wave-uniform there: computed in the contact kernel it costs ~300 FP64
// instructions per pair ISSUED ON 64 LANES for one lane's worth of work (no scalar FP64 unit on gfx950).  Here the
// same arithmetic runs with one pair per lane — 1/64 of the vector issue — and leaves a 320-byte record per pair that
// the contact kernel picks up with one coalesced load (pair_kernel.hpp).  HBM bound: 56 B gathered per atom of the
// pair, 336 B written per pair.  The records of a workgroup are staged in LDS and leave as full 512-byte wave stores
// (written lane by lane they are 8-byte stores 320 bytes apart: 1.4x the bytes at the memory side, measured).
// Included by shpair_api.hip only (the kernel is not a template: one definition per library).
#pragma once
#include "pair_kernel.hpp"

namespace shp {

constexpr int kSetupBlock = 64;
constexpr int kSetupPad = kRecStride + 1;   // LDS row stride: odd, so that the lanes' rows start in different banks

__device__ __forceinline__ void pair_setup_one(const PairParams& P, const int w, double* __restrict__ o, int* __restrict__ ri);

// cos, sin of the Euler angles of M = [b1 b2 bc] = Rz(alpha) Ry(beta) Rz(gamma) (pair_kernel.hpp cap_frame_rotate), six
// doubles.  sin(beta) from the x,y components of the pole, NOT sqrt(1 - cos^2): near the poles the latter is
// quantised at 1e-8 and rotates by a wrong tilt (4.8e-9 in r at L = 6, caught by tests/test_host_tables.py); gamma
// from the WELL CONDITIONED sum (cos beta >= 0) or difference of the two angles.
__device__ __forceinline__ void euler_zyz(const double* b1, const double* b2, const double* bc, double* __restrict__ o)
{
  const double cb = bc[2];
  const double sb2 = bc[0] * bc[0] + bc[1] * bc[1];
  double sb = 0.0, ca = 1.0, sa = 0.0;
  if (sb2 > 1e-280) {  // below: exactly polar (and v_rsq_f64 would meet a denormal)
    const double n = rsqrt_nr(sb2);
    sb = sb2 * n;
    ca = bc[0] * n;
    sa = bc[1] * n;
  }
  double cg, sgm;
  if (cb >= 0.0) {
    const double iv = rcp_nr(1.0 + cb);
    const double cs = (b1[0] + b2[1]) * iv, ss = (b1[1] - b2[0]) * iv;  // alpha + gamma
    cg = cs * ca + ss * sa;
    sgm = ss * ca - cs * sa;
  } else {
    const double iv = rcp_nr(1.0 - cb);
    const double cd = -(b1[0] - b2[1]) * iv, sd = -(b1[1] + b2[0]) * iv;  // alpha - gamma
    cg = ca * cd + sa * sd;
    sgm = sa * cd - ca * sd;
  }
  o[0] = ca; o[1] = sa;
  o[2] = cb; o[3] = sb;
  o[4] = cg; o[5] = sgm;
}

__global__ __launch_bounds__(kSetupBlock) void pair_setup_kernel(const PairParams P, double* __restrict__ rec,
                                                                  int* __restrict__ rec_i)
{
  __shared__ double srec[kSetupBlock * kSetupPad];
  const int base = P.slot0 + blockIdx.x * kSetupBlock;   // the launch covers the slots [slot0, npairs)
  const int w = base + threadIdx.x;
  if (w < P.npairs) pair_setup_one(P, w, srec + threadIdx.x * kSetupPad, rec_i + 4 * (size_t)w);
  __syncthreads();
  // the workgroup's records, contiguous in memory: 64 x kRecStride doubles as wave-wide stores
  const int nrec = (P.npairs - base < kSetupBlock) ? P.npairs - base : kSetupBlock;
  double* out = rec + (size_t)kRecStride * base;
  for (int t = threadIdx.x; t < nrec * kRecStride; t += kSetupBlock) {
    const int r = t / kRecStride, k = t - r * kRecStride;
    out[t] = srec[r * kSetupPad + k];
  }
}

__device__ __forceinline__ void pair_setup_one(const PairParams& P, const int w, double* __restrict__ o, int* __restrict__ ri)
{
  const int i = P.pair_i[w], j = P.pair_j[w];
  const int si = P.shtype[i], sj = P.shtype[j];
  ri[1] = si;
  ri[2] = sj;
  if ((unsigned)si >= (unsigned)P.nshapes || (unsigned)sj >= (unsigned)P.nshapes) {
    atomicOr(P.err, kPairErrShape);
    ri[0] = 0;
    ri[3] = 0;
    return;
  }
  const double Ri = P.rmax[si], Rj = P.rmax[sj];
  const double d0 = P.x[3 * j] - P.x[3 * i], d1 = P.x[3 * j + 1] - P.x[3 * i + 1], d2 = P.x[3 * j + 2] - P.x[3 * i + 2];
  const double rho2 = d0 * d0 + d1 * d1 + d2 * d2;
  const double rho = sqrt(rho2);  // IEEE: decides the pair (SPEC §2.1) exactly as the oracle does
  if (rho >= Ri + Rj) {
    ri[0] = 0;
    ri[3] = 0;
    return;
  }
  if (!(rho > 0.0)) {   // SPEC §2 step 1: coincident centres (or a separation that is not a number): no line of centres,
    atomicOr(P.err, kPairErrCoincident);   // the pair contributes nothing and is reported
    ri[0] = 0;
    ri[3] = 0;
    return;
  }
  // the force law's operands, for the contact kernel's epilogue (read from its frame: looked up there they are a chain
  // of three dependent table loads — pair_i/j -> type -> kn — at the end of every pair, while the wave still holds
  // all its registers and LDS)
  const int ti = P.type[i], tj = P.type[j];
  if (ti < 1 || ti > P.ntypes || tj < 1 || tj > P.ntypes) {
    atomicOr(P.err, kPairErrType);
    ri[0] = 0;
    ri[3] = 0;
    return;
  }
  o[FR_KN] = P.kn[ti * (P.ntypes + 1) + tj];
  o[FR_EXPO] = P.expo[ti * (P.ntypes + 1) + tj];
  {
    int* ij = (int*)(o + FR_IJ);
    ij[0] = i;
    ij[1] = j;
  }
  ri[0] = 1;
  ri[3] = rho < Rj ? 1 : 0;

  // SPEC §2.2 cap.  The branch conditions are exact; the value only places the nodes, so Newton-refined
  // reciprocals / roots (1-2 ulp) do instead of the IEEE sequences
  const double irho = rcp_nr(rho);
  const double pj = rho2 - Rj * Rj;
  double cosa;
  if (rho <= Rj) cosa = -1.0;
  else if (pj <= Ri * Ri) cosa = sqrt_nr(fmax(pj, 0.0)) * irho;
  else cosa = (pj + Ri * Ri) * (0.5 * irho * rcp_nr(Ri));

  // SPEC §2.3 frame (space)
  const double c0 = d0 * irho, c1 = d1 * irho, c2 = d2 * irho;
  const double sg = copysign(1.0, c2);
  const double aa = -rcp_nr(sg + c2);
  const double bb = c0 * c1 * aa;
  const double e10 = 1.0 + sg * c0 * c0 * aa, e11 = sg * bb, e12 = -sg * c0;
  const double e20 = bb, e21 = sg + c1 * c1 * aa, e22 = -c1;

  double Rmi[9], Rmj[9];
  quat_to_mat(P.quat[4 * i], P.quat[4 * i + 1], P.quat[4 * i + 2], P.quat[4 * i + 3], Rmi);
  quat_to_mat(P.quat[4 * j], P.quat[4 * j + 1], P.quat[4 * j + 2], P.quat[4 * j + 3], Rmj);

  // the cap axes in i's body frame (columns of M) and in j's body frame, d in j's frame
  double b1[3], b2[3], bc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    b1[a] = Rmi[a] * e10 + Rmi[3 + a] * e11 + Rmi[6 + a] * e12;
    b2[a] = Rmi[a] * e20 + Rmi[3 + a] * e21 + Rmi[6 + a] * e22;
    bc[a] = Rmi[a] * c0 + Rmi[3 + a] * c1 + Rmi[6 + a] * c2;
    o[FR_BJ1 + a] = Rmj[a] * e10 + Rmj[3 + a] * e11 + Rmj[6 + a] * e12;
    o[FR_BJ2 + a] = Rmj[a] * e20 + Rmj[3 + a] * e21 + Rmj[6 + a] * e22;
    o[FR_BJC + a] = Rmj[a] * c0 + Rmj[3 + a] * c1 + Rmj[6 + a] * c2;
    o[FR_DJ + a] = Rmj[a] * d0 + Rmj[3 + a] * d1 + Rmj[6 + a] * d2;
  }
  o[FR_E1] = e10; o[FR_E1 + 1] = e11; o[FR_E1 + 2] = e12;
  o[FR_E2] = e20; o[FR_E2 + 1] = e21; o[FR_E2 + 2] = e22;
  o[FR_C] = c0; o[FR_C + 1] = c1; o[FR_C + 2] = c2;
  o[FR_D] = d0; o[FR_D + 1] = d1; o[FR_D + 2] = d2;
  o[FR_RJ] = Rj; o[FR_RJ2] = Rj * Rj; o[FR_RHO2] = rho2;
  const double hw0 = 0.5 * (1.0 - cosa);
  o[FR_HW] = hw0; o[FR_HM] = 0.5 * (1.0 + cosa);
  o[FR_WSC] = hw0 * (6.283185307179586476925286766559 / (double)(2 * P.nq));  // hw dpsi
  o[FR_RHO] = rho;

  euler_zyz(b1, b2, bc, o + FR_EULER);
  if (P.jpoly) {
    // compiled orders: particle j is rotated into the common frame like particle i (pair_kernel.hpp jpoly_build);
    // the Euler angles of M_j = [BJ1 BJ2 BJC] take the slots of BJ1, BJ2, which those kernels never read
    double j1[3], j2[3], jc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      j1[a] = o[FR_BJ1 + a];
      j2[a] = o[FR_BJ2 + a];
      jc[a] = o[FR_BJC + a];
    }
    euler_zyz(j1, j2, jc, o + FR_EULERJ);
    // ... and wave-uniform FP64 products of the inner-radius search take the slots of BJC and d_j: computed in the
    // contact kernel they are loop invariants the compiler parks in vector registers (no scalar FP64 unit)
    o[FR_JPJ] = rho2 - Rj * Rj;
    o[FR_JTOL1] = 1e-7 * Rj;
    o[FR_JTOL3] = SHP_TAU3 * Rj;
    o[FR_JTINY] = 1e-14 * Rj;
  }
}

inline void launch_pair_setup(const PairParams& P, double* rec, int* rec_i, hipStream_t st)
{
  if (P.npairs <= P.slot0) return;
  hipLaunchKernelGGL(pair_setup_kernel, dim3((P.npairs - P.slot0 + kSetupBlock - 1) / kSetupBlock), dim3(kSetupBlock), 0, st, P, rec, rec_i);
}

}  // namespace shp
