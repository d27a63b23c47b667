// pair_kernel.hpp — the contact kernel: ONE WAVEFRONT PER HALF-LIST PAIR.
//
// docs/SPEC.md §2.  Two phases per pair, both 64 nodes wide:
//   phase 1  the lanes stride the Q = 2 nq^2 cap nodes in slabs of 64: r_i at
//            the node, the surface point in j's frame, r_j there -> inside?
//            Inside nodes are appended to a per-wave LDS queue (ballot + mbcnt).
//   phase 2  whenever 64 inside nodes are queued (and once for the rest), one
//            node per lane: the inner radius by safeguarded secant (overlap
//            volume) and the surface gradient of i (vector area, torque arm).
// Only ~30 % of the cap nodes of a packed bed are inside the neighbour, so
// queueing them keeps the two expensive passes at full lane occupancy.
// Per-pair data is wave-uniform: shape coefficients arrive as scalar loads,
// recurrence constants as s_mov immediates, the pair frame sits in LDS.  The
// seven integrals (V, S_n, T_n) are reduced with cross-lane shuffles and lane
// 0 applies the force law and issues the FP64 atomics.
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
  unsigned long long* dbg;  // SHP_STATS builds only: work counters (tools/kernel_stats.py)
};

constexpr int kWavesPerBlock = 4;
#ifndef SHP_MIN_WAVES
#define SHP_MIN_WAVES 4  // waves per SIMD the register allocator must leave room for (<= 128 VGPRs)
#endif

// docs/SPEC.md §2.6: residual below which the inverse-quadratic extrapolation is accepted
#ifndef SHP_TAU3
#define SHP_TAU3 1e-4
#endif

// Per-wave LDS: the pair frame (everything per pair the node loops need, kept
// out of VGPRs) and the queue of inside nodes waiting for phase 2.
constexpr int kQueue = 128;  // entries; a slab adds <= 64 to a queue holding < 64
struct WaveLds {
  double frame[36];      // see FR_* below
  double qri[kQueue];    // r_i at the node
  double qrj[kQueue];    // r_j at the node's surface point (root-finder start)
  int qp[kQueue];        // node index p = k * npsi + l
};
enum { FR_BI1 = 0, FR_BI2 = 3, FR_BIC = 6, FR_BJ1 = 9, FR_BJ2 = 12, FR_BJC = 15, FR_DJ = 18, FR_RMI = 21, FR_D = 30 };

__device__ __forceinline__ unsigned launder_u32(unsigned v)
{
  asm volatile("" : "+v"(v));
  return v;
}

// 1/sqrt(x) to the last ulp or two: v_rsq_f64 (2^-26) + two Newton steps.
// Half the VALU work of sqrt() followed by a division.
__device__ __forceinline__ double rsqrt_nr(const double x)
{
  double y = __builtin_amdgcn_rsq(x);
  double h = fma(-x * y, y, 1.0);
  y = fma(y * 0.5, h, y);
  h = fma(-x * y, y, 1.0);
  y = fma(y * 0.5, h, y);
  return y;
}

// 1/d to the last ulp or two: v_rcp_f64 + two Newton steps (5 VALU ops instead of
// the ~12 of an IEEE division); 0 and denormals give inf/NaN, which the callers test.
__device__ __forceinline__ double rcp_nr(const double d)
{
  double r = __builtin_amdgcn_rcp(d);
  r = fma(fma(-d, r, 1.0), r, r);
  r = fma(fma(-d, r, 1.0), r, r);
  return r;
}

template <int L, bool NEEDV>
__global__ void __launch_bounds__(64 * kWavesPerBlock, SHP_MIN_WAVES) pair_contact_kernel(const PairParams P)
{
  __shared__ WaveLds lds_all[kWavesPerBlock];
  const int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWavesPerBlock)) + wib;
  if (w >= P.npairs) return;
  WaveLds& lds = lds_all[wib];
  // The frame is loop invariant: a plain LDS load would be hoisted out of the
  // node loops and pinned in ~70 VGPRs, which is what it is in LDS to avoid.
  // Each loop iteration therefore re-derives its frame pointer from a byte
  // offset laundered through an empty asm (an integer, so that the compiler
  // still sees an LDS address and emits ds_read, not flat loads).
  const unsigned frame_off = (unsigned)(wib * sizeof(WaveLds));
#define SHP_FRAME() ((const double*)((const char*)&lds_all[0] + launder_u32(frame_off)))
  const double* fr = SHP_FRAME();

  const int i = P.pair_i[w];
  const int j = P.pair_j[w];
  const int si = P.shtype[i], sj = P.shtype[j];
  const double Ri = P.rmax[si], Rj = P.rmax[sj];
  double rho2, rho, cosa;
  {
    const double d0 = P.x[3 * j] - P.x[3 * i], d1 = P.x[3 * j + 1] - P.x[3 * i + 1],
                 d2 = P.x[3 * j + 2] - P.x[3 * i + 2];
    rho2 = d0 * d0 + d1 * d1 + d2 * d2;
    rho = sqrt(rho2);
    if (rho >= Ri + Rj) return;  // SPEC §2.1, wave-uniform

    // SPEC §2.2 cap
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

    // the cap frame in both body frames (b?1 = R^T e1, b?2 = R^T e2, b?c = R^T c), d in j's frame
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        lds.frame[FR_BI1 + a] = Rmi[a] * e10 + Rmi[3 + a] * e11 + Rmi[6 + a] * e12;
        lds.frame[FR_BI2 + a] = Rmi[a] * e20 + Rmi[3 + a] * e21 + Rmi[6 + a] * e22;
        lds.frame[FR_BIC + a] = Rmi[a] * c0 + Rmi[3 + a] * c1 + Rmi[6 + a] * c2;
        lds.frame[FR_BJ1 + a] = Rmj[a] * e10 + Rmj[3 + a] * e11 + Rmj[6 + a] * e12;
        lds.frame[FR_BJ2 + a] = Rmj[a] * e20 + Rmj[3 + a] * e21 + Rmj[6 + a] * e22;
        lds.frame[FR_BJC + a] = Rmj[a] * c0 + Rmj[3 + a] * c1 + Rmj[6 + a] * c2;
        lds.frame[FR_DJ + a] = Rmj[a] * d0 + Rmj[3 + a] * d1 + Rmj[6 + a] * d2;
      }
#pragma unroll
      for (int a = 0; a < 9; ++a) lds.frame[FR_RMI + a] = Rmi[a];
      lds.frame[FR_D] = d0; lds.frame[FR_D + 1] = d1; lds.frame[FR_D + 2] = d2;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const double* rc = P.rc;
  const double* cwi = P.coef + (size_t)si * P.cstride;
  const double* cwj = P.coef + (size_t)sj * P.cstride;
  const int lrt = P.lmax;
  const double Rj2 = Rj * Rj;

  // SPEC §2.6: centre of i inside j (only possible when rho < Rj)
  bool centre_inside = false;
  if (NEEDV && rho < Rj) {
    double rj, t0, t1, t2;
    const double ir = 1.0 / rho;
    sh_eval<L, false>(rc, cwj, lrt, -fr[FR_DJ] * ir, -fr[FR_DJ + 1] * ir, -fr[FR_DJ + 2] * ir, rj, t0, t1, t2);
    centre_inside = (rho - rj <= 0.0);
  }

  const int nq = P.nq;
  const int npsi = 2 * nq;
  const int Q = nq * npsi;
  // p / npsi for 0 <= p < Q <= 2^15 as a multiply-shift: exact because
  // magic * npsi - 2^24 < npsi <= 256 < 2^24 / 2^15
  const unsigned magic = ((1u << 24) + (unsigned)npsi - 1u) / (unsigned)npsi;
  const int nslabs = (Q + 63) >> 6;
  const double hw = 0.5 * (1.0 - cosa), hm = 0.5 * (1.0 + cosa);
  const double dpsi = 6.283185307179586476925286766559 / (double)npsi;

  double aV = 0.0, aS0 = 0.0, aS1 = 0.0, aS2 = 0.0, aT0 = 0.0, aT1 = 0.0, aT2 = 0.0;
  int qhead = 0, qcount = 0, slab = 0;  // wave-uniform

  for (;;) {
    // ---------------------------------------------------------------- phase 1
    // classify slabs of 64 cap nodes until 64 inside nodes are queued
    while (qcount < 64 && slab < nslabs) {
      fr = SHP_FRAME();
      const int p = (slab << 6) + lane;
      ++slab;
      const bool valid = p < Q;
      const int k = valid ? (int)(((unsigned)p * magic) >> 24) : 0;
      const int l = valid ? p - k * npsi : 0;
      const double mu = fma(hw, P.glt[k], hm);
      const double sig = sqrt(fmax(0.0, fma(-mu, mu, 1.0)));
      const double a1 = sig * P.cpsi[l], a2 = sig * P.spsi[l];
      // node direction in i's body frame and r_i there
      const double ui0 = fma(a1, fr[FR_BI1], fma(a2, fr[FR_BI2], mu * fr[FR_BIC]));
      const double ui1 = fma(a1, fr[FR_BI1 + 1], fma(a2, fr[FR_BI2 + 1], mu * fr[FR_BIC + 1]));
      const double ui2 = fma(a1, fr[FR_BI1 + 2], fma(a2, fr[FR_BI2 + 2], mu * fr[FR_BIC + 2]));
      double ri, t0, t1, t2;
      sh_eval<L, false>(rc, cwi, lrt, ui0, ui1, ui2, ri, t0, t1, t2);
      // the surface point seen from x_j, in j's body frame
      const double uj0 = fma(a1, fr[FR_BJ1], fma(a2, fr[FR_BJ2], mu * fr[FR_BJC]));
      const double uj1 = fma(a1, fr[FR_BJ1 + 1], fma(a2, fr[FR_BJ2 + 1], mu * fr[FR_BJC + 1]));
      const double uj2 = fma(a1, fr[FR_BJ1 + 2], fma(a2, fr[FR_BJ2 + 2], mu * fr[FR_BJC + 2]));
      const double q0 = fma(ri, uj0, -fr[FR_DJ]), q1 = fma(ri, uj1, -fr[FR_DJ + 1]),
                   q2 = fma(ri, uj2, -fr[FR_DJ + 2]);
      const double s2 = q0 * q0 + q1 * q1 + q2 * q2;
      const bool cand = valid && (s2 < Rj2);
#ifdef SHP_STATS
      if (lane == 0) atomicAdd(&P.dbg[0], 1ULL);
      if (cand) atomicAdd(&P.dbg[1], 1ULL);
      { const bool a_ = __any(cand); if (lane == 0 && a_) atomicAdd(&P.dbg[2], 1ULL); }
#endif
      if (!__any(cand)) continue;  // wave-uniform: the whole 64-node slab misses B_j

      const bool szero = !(s2 > 0.0);
      const double inv = szero ? 0.0 : rsqrt_nr(s2);
      double rj0;
      sh_eval<L, false>(rc, cwj, lrt, szero ? 0.0 : q0 * inv, szero ? 0.0 : q1 * inv, szero ? 1.0 : q2 * inv, rj0,
                        t0, t1, t2);
      if (szero) rj0 = Rj;
      // s < r_j  <=>  s2 < r_j^2 (both non-negative); SPEC: inside iff s < r_j, s == 0 is inside
      const bool inside = cand && (szero || s2 * inv < rj0);
      const unsigned long long m = __ballot(inside);
#ifdef SHP_STATS
      if (inside) atomicAdd(&P.dbg[3], 1ULL);
#endif
      if (m == 0ULL) continue;
      if (inside) {
        const int pos = (qhead + qcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) & (kQueue - 1);
        lds.qp[pos] = p;
        lds.qri[pos] = ri;
        lds.qrj[pos] = rj0;
      }
      qcount += __builtin_popcountll(m);
    }
    if (qcount == 0) break;

    // ---------------------------------------------------------------- phase 2
    // up to 64 queued inside nodes, one per lane: inner radius, then gradient
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    fr = SHP_FRAME();
    const int cnt = qcount < 64 ? qcount : 64;
    const bool active = lane < cnt;
    const int e = (qhead + (active ? lane : 0)) & (kQueue - 1);
    qhead = (qhead + cnt) & (kQueue - 1);
    qcount -= cnt;
#ifdef SHP_STATS
    if (lane == 0) atomicAdd(&P.dbg[4], 1ULL);
    if (active) atomicAdd(&P.dbg[7], 1ULL);
#endif
    const int p = lds.qp[e];
    const double ri = lds.qri[e];
    const int k = (int)(((unsigned)p * magic) >> 24);
    const int l = p - k * npsi;
    const double mu = fma(hw, P.glt[k], hm);
    const double sig = sqrt(fmax(0.0, fma(-mu, mu, 1.0)));
    const double omi = active ? hw * P.glw[k] * dpsi : 0.0;
    const double a1 = sig * P.cpsi[l], a2 = sig * P.spsi[l];

    double rin = 0.0;
    if (NEEDV) {
      // SPEC §2.6 inner radius by safeguarded secant, all lanes in lock step
      const double rj0 = lds.qrj[e];
      const double uj0 = fma(a1, fr[FR_BJ1], fma(a2, fr[FR_BJ2], mu * fr[FR_BJC]));
      const double uj1 = fma(a1, fr[FR_BJ1 + 1], fma(a2, fr[FR_BJ2 + 1], mu * fr[FR_BJC + 1]));
      const double uj2 = fma(a1, fr[FR_BJ1 + 2], fma(a2, fr[FR_BJ2 + 2], mu * fr[FR_BJC + 2]));
      bool act = active && !centre_inside;
      // three most recent points: (xa,ga) oldest, (xb,gb), (lam,gl) newest
      double lo = 0.0, hi = ri, lam, xa = ri, ga, xb = ri, gb;
      {
        const double dj0 = fr[FR_DJ], dj1 = fr[FR_DJ + 1], dj2 = fr[FR_DJ + 2];
        const double bp = uj0 * dj0 + uj1 * dj1 + uj2 * dj2;
        if (!(rho < Rj)) lo = bp - sqrt(fmax(0.0, fma(bp, bp, -(rho2 - Rj2))));
        lam = bp - sqrt(fmax(0.0, fma(bp, bp, -(rho2 - rj0 * rj0))));
        if (!(lam > lo && lam < hi)) lam = 0.5 * (lo + hi);
        const double q0 = fma(ri, uj0, -dj0), q1 = fma(ri, uj1, -dj1), q2 = fma(ri, uj2, -dj2);
        ga = gb = sqrt(q0 * q0 + q1 * q1 + q2 * q2) - rj0;
      }
      const double tolx = 1e-14 * Rj;
      if (!act) lam = ri;
      for (int it = 0; it < 60; ++it) {
        if (!__any(act)) break;
        fr = SHP_FRAME();
#ifdef SHP_STATS
        if (lane == 0) atomicAdd(&P.dbg[5], 1ULL);
        if (act) atomicAdd(&P.dbg[6], 1ULL);
#endif
        const double y0 = fma(lam, uj0, -fr[FR_DJ]), y1 = fma(lam, uj1, -fr[FR_DJ + 1]),
                     y2 = fma(lam, uj2, -fr[FR_DJ + 2]);
        const double ss2 = y0 * y0 + y1 * y1 + y2 * y2;
        const bool z0 = !(ss2 > 0.0);
        const double iv = z0 ? 0.0 : rsqrt_nr(ss2);
        double rj, t0, t1, t2;
        sh_eval<L, false>(rc, cwj, lrt, z0 ? 0.0 : y0 * iv, z0 ? 0.0 : y1 * iv, z0 ? 1.0 : y2 * iv, rj, t0, t1, t2);
        const double gl = z0 ? -Rj : ss2 * iv - rj;
        if (act) {
          if (gl >= 0.0) lo = lam; else hi = lam;
          const bool have3 = it >= 1;
          const double dbl = gb - gl;
          const double sec = fma(gl * (lam - xb), rcp_nr(dbl), lam);
          double ext = sec;  // extrapolation to g = 0 through all known points
          if (have3) {       // inverse quadratic interpolation over one common denominator
            const double dab = ga - gb, dal = ga - gl;
            const double num = fma(xa * gb, gl * dbl, fma(lam * ga, gb * dab, -(xb * ga) * (gl * dal)));
            ext = num * rcp_nr(dab * dal * dbl);
          }
          if (!(fabs(ext) <= 1e300)) ext = sec;
          if (fabs(gl) <= (have3 ? SHP_TAU3 : 1e-7) * Rj) {  // accept the extrapolated point
            rin = (fabs(ext) <= 1e300) ? fmin(fmax(ext, lo), hi) : lam;
            act = false;
          } else {
            double nxt = ext;
            if (!(nxt > lo && nxt < hi)) nxt = sec;
            if (!(nxt > lo && nxt < hi)) nxt = 0.5 * (lo + hi);
            if (hi - lo <= tolx) {
              rin = 0.5 * (lo + hi);
              act = false;
            } else {
              rin = nxt;
              xa = xb; ga = gb; xb = lam; gb = gl; lam = nxt;
            }
          }
        }
      }
      aV = fma(omi * (1.0 / 3.0), ri * ri * ri - rin * rin * rin, aV);
    }

    // surface gradient of i at the node (the value it recomputes is bit-identical to ri)
    fr = SHP_FRAME();
    const double ui0 = fma(a1, fr[FR_BI1], fma(a2, fr[FR_BI2], mu * fr[FR_BIC]));
    const double ui1 = fma(a1, fr[FR_BI1 + 1], fma(a2, fr[FR_BI2 + 1], mu * fr[FR_BIC + 1]));
    const double ui2 = fma(a1, fr[FR_BI1 + 2], fma(a2, fr[FR_BI2 + 2], mu * fr[FR_BIC + 2]));
    double ri2, g0, g1, g2;
    sh_eval<L, true>(rc, cwi, lrt, ui0, ui1, ui2, ri2, g0, g1, g2);
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
    // the queue slots just read may be overwritten by the next phase 1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  aS0 = wave_sum(aS0); aS1 = wave_sum(aS1); aS2 = wave_sum(aS2);
  aT0 = wave_sum(aT0); aT1 = wave_sum(aT1); aT2 = wave_sum(aT2);
  if (NEEDV) aV = wave_sum(aV);
  if (lane != 0) return;
  fr = SHP_FRAME();
#undef SHP_FRAME

  // rotate the body-frame integrals of i to the space frame
  const double S0 = fr[FR_RMI + 0] * aS0 + fr[FR_RMI + 1] * aS1 + fr[FR_RMI + 2] * aS2;
  const double S1 = fr[FR_RMI + 3] * aS0 + fr[FR_RMI + 4] * aS1 + fr[FR_RMI + 5] * aS2;
  const double S2 = fr[FR_RMI + 6] * aS0 + fr[FR_RMI + 7] * aS1 + fr[FR_RMI + 8] * aS2;
  const double T0 = fr[FR_RMI + 0] * aT0 + fr[FR_RMI + 1] * aT1 + fr[FR_RMI + 2] * aT2;
  const double T1 = fr[FR_RMI + 3] * aT0 + fr[FR_RMI + 4] * aT1 + fr[FR_RMI + 5] * aT2;
  const double T2 = fr[FR_RMI + 6] * aT0 + fr[FR_RMI + 7] * aT1 + fr[FR_RMI + 8] * aT2;
  const double d0 = fr[FR_D], d1 = fr[FR_D + 1], d2 = fr[FR_D + 2];

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
