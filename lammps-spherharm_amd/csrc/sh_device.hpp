// sh_device.hpp — device-side spherical-harmonic radius evaluation for gfx950.
//
// The body-frame evaluation of particle j (kernel family 0 of pair_kernel.hpp: very small n_q, the weighted rule, the
// run-time orders; particle i is evaluated from the ring tables there).  The default family for the compiled orders
// does not come through here: all points a pair evaluates r_j at lie on the rule's azimuths about the line of
// centres, and r_j becomes a pair of polynomials per azimuth (pair_kernel.hpp, jpoly_build).
//
// docs/SPEC.md §1 in the angle-free polynomial form: for a unit vector (x,y,z)
//   r = sum_m Re[ W_m(z) (x+iy)^m ],  W_m(z) = sum_n (2-delta_m0) a_nm Pi_n^m(z),
// W_m a complex polynomial of degree L-m in z.
//  * compiled orders (L <= 12): W_m in monomial form, Horner (see sh_term below);
//  * run-time orders (13..20): W_m = sum_n Q_n^m(z) cw_nm with the two-term recurrence
//      Q_m^m = 1, Q_{m+1}^m = a'_{m+1,m} z, Q_n^m = a'_nm z Q_{n-1}^m - Q_{n-2}^m
//    (Pi_n^m = s_nm Pi_m^m Q_n^m; s_nm Pi_m^m folded into cw on the host; a' in the rc table).
//
// Where the operands live: one wavefront per pair, so the coefficients are
// wave-uniform and arrive as scalar loads (s_load_dwordx16) through the scalar
// data cache into SGPRs, four complex terms at a time, software pipelined —
// no VGPR, no LDS.
//
// Table layout (built on the host by sh_tables.cpp), m-major so that the
// entries of one m-block are contiguous: k = sh_index(L, n, m)
//   monomial table: entry k = coefficient of z^(L-n) of W_m      (compiled orders)
//   rc[k]                 : n == m -> 1 ; n > m -> a'_nm            (run-time orders)
//   cw[2k], cw[2k+1]      : (2-delta_m0) a_nm s_nm Pi_m^m
// Reference: the SH math helpers of the reference are ABSENT FROM MOUNT
// (SURVEY.md §2.2); this is the build's own formulation.
#pragma once
#include <hip/hip_runtime.h>

#include "sh_const.hpp"

namespace shp {

// Makes a wave-uniform pointer opaque to the optimiser. The tables are loop
// invariant, so without this LICM hoists every scalar load out of the node
// loop and the register allocator spills ~170 SGPR pairs into VGPR lanes
// (v_writelane/v_readlane, 256+ VGPRs, 1 wave/SIMD).
// The result is typed as a constant-address-space (AS4) pointer: after the asm
// the compiler no longer knows the pointer is global, and a generic pointer
// would be read with per-lane flat_load into VGPRs instead of s_load.
typedef const double __attribute__((address_space(4))) * cdptr;
__device__ __forceinline__ cdptr launder_uniform(const double* p)
{
  asm volatile("" : "+s"(p));
  return (cdptr)p;
}

// Shape coefficients travel in chunks of 4 complex terms = one s_load_dwordx16
// through the scalar data cache.  The load is an ordinary AS4 load (so the
// compiler tracks it with a counted s_waitcnt and may keep it in flight), taken
// through the evaluation's laundered table pointer (so it cannot be hoisted out
// of the node loop, while the chunk offset still folds into the instruction).
// Chunks are software pipelined: chunk k+1 is requested before the terms of
// chunk k are computed, across m-block boundaries as well, and a
// sched_barrier after every chunk keeps the request where it was written.
// A chunk may run past the end of its m-block (it then holds the head of the
// next block, unused) and the last one past the end of the shape's table,
// which the host pads (sh_chunk_stride()).
constexpr int kChunk = 4;  // complex terms per scalar load
typedef double sh_d8 __attribute__((ext_vector_type(2 * kChunk)));
typedef sh_d8 sh_d8_u __attribute__((aligned(8)));
typedef const sh_d8_u __attribute__((address_space(4))) * cd8ptr;
__device__ __forceinline__ sh_d8 sload_chunk(const cdptr base, const int off)
{
  return *(cd8ptr)(base + off);
}

struct ShAcc {
  double Wr, Wi;   // W_m (Horner accumulators)
};

// w * z + c with the wave-uniform coefficient c as the SGPR addend of ONE v_fma_f64.
// Written as asm because the compiler prefers the VOP2 form v_fmac_f64 (addend = destination
// = VGPR) and then copies every coefficient into a fresh VGPR pair first: three VALU
// instructions per Horner step instead of one.
__device__ __forceinline__ double horner_step(const double w, const double z, const double c)
{
  double o;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(o) : "v"(w), "v"(z), "s"(c));
  return o;
}

// w + c with the wave-uniform c as the SGPR operand of one v_add_f64.  Without it the compiler contracts the first
// two Horner steps (c0 z) + c1 into an FMA whose addend must be a VGPR, and copies c1 there first: two v_mov_b32 and a
// v_fmac_f64 instead of a v_mul_f64 and a v_add_f64 (26 v_mov_b32 per radius evaluation at L = 6).
__device__ __forceinline__ double sgpr_add(const double w, const double c)
{
  double o;
  asm("v_add_f64 %0, %1, %2" : "=v"(o) : "v"(w), "s"(c));
  return o;
}

// Compiled orders evaluate W_m(z) in MONOMIAL form by Horner (coefficients in descending powers,
// built on the host in long double, sh_tables.cpp: build_monomial): one v_fma_f64 per coefficient
// and accumulator instead of the recurrence's mul + fma + fmac, and no recurrence constants at
// all.  Conditioning is fine for L <= 12 (|Horner - exact| <= 2e-15 at L = 12, 7e-13 at L = 20,
// tools/proto/horner_proto.py), so the run-time-order kernel (L = 13..20) keeps the recurrence.
//
// Position K of block M holds the coefficient of z^(L-M-K), (cr, ci) already in SGPRs.  Only one
// SGPR operand fits a VALU instruction, so the first two steps are a mul and an add.
template <int L, int M, int K>
__device__ __forceinline__ void sh_term(const double cr, const double ci, const double z, ShAcc& s)
{
  constexpr int D = L - M;
  if constexpr (K > D) {
    return;
  } else if constexpr (K == 0) {
    if constexpr (D > 0) {
      s.Wr = cr * z;
      if constexpr (M > 0) s.Wi = ci * z;
    } else {
      s.Wr = cr;   // degree 0: stays wave-uniform
      s.Wi = ci;
    }
  } else if constexpr (K == 1) {
    s.Wr = sgpr_add(s.Wr, cr);              // c0 z + c1
    if constexpr (M > 0) s.Wi = sgpr_add(s.Wi, ci);
  } else {
    s.Wr = horner_step(s.Wr, z, cr);
    if constexpr (M > 0) s.Wi = horner_step(s.Wi, z, ci);
  }
}

struct ShState {
  double ar, ai;   // the running sum over the blocks done so far, see sh_block_end
  ShAcc a;
};

// r = sum_m Re[W_m e^m], e = x + i y, is itself a Horner scheme in e:
//   r = Re[W_0 + e (W_1 + e (W_2 + ... + e W_L))],
// so the blocks are walked from m = L DOWN to 0 and block m costs one complex multiply-add
// acc <- acc e + W_m (4 v_fma_f64; 2 for the last, whose imaginary part is not needed) instead of a power
// recurrence E_{m+1} = E_m e (4) plus the fold r += Re[W_m E_m] (2): 4(L-1)+2 operations instead of 6(L-1)+2.
// W_L is a constant (degree 0 in z): the chain starts from two wave-uniform values.
template <int L, int M>
__device__ __forceinline__ void sh_block_end(const double x, const double y, ShState& t)
{
  const ShAcc& s = t.a;
  if constexpr (M == L) {
    t.ar = s.Wr;
    t.ai = s.Wi;   // L == 0: never read
  } else if constexpr (M > 0) {
    const double nr = fma(t.ar, x, fma(-t.ai, y, s.Wr));
    t.ai = fma(t.ar, y, fma(t.ai, x, s.Wi));
    t.ar = nr;
  } else {
    t.ar = fma(t.ar, x, fma(-t.ai, y, s.Wr));
  }
}

// The chunk that starts at position N0 - M of block M, its coefficients already requested in
// `cur`; requests its successor, computes its terms, recurses.  Blocks in DESCENDING m.
template <int L, int M, int N0>
struct ShStep {
  static __device__ __forceinline__ void run(const cdptr cw_in, const sh_d8 cur, const double x, const double y,
                                             const double z, ShState& t)
  {
    constexpr bool block_done = (N0 + kChunk > L);
    constexpr bool has_next = !block_done || (M >= 1);
    constexpr int Mn = block_done ? M - 1 : M;
    constexpr int Nn = block_done ? M - 1 : N0 + kChunk;
    sh_d8 nxt = cur;
    if constexpr (has_next) nxt = sload_chunk(cw_in, 2 * sh_index(L, Nn, Mn));
    sh_term<L, M, N0 - M>(cur[0], cur[1], z, t.a);
    sh_term<L, M, N0 - M + 1>(cur[2], cur[3], z, t.a);
    sh_term<L, M, N0 - M + 2>(cur[4], cur[5], z, t.a);
    sh_term<L, M, N0 - M + 3>(cur[6], cur[7], z, t.a);
    if constexpr (block_done) sh_block_end<L, M>(x, y, t);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (has_next) ShStep<L, Mn, Nn>::run(cw_in, nxt, x, y, z, t);
  }
};

// ---- run-time order (L < 0): plain loops over the rc / cw tables.
__device__ __forceinline__ double sh_eval_rt(const double* rc_in, const double* cw_in, const int LL, const double x,
                                             const double y, const double z)
{
  double Cm = 1.0, Sm = 0.0, r = 0.0;
  for (int m = 0; m <= LL; ++m) {
    const int o = sh_moff(LL, m);
    const cdptr rc = launder_uniform(rc_in + o);
    const cdptr cw = launder_uniform(cw_in + 2 * o);
    double Wr = cw[0], Wi = cw[1];
    if (m + 1 <= LL) {
      double p2 = 1.0;
      double p1 = rc[1] * z;
      Wr = fma(cw[2], p1, Wr);
      Wi = fma(cw[3], p1, Wi);
      for (int n = m + 2; n <= LL; ++n) {
        const int k = n - m;
        const double p = fma(rc[k], z * p1, -p2);
        Wr = fma(cw[2 * k], p, Wr);
        Wi = fma(cw[2 * k + 1], p, Wi);
        p2 = p1;
        p1 = p;
      }
    }
    r = fma(Wr, Cm, r);
    r = fma(-Wi, Sm, r);
    const double c = fma(Cm, x, -(Sm * y)), sn = fma(Cm, y, Sm * x);
    Cm = c;
    Sm = sn;
  }
  return r;
}

// r at unit (x,y,z).  L >= 0: compile-time order, fully unrolled.  L < 0: run-time order lrt.
template <int L>
__device__ __forceinline__ double sh_eval(const double* rc_in, const double* cw_in, const int lrt, const double x,
                                          const double y, const double z)
{
  double r;
  if constexpr (L >= 0) {
    ShState t;
    t.ar = 0.0;
    t.ai = 0.0;
    const cdptr cwl = launder_uniform(cw_in);
    const sh_d8 first = sload_chunk(cwl, 2 * sh_index(L, L, L));
    __builtin_amdgcn_sched_barrier(0);
    ShStep<L, L, L>::run(cwl, first, x, y, z, t);
    r = t.ar;
  } else {
    r = sh_eval_rt(rc_in, cw_in, lrt, x, y, z);
  }
  // Pin the result here. Without a fixed use the optimiser sinks the whole
  // VALU body of the evaluation into whichever later conditional first reads
  // the result, away from its (immovable) s_load / s_mov statements, and has to
  // carry every SGPR operand there through VGPR lanes.
  asm volatile("" : "+v"(r));
  return r;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// LAMMPS quaternion (w,x,y,z) -> rotation matrix, row-major R[3*a+b], body->space.
__device__ __forceinline__ void quat_to_mat(const double q0, const double q1, const double q2, const double q3,
                                            double* R)
{
  const double w2 = q0 * q0, i2 = q1 * q1, j2 = q2 * q2, k2 = q3 * q3;
  const double twoij = 2.0 * q1 * q2, twoik = 2.0 * q1 * q3, twojk = 2.0 * q2 * q3;
  const double twoiw = 2.0 * q1 * q0, twojw = 2.0 * q2 * q0, twokw = 2.0 * q3 * q0;
  R[0] = w2 + i2 - j2 - k2; R[1] = twoij - twokw;     R[2] = twojw + twoik;
  R[3] = twoij + twokw;     R[4] = w2 - i2 + j2 - k2; R[5] = twojk - twoiw;
  R[6] = twoik - twojw;     R[7] = twojk + twoiw;     R[8] = w2 - i2 - j2 + k2;
}

}  // namespace shp
