// pair_kernel.hpp — the contact kernel: ONE WAVEFRONT PER HALF-LIST PAIR.
//
// docs/SPEC.md §2.  Per pair: a set-up step and two phases, all 64 lanes wide:
//   set-up   particle i's expansion is rotated into the CAP frame (pole = the
//            direction to the neighbour): real-SH coefficients through
//            Z(alpha) X^T Z(beta) X Z(gamma) with the constant matrices
//            X^l = T(Rx(90 deg)) (no Wigner recursion, no trigonometry beyond
//            three sin/cos pairs), then per quadrature ring k and order m the
//            Fourier coefficients A_km, B_km of r_i around the ring and their
//            mu-derivatives.  After that r_i at a cap node costs 6 FP64 ops per
//            m instead of a full (L+1)(L+2)/2-term evaluation, and so does its
//            surface gradient.
//   phase 1  the lanes stride the Q = 2 nq^2 cap nodes in slabs of 64: r_i at
//            the node, the surface point in j's frame, r_j there -> inside?
//            Inside nodes are appended to a per-wave LDS queue (ballot + mbcnt).
//   phase 2  whenever 64 inside nodes are queued (and once for the rest), one
//            node per lane: the inner radius by safeguarded secant (overlap
//            volume) and the surface gradient of i (vector area, torque arm).
// Only ~30 % of the cap nodes of a packed bed are inside the neighbour, so
// queueing them keeps the two expensive passes at full lane occupancy.
// Per-pair data is wave-uniform: particle j's coefficients (monomial form, Horner,
// sh_device.hpp) arrive as scalar loads, the pair frame, the rotation's work vectors,
// the ring tables and the queue sit in per-wave LDS.  The seven integrals (V, S_n,
// T_n) are transposed through LDS and added in two levels; lanes 0-5 then own one
// component of the force / torque each, apply the force law and issue ONE 6-lane FP64
// atomic per atom (or, in the deterministic mode, one store per pair: det_kernels.hpp).
// One wave per workgroup — or, for large tables, two waves per pair (pair_lds_layout2).
// No MFMA: the work is polynomial evaluation per node, FP64 VALU bound.
// Two kernel families differ in how particle j's radius is evaluated (template parameter JPT, chosen per (L, n_q) by
// shpair_api.hip use_jpoly): 0 in j's body frame from scalar-fed monomial coefficients (sh_device.hpp); 1 — the
// default almost everywhere — from per-azimuth polynomials in the pair's common frame, with the coefficient rotations
// of both particles in a kernel of their own (pair_rotate_lane_kernel) and node PAIRS per lane in phase 1: see the
// block comment above jpoly_build.
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
  int npairs;           // end of the slot range of this launch (exclusive); the whole list unless a caller splits it
  int slot0;            // ... and its first slot: a multiple of 32 (rotation tiles hold 64 rotations = 32 slots).  The halo
                        // loop runs the slots whose atoms are all owned before the forward exchange has landed (shhalo_api.hip)
  int nlocal;
  int newton_pair;
  // shape tables
  const double* rc;     // recurrence constants a'_nm for lmax, m-major (ring tables; run-time-order kernel)
  const double* coef;   // nshapes x cstride doubles: monomial table (compiled orders) or cw (run-time order)
  const double* rmax;   // nshapes
  int nshapes;
  int* err;             // device error bits (kPairErr*), raised instead of an out-of-bounds table read
  int cstride;
  int lmax;
  // pair coefficients, (ntypes+1)^2 row-major
  const double* kn;
  const double* expo;
  int ntypes;
  // cap-frame evaluation of particle i (sh_tables.hpp)
  const double* creal;   // nshapes x (lmax+1)^2 real-basis coefficients
  const double* xval;    // X = T(Rx(+90)) and X^T in ELL form: 2 x (lmax+1)^2 rows x (lmax/2+1) values
  const int* xcol;       // ... and absolute column indices
  const int* xinfo;      // (lmax+1)^2: l | (m + l) << 8
  const double* gscale;  // (lmax+1)^2 ring-recurrence scale g_lm
  // particle j in the pair's common frame (compiled orders; jpoly_build below)
  const double* jval;    // first stage, ELL: (2 lmax + 4)(lmax + 1) rows x (lmax/2+1) values (sh_tables.cpp build_jpoly_ell)
  const int* jcol;       // ... and indices into the rotated coefficient vector
  const double* trigj;   // (cos, sin)(m psi_l), m = 0..lmax + 1, of the first nq azimuths, l-major
  const double* rot;     // compiled orders: rotated, scaled coefficient vectors of slot w's particles, rotation 2 w + which
                         // (which 0: i, 1: j), in the tiled layout of rot_index(); written by pair_rotate_lane_kernel
  int jpoly;             // 1: the pair records carry the Euler angles of j's frame in the slots of FR_BJ1 / FR_BJ2
  int split;             // 1: two waves per pair (pair_contact_kernel<..., WPP = 2>); wave_lds_bytes is then the PAIR's LDS
  // per-pair records written by pair_setup_kernel (pair_setup.hpp), read here instead of redoing the scalar set-up on
  // 64 lanes: rec[kRecStride * w] = the pair frame FR_* and the Euler cos/sin; rec_i[4 w] = status, shape i, shape j,
  // [rho < R_j]
  const double* rec;
  const int* rec_i;
  int wave_lds_bytes;    // dynamic LDS per wave (wave_lds_layout)
  int ring_rows;         // quadrature rings whose tables are resident at a time (<= nq)
  int qcap;              // per-azimuth kernels: entries of a wave's node queue (queue_capacity)
  int waves_per_block;
  int spec;              // 1: a launch whose (n_q, ring_rows, qcap) are those of PairSpec<L> takes the specialised instance
  // quadrature tables
  const double* glt;    // nq Gauss-Legendre nodes on [-1,1]
  const double* glw;    // nq weights
  const double* cpsi;   // 2nq cos(psi_l)
  const double* spsi;   // 2nq sin(psi_l)
  int rule;             // 0: sharp inside test (SPEC §2.5); 1: covered-fraction weights (SPEC §2.8)
  double* eatom;        // nullable: per-atom energy  [nall], LAMMPS eatom (ev_tally_xyz halves)
  double* vatom;        // nullable: per-atom virial  [nall][6] (xx,yy,zz,xy,xz,yz)
  const double* trig;   // (cos, sin)(m psi_l), m = 2..lmax; trig_lmajor(lmax): at trig[l * trig_stride + 2 (m - 2)],
                        // else at trig[(m - 2) * trig_stride + 2 l]
  int trig_stride;      // doubles between consecutive azimuths (l-major: 2 (lmax - 1)) / orders (m-major: 4 nq)
  int nq;
  // outputs / flags
  double* ev;           // 7 doubles or null: where tally_reduce_kernel adds the sums of pair_ev (the pair kernels do not touch it)
  double* pair_ev;      // eflag / vflag: 8 doubles per slot, E xx yy zz xy xz yz -, zeroed before the launch; or null
  double* pair_out;     // 7 doubles per slot or null
  double* pair_ft;      // deterministic mode (det_kernels.hpp): 12 doubles per slot, F_i tau_i | F_j tau_j, written instead
                        // of the atomics; null in the default mode
  unsigned char* flags;  // per slot: 1 = contact pair, 2 = touching pair; or null (stats only)
  int eflag;
  int vflag;
  unsigned long long* dbg;  // SHP_STATS builds only: work counters (tools/kernel_stats.py)
};

constexpr int kMaxWavesPerBlock = 4;
constexpr int kPairErrShape = 1;  // a shape index outside [0, nshapes) reached the kernel: the pair was skipped
constexpr int kPairErrType = 2;   // an atom type outside [1, ntypes]
constexpr int kPairErrCoincident = 4;   // two centres coincide (rho = 0) or their separation is not a number: SPEC §2 step 1
// Waves per SIMD the register allocator must leave room for, chosen per kernel so that NO kernel spills a vector
// register or touches scratch (tests/test_kernel_resources.py reads the code objects):
//   forces-only kernels (no root finder) fit 80 VGPRs = 6 waves up to L = 6;
//   kernels with the volume path fit 80 VGPRs at L = 0, 1 and 6 and need 96 (5 waves) in between (at 80 they would
//   spill 1-7 registers; round 1 shipped those with 8-32 bytes of scratch) and from L = 7 on;
//   the weighted variant (three-slab window) fits 96 VGPRs up to L = 6 except at L = 3 (128: 4 waves), 128 beyond.
// Interleaved A/B of 6 against 5 waves where both compile clean (round 1): L = 6, n_q = 16 +1 %, n_q = 8 +7 %.
#ifndef SHP_WMIN_WAVES
#define SHP_WMIN_WAVES(L) (((L) >= 0 && (L) <= 6 && (L) != 3) ? 5 : 4)
#endif
#ifndef SHP_MIN_WAVES
#define SHP_MIN_WAVES(L, NEEDV) \
  ((NEEDV) ? (((L) == 0 || (L) == 1 || (L) == 6) ? 6 : 5) : (((L) >= 0 && (L) <= 6) ? 6 : 5))
#endif
// kernels that evaluate particle j from per-azimuth polynomials (JPT): the rows of j's table are read from LDS
// (80 registers up to L = 4, 96 up to L = 6 and for the one-wave kernel of L = 9, 128 beyond — L = 5, 8 and the two-wave
// kernel of L = 9 come out a step below their bound; A/B per order: profiles/r03_zzzz_ab_root_loop.txt, r03_zzzzzz_ab_lds_abs.txt)
#ifndef SHP_JMIN_WAVES
#define SHP_JMIN_WAVES(L, NEEDV, WPP) (((L) <= 4) ? 6 : (((L) <= 6 || ((L) == 9 && (WPP) == 1 && !(NEEDV))) ? 5 : 4))
#endif

// docs/SPEC.md §2.6: residual below which the inverse-quadratic extrapolation is accepted
#ifndef SHP_PEEL
#define SHP_PEEL(L) ((L) >= 6)
#endif
#ifndef SHP_TAU3
#define SHP_TAU3 1e-4
#endif

// ---- per-wave dynamic LDS (doubles unless noted) ---------------------------
//   frame[kFrame]        pair frame, FR_* below
//   trig[6 (L+1)]        cos/sin of m alpha, m beta, m gamma
//   v0[(L+1)^2], v1[..]  ping-pong coefficient vectors of the rotation
//   ring[rows][L+1][4]   A_km, B_km, dA/dmu, dB/dmu of `rows` consecutive rings; the two B slots
//                        of m = 0 (identically zero) carry mu_k and sigma_k.  rows = nq when that
//                        leaves the CU enough waves, else the cap is processed in ring groups.
//   qri[n], qrj[n], qp[n] (16-bit)   queue of inside nodes, n = kQueue (body-frame and weighted kernels: a ring buffer)
//                        or queue_capacity() (per-azimuth kernels: a stack)
constexpr int kQueue = 128;  // entries: a slab of 64 nodes adds <= 64 to a queue holding < 64 (the per-azimuth kernels' slabs
                             // are 64 node PAIRS: see queue_capacity)
constexpr int kFrame = 40;
constexpr int kRedStride = 72;    // epilogue reduction: doubles between the 64-entry rows of the seven sums (64 + 8: rows
                                  // four apart share banks, not all seven)
constexpr int kRedDoubles = 7 * kRedStride + 56 + 7 + 6;   // scratch of the epilogue behind the frame
// The per-azimuth kernels keep only the slots they read from LDS — E1 ... WSC (12..29) and RHO, KN, EXPO, IJ (36..39):
// the Euler angles are the rotation kernel's, the pair's scalars arrive as scalar loads — packed to the front: 22
// doubles instead of 40.  LDS is allocated in granules of 1 280 B (profiles/r04_ac_lds_granule.txt); the 128 B put
// L = 7 / n_q = 16 and L = 10 / n_q = 16 a granule lower (18 instead of 16, 14 instead of 12 waves per CU).
constexpr int kFrameJ = 24;
__host__ __device__ constexpr int frj(const int slot) { return slot >= 36 ? slot - 18 : slot - 12; }
constexpr int kRecStride = 40;   // doubles per pair record: the first kRecUsed are copied into the frame
constexpr int kRecUsed = 40;
// per-pair scalars live in the frame too: as VALU results they would sit in VGPR pairs for
// the whole kernel (wave-uniform FP64 values cannot be SGPRs without readfirstlane)
// With P.jpoly the six slots of FR_BJ1 / FR_BJ2 carry cos, sin of the Euler angles of j's frame M_j = [BJ1 BJ2 BJC]
// instead (FR_EULERJ): the compiled orders never form a direction in j's body frame.
enum { FR_EULERJ = 0, FR_JPJ = 6 /* rho^2 - R_j^2 */, FR_JTOL1 = 7 /* 1e-7 R_j */, FR_JTOL3 = 8 /* SHP_TAU3 R_j */,
       FR_JTINY = 9 /* 1e-14 R_j */ };   // ... and the slots of BJC, d_j these (pair_setup.hpp)
enum { FR_BJ1 = 0, FR_BJ2 = 3, FR_BJC = 6, FR_DJ = 9, FR_E1 = 12, FR_E2 = 15, FR_C = 18, FR_D = 21,
       FR_RJ = 24, FR_RJ2 = 25, FR_RHO2 = 26, FR_HW = 27, FR_HM = 28, FR_WSC = 29,
       FR_EULER = 30 /* cos, sin of alpha, beta, gamma */, FR_RHO = 36,
       // the force law's operands, looked up by the set-up kernel (pair_setup.hpp)
       FR_KN = 37, FR_EXPO = 38, FR_IJ = 39 /* i, j as two ints */ };

#ifndef SHP_ALIAS_FROM_L
#define SHP_ALIAS_FROM_L 7
#endif
struct WaveLdsLayout {
  int trig, v0, v1, ring, qri, qrj, qp, bytes;  // offsets in doubles (qp: in doubles too), total bytes
  int qw;                                        // weighted rule only: the queued nodes' weights
  int coef;                                      // end of the queue region (the table of particle j starts here)
  int pj, gh;                                    // particle j's polynomials: first-stage scratch, per-azimuth table
  int tr;                                        // even L: (cos, sin)(psi_l), l < n_q, 2 doubles each, behind the table's rows
  int pi, v0i;                                   // JPT kernels: particle i's first-stage polynomials PJ^i (they stay for every ring
                                                 // group); particle i's rotated vector beside particle j's (both in the rows of
                                                 // the per-azimuth table, which is built after the first stage has read them)
  int glw;                                       // JPT kernels: the Gauss-Legendre weights (nqj doubles)
  int park;                                      // JPT kernels with ring groups: 2 x 64 sums parked around the builds of the later groups (in the empty queue)
  int stash;                                     // JPT kernels: 64 prefetched Gauss nodes for the first pass of the first ring build
  int qstride;                                   // two waves per pair: doubles between the waves' private queue regions
  int qcap;                                      // JPT kernels: entries of the node queue (kQueue ... kQueue + 64, see queue_capacity)
};
constexpr int kLdsGranule = 1280;                // bytes: a workgroup's LDS is allocated in 1/128 of the CU's 160 KB
// Row of the per-azimuth table: G_l (L + 1 coefficients, descending powers), H_l (L), cos(psi_l), sin(psi_l) (the
// higher orders follow by the angle-addition recurrence where r_i is evaluated), the Gauss-Legendre weight of the
// RING with the row's index (n_q rows, n_q rings: the table doubles as the weight table), then padding to 16-byte
// rows whose stride is 2 mod 4 doubles: sixteen lanes reading sixteen rows with ds_read_b128 then spread over all
// banks (a 128-byte stride, 2L + 4 = 16 at L = 6, puts every row on the same banks: the kernel ran 3x slower).
// Round 4: for EVEN L the 2L + 1 coefficients and the weight are 2L + 2 doubles — already 2 mod 4 — and (cos, sin)(psi_l)
// live in an array of their own behind the rows (jpoly_trig_sep; W.tr): 4 doubles per row less than the padded
// 2L + 6.  For odd L the row of 2L + 4 doubles holds all of it, as before.
__host__ __device__ constexpr bool jpoly_trig_sep(const int L) { return (L % 2) == 0; }
__host__ __device__ constexpr int jpoly_row(const int L) { return jpoly_trig_sep(L) ? 2 * L + 2 : 2 * L + 4; }
__host__ __device__ constexpr int jpoly_trig(const int L) { return 2 * L + 2; }   // odd L: offset of cos(psi_l) in a row; sin follows (one 16-byte pair)
__host__ __device__ constexpr int jpoly_glw(const int L) { return 2 * L + 1; }    // offset of the weight of ring `row index` (the odd slot behind the 2L + 1 coefficients)
// Rows of the first-stage table PJ: (order m, part) for m = 0..L+1 — the order L + 1 is empty (zeros), see jpoly_build.
__host__ __device__ constexpr int jpoly_rows(const int L) { return 2 * L + 4; }
// ... of which particle i needs the real orders only: PJ^i, (2L + 2) polynomials of L + 1 coefficients (an even count)
__host__ __device__ constexpr int jpoly_pi_doubles(const int L) { return (2 * L + 2) * (L + 1); }
__host__ __device__ inline WaveLdsLayout wave_lds_layout(const int L, const int rows, const bool weighted = false,
                                                         const int nqj = 0, const int qcap = kQueue)
{
  WaveLdsLayout w;
  const int ns = (L + 1) * (L + 1);
  // frame | [rotation scratch] | rotated coefficients v0 | ring rows | queue.  The scratch of the coefficient
  // rotation (the Euler trig tables and the second work vector v1) is dead before the first node is queued.  From
  // L = 7 on it lies over the queue, which leaves room for more resident ring rows (L = 12, n_q = 32: +3 %); up to
  // L = 6 it keeps its own place: the wave count is limited elsewhere there (A/B: no gain from 24 instead of 21
  // waves per CU) and the separate layout compiles without a spill under the 80-VGPR bound.
  // Compiled orders (nqj > 0): the rotations run in pair_rotate_kernel; frame | v0 | ring rows | queue | per-azimuth
  // polynomials of particle j.
  const bool alias = SHP_ALIAS_FROM_L <= L;
  w.trig = kFrame;
  w.pi = w.v0i = 0;
  // JPT kernels (nqj > 0), round 4: the ring tables are Horner evaluations of particle i's first-stage polynomials
  // PJ^i (cap_frame_rings_poly), (2L + 2)(L + 1) doubles that replace the rotated vector as what has to survive for the
  // ring builds.  With all rings resident (one ring group) they lie over the queue, which is empty while rings are
  // built; with ring groups they keep a place of their own behind the frame.  Both rotated vectors wait for the first
  // stage in the rows of particle j's table.
  const int npi = jpoly_pi_doubles(L);
  const bool one_group = nqj > 0 && rows >= nqj;
  w.v0 = (alias || nqj > 0) ? kFrame : w.trig + 6 * (L + 1);
  w.v1 = w.v0 + ns;
  w.ring = (nqj > 0) ? (one_group ? kFrameJ : kFrameJ + npi) : (alias ? w.v0 + ns : w.v1 + ns);
  w.ring += w.ring & 1;  // 16-byte aligned rows for ds_read_b128
  // the first stage of particle j's polynomials ((2L+4)(L+1) doubles, +2: a read one past a row's end) lies over the ring rows, which are built later
  int ringsz = 4 * rows * (L + 1);
  if (nqj > 0 && rows > 0 && ringsz < jpoly_rows(L) * (L + 1) + 2) ringsz = jpoly_rows(L) * (L + 1) + 2;
  w.pj = w.ring;
  w.qcap = qcap;   // (a multiple of 4: the 16-bit node indices end on an 8-byte boundary)
  w.qri = w.ring + ringsz;
  w.qrj = w.qri + qcap;
  w.qp = w.qrj + qcap;
  w.qw = w.qp + qcap / 4;
  w.park = w.qri;
  w.coef = w.qw + (weighted ? qcap : 0);
  w.stash = w.qri + 128;
  if (nqj > 0) {
    // PJ^i over the queue (one ring group: a single build, before any sum exists — nothing is parked) or behind the
    // frame (ring groups: the later builds park two sums in the empty queue).  The prefetched Gauss nodes of the first
    // pass wait at the end of the ring rows where the first pass (entries 0..63 = doubles 0..255) does not write and
    // particle j's first stage does not reach, else behind the polynomials / the parked sums.
    w.pi = one_group ? w.qri : kFrameJ;
    w.park = w.qri;
    const int pjsz = jpoly_rows(L) * (L + 1) + 2;
    if (ringsz - 64 >= 256 && ringsz - 64 >= pjsz) w.stash = w.ring + ringsz - 64;
    else w.stash = one_group ? w.pi + npi : w.qri + 128;
    int need = one_group ? w.pi + npi : w.park + 128;
    if (w.stash >= w.qri && w.stash + 64 > need) need = w.stash + 64;
    if (need > w.coef) w.coef = need;   // large L: the polynomials are longer than the queue
  }
  if (alias && nqj == 0) {
    w.trig = w.qri;
    w.v1 = w.trig + 6 * (L + 1);
    if (w.v1 + ns > w.coef) w.coef = w.v1 + ns;  // large L: the scratch is longer than the queue
  }
  w.coef += w.coef & 1;
  w.gh = w.coef;   // per-azimuth polynomials of particle j: nqj rows, resident for the whole pair
  w.glw = w.gh + jpoly_glw(L);   // weight of ring k at glw + k * jpoly_row(L)
  int ghsz = nqj * jpoly_row(L);
  w.tr = w.gh + ghsz;   // (even L; 16-byte aligned: rows are an even number of doubles)
  if (nqj > 0) {
    if (jpoly_trig_sep(L)) ghsz += 2 * nqj;
    w.v0 = w.gh;         // particle j's rotated vector, then particle i's behind it: read by the first stage only
    w.v0i = w.gh + ns;
    if (ghsz < 2 * ns) ghsz = 2 * ns;
  }
  w.bytes = 8 * (w.gh + ghsz);
  // the epilogue's reduction scratch lies behind the frame, over everything that is dead by then
  if (w.bytes < 8 * ((nqj > 0 ? kFrameJ : kFrame) + kRedDoubles)) w.bytes = 8 * ((nqj > 0 ? kFrameJ : kFrame) + kRedDoubles);
  w.bytes = (w.bytes + 15) & ~15;
#ifdef SHP_LDS_PAD   // experiment builds only (make variant): what do fewer resident waves cost?
  if (nqj > 0) w.bytes += SHP_LDS_PAD;
#endif
  w.qstride = 0;
  return w;
}
constexpr int kRedPerWave = (7 * kRedStride + 56 + 7 + 6 + 1) & ~1;   // epilogue scratch of one wave, even
__host__ __device__ inline WaveLdsLayout pair_lds_layout2(const int L, const int rows, const int nq, const int qcap)
{
  WaveLdsLayout w;
  const int ns = (L + 1) * (L + 1);
  w.trig = w.v1 = w.qw = w.coef = 0;   // not used by the JPT kernels
  w.pi = kFrameJ;                       // particle i's first-stage polynomials: they stay for the ring groups
  w.ring = w.pi + jpoly_pi_doubles(L);
  w.ring += w.ring & 1;
  int ringsz = 4 * rows * (L + 1);
  if (ringsz < jpoly_rows(L) * (L + 1) + 2) ringsz = jpoly_rows(L) * (L + 1) + 2;
  ringsz += ringsz & 1;
  w.pj = w.ring;
  w.gh = w.ring + ringsz;
  w.glw = w.gh + jpoly_glw(L);
  w.v0 = w.gh;        // both rotated vectors wait for the first stage in the rows of particle j's table
  w.v0i = w.gh + ns;
  int ghsz = nq * jpoly_row(L);
  w.tr = w.gh + ghsz;
  if (jpoly_trig_sep(L)) ghsz += 2 * nq;
  if (ghsz < 2 * ns) ghsz = 2 * ns;
  int shared_end = w.gh + ghsz;
  // the epilogue's scratch (one block per wave) lies over everything behind the frame, the queues included: wave 0's
  // from the frame on, wave 1's at the end of the pair's LDS
  const int qs = 2 * qcap + qcap / 4;
  w.qcap = qcap;
  if (shared_end + 2 * qs < kFrameJ + 2 * kRedPerWave) shared_end = kFrameJ + 2 * kRedPerWave - 2 * qs;
  shared_end += shared_end & 1;
  w.qri = shared_end;
  w.qrj = w.qri + qcap;
  w.qp = w.qrj + qcap;
  w.park = w.qri;                          // 2 x 64 parked sums while the ring rows of a later group are built (the queue is empty then)
  w.stash = w.qri + 128;                   // (not used: two-wave kernels request their Gauss nodes where they use them)
  w.qstride = qs;                          // 288 at 128 entries
  w.bytes = (8 * (shared_end + 2 * w.qstride) + 15) & ~15;
  return w;
}

// 16-byte LDS reads.  Rows of the ring tables and of particle j's table are 16-byte aligned (wave_lds_layout), but the
// compiler only knows that a double* is 8-byte aligned and reads adjacent doubles with ds_read2_b64 — two 8-byte
// accesses per lane, serviced at HALF the rate of ds_read_b128 (128 against 256 B/clk/CU) and banked modulo 32 instead
// of 64 dwords, where the 36-dword row stride of particle j's table (chosen for ds_read_b128) puts rows l and l + 8 on
// the same banks.  Measured on the round-2 kernel (profiles/r03_b_lds_sites.txt): 9 LDS-array cycles per LDS
// instruction in the node loops, 27 % of them bank conflicts, the LDS pipe 83 % busy beside an 80 % busy VALU.
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2d lds2(const double* p) { return *(const v2d*)__builtin_assume_aligned(p, 16); }
// TWO WAVES PER PAIR (template parameter WPP = 2 of pair_contact_kernel; JPT kernels): the workgroup is one pair, the
// tables — frame, particle i's rotated vector, the ring rows, particle j's per-azimuth polynomials — are shared and
// built by all 128 lanes, each wave classifies and integrates HALF of the azimuths (wave h the node pairs l, l + n_q
// with h n_q / 2 <= l < (h + 1) n_q / 2) with a node queue of its own.  For the orders and rules where one wave's
// private copy of the tables leaves a CU too few waves: L = 12, n_q = 32 needs 14.6 KB per one-wave pair (11 waves per
// CU, VALU 66 % busy, profiles/r03_e_L12_pmc.txt), 17.3 KB per two-wave pair (18 waves' worth; the registers allow 16).
//   frame | PJ^i (particle i's first-stage polynomials; stay for the ring groups) | ring rows (first: first stage of j's table) | j's table |
//   [epilogue scratch of both waves over everything behind the frame] | queue of wave 0 | queue of wave 1
__host__ __device__ inline WaveLdsLayout pair_lds_layout2(const int L, const int rows, const int nq, const int qcap = kQueue);
// Entries of the node queue of the per-azimuth kernels.  A slab of node pairs brings up to 128 inside nodes to a queue
// that holds fewer than 64: 128 entries overflow when a dense slab meets a leftover (the slab is then classified a second
// time after a short batch: 0.74 slabs per pair at the headline, 5 % of the kernel's instructions), 191 never do.  The
// LDS of a workgroup is allocated in granules of 1 280 bytes: the queue takes what the layout leaves of its last granule
// (18 bytes per entry; `waves` = queues in the workgroup's LDS), at no cost in resident waves.
__host__ __device__ inline int queue_capacity(const int bytes_at_128, const int waves)
{
  const int slack = (bytes_at_128 + kLdsGranule - 1) / kLdsGranule * kLdsGranule - bytes_at_128;
  int extra = (slack / (18 * waves)) & ~3;
  if (extra > 64) extra = 64;
  return kQueue + extra;
}

// Integer products of the node loops through the 24-bit multiplier (v_mul_u32_u24 / v_mul_i32_i24: full rate;
// v_mul_lo_u32 is a quarter-rate instruction).  Operands are node, ring and azimuth indices (< 2^15) and the
// multiply-shift constants (< 2^24).  Only the JPT kernels take it: in four forces-only body-frame kernels the changed
// instruction mix tips the register allocator into 2-4 spills.
template <bool ON>
__device__ __forceinline__ unsigned umul_sel(const unsigned a, const unsigned b) { return ON ? __umul24(a, b) : a * b; }
template <bool ON>
__device__ __forceinline__ int mul_sel(const int a, const int b) { return ON ? __mul24(a, b) : a * b; }

// Wave votes as SCALAR mask arithmetic.  HIP's __any() goes through a 0 / 1 value per lane (v_cndmask + v_cmp, two
// vector instructions per vote) and boolean algebra on lane predicates is materialised the same way; a ballot is the
// compare's own SGPR pair, masks combine on the scalar unit, and lane_of() hands a mask back as a lane predicate
// (s_and_saveexec on the mask itself).
// (the HIP wrappers __ballot / __any take an int: the predicate is first turned into 0 / 1 per lane and compared again)
__device__ __forceinline__ unsigned long long wave_ballot(const bool p) { return __builtin_amdgcn_ballot_w64(p); }
// wave_any: the mask passes through an (empty) scalar asm operand — compared directly, LLVM turns `ballot != 0` back into
// the 0 / 1-per-lane idiom (v_cndmask + v_cmp + branch on vccz); through the operand it is s_cmp_lg_u64 + a scalar branch
// (SCALAR = false, the plain comparison: the body-frame kernels, which have no scalar register to spare for it)
template <bool SCALAR = true>
__device__ __forceinline__ bool mask_any(unsigned long long m)
{
  if constexpr (SCALAR) asm("" : "+s"(m));
  return m != 0ULL;
}
template <bool SCALAR = true>
__device__ __forceinline__ bool wave_any(const bool p) { return mask_any<SCALAR>(__builtin_amdgcn_ballot_w64(p)); }
__device__ __forceinline__ bool lane_of(const unsigned long long mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }

__device__ __forceinline__ unsigned launder_s32(unsigned v)   // ... of a wave-uniform value: it stays in a scalar register
{
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ unsigned launder_u32(unsigned v)
{
  asm volatile("" : "+v"(v));
  return v;
}

// The lane index, made where it is asked for.  Anything derived from threadIdx is invariant everywhere: addresses the
// epilogue computes from it (lane * 8 as a 64-bit offset, ...) are merged with the prologue's and then carried —
// or spilled — through the node loops, where registers are scarcest.  Two instructions.
__device__ __forceinline__ int fresh_lane()
{
  int v;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(v));
  return v;
}

// The kernel's arguments, read where they are used.  A by-value argument struct is loaded from the kernarg segment in
// the entry block; the epilogue's sixteen pointers and flags (f, torque, pair_i, pair_j, type, kn, ...) would then sit
// in ~30 scalar registers through the node loops, where the coefficient windows of sh_eval need them: the allocator
// parks them in lanes of a vector register and restores eight of them in EVERY iteration of the root loop and of the
// slab loop (v_readlane, ~330 vector instructions per pair at L = 6).  Reading them through a kernarg pointer the
// compiler cannot see through makes them plain scalar loads at the point of use.
typedef const PairParams __attribute__((address_space(4))) LateParams;
__device__ __forceinline__ LateParams* late_params()
{
  unsigned long long a = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(a));
  return (LateParams*)a;
}

__device__ __forceinline__ void wave_lds_sync()
{
  // LDS written by some lanes of the wave, read by others: DS operations of one
  // wave execute in order, so only the compiler has to be kept from reordering.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ... and between the waves of a pair (WPP = 2): a workgroup barrier
template <int WPP>
__device__ __forceinline__ void pair_sync()
{
  if constexpr (WPP == 1) wave_lds_sync();
  else __syncthreads();
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

// The same with ONE Newton step: from v_rsq_f64's 2^-26 the step leaves 3/2 (2^-26)^2 = 3.3e-16 plus its own
// rounding, i.e. 2-3 ulp.  Used where the root only normalises a direction or feeds a residual that is compared
// with tolerances of 1e-7 and more (node loops: 4 VALU instructions fewer per radius evaluation).
__device__ __forceinline__ double rsqrt_nr1(const double x)
{
  double y = __builtin_amdgcn_rsq(x);
  const double h = fma(-x * y, y, 1.0);
  y = fma(y * 0.5, h, y);
  return y;
}

// sqrt(x) for x >= 0 as x * rsqrt(x): a third of the VALU work of the IEEE sqrt() expansion
// (which rescales, iterates and fixes up special cases), accurate to the last ulp or two.
__device__ __forceinline__ double sqrt_nr(const double x) { return (x > 0.0) ? x * rsqrt_nr(x) : 0.0; }

// sqrt(x) with the one-step root: 2-3 ulp.  For the brackets, first iterate and end-point residual of the inner-radius
// search and wherever else 1e-15 relative is far inside what the value is used for.
// (x <= 0 and NaN give 1e-150 for 0: one v_max_f64 — written as such, fmax() adds a canonicalising v_max_f64 under IEEE
// mode — instead of a compare and two selects)
__device__ __forceinline__ double max_raw(const double x, const double c)
{
  double r;
  asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(x), "v"(c));
  return r;
}
template <bool CLAMP = false>
__device__ __forceinline__ double sqrt_nr1(const double x)
{
  if constexpr (CLAMP) {
    const double t = max_raw(x, 1e-300);
    return t * rsqrt_nr1(t);
  } else {
    return (x > 0.0) ? x * rsqrt_nr1(x) : 0.0;   // (the body-frame kernels have no register for the constant)
  }
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

// 1/d with ONE Newton step: v_rcp_f64's 2^-26 squared is 2^-52, plus the step's own rounding: 2-3 ulp.
__device__ __forceinline__ double rcp_nr1(const double d)
{
  const double r = __builtin_amdgcn_rcp(d);
  return fma(fma(-d, r, 1.0), r, r);
}

// V^e for the exponents the force law usually asks for (m - 1 or m a multiple of 1/4) by square
// roots: ~25 VALU instructions instead of the ~150 of pow(); the whole wave issues them for lane 0.
__device__ __forceinline__ double pow_quarter(const double v, const double e)
{
  if (e == 0.25) return sqrt_nr(sqrt_nr(v));
  if (e == 0.5) return sqrt_nr(v);
  if (e == 0.75) { const double s = sqrt_nr(v); return s * sqrt_nr(s); }
  if (e == 1.0) return v;
  if (e == 1.25) return v * sqrt_nr(sqrt_nr(v));
  if (e == 1.5) return v * sqrt_nr(v);
  if (e == 2.0) return v * v;
  return pow(v, e);
}

// ---- set-up: particle i's expansion in the cap frame ------------------------
// M = [b1 b2 bc]: the cap axes (e1, e2, c) in i's body frame, so that
// r_cap(u') = r_body(M u').  M = Rz(alpha) Ry(beta) Rz(gamma), Ry(beta) =
// Rx(-90) Rz(beta) Rx(90); with (O_A f)(u) = f(A u), O_{AB} = O_B O_A, hence
//   c' = Z(gamma) X Z(beta) X^T Z(alpha) c ,  X = T(Rx(+90)) (constant).
// alpha is read off the third column of M; gamma follows from the WELL
// CONDITIONED sum (cos beta >= 0) or difference (cos beta < 0) of the two
// angles, so that the 1/sin(beta) error of alpha near the poles only moves the
// axis of a vanishing tilt.
template <int L>
__device__ __forceinline__ void cap_frame_rotate(const PairParams& P, double* __restrict__ lw, const WaveLdsLayout& W,
                                                 const int LL, const int si, const int lane, const int fr_euler = FR_EULER)
{
  const int ns = (LL + 1) * (LL + 1);
  double* trig = lw + W.trig;
  double* v0 = lw + W.v0;
  double* v1 = lw + W.v1;
  // Euler angles (computed per pair by pair_setup_kernel): lanes 0,1,2 tabulate cos/sin(m angle) for alpha, beta, gamma
  if (lane < 3) {
    const double c1 = lw[fr_euler + 2 * lane], s1 = lw[fr_euler + 2 * lane + 1];
    double* t = trig + 2 * (LL + 1) * lane;
    double cm = 1.0, sm = 0.0;
    for (int m = 0; m <= LL; ++m) {
      t[2 * m] = cm;
      t[2 * m + 1] = sm;
      const double c = fma(cm, c1, -(sm * s1)), s = fma(cm, s1, sm * c1);
      cm = c;
      sm = s;
    }
  }
  wave_lds_sync();
  const double* creal = P.creal + (size_t)si * ns;
  const int XW = LL / 2 + 1;
  // five steps; step s reads `src`, writes `dst`; element e = row (l, m) of the vector
  for (int step = 0; step < 5; ++step) {
    const double* src = (step == 0) ? creal : ((step & 1) ? v0 : v1);
    double* dst = (step & 1) ? v1 : v0;
    for (int e = lane; e < ns; e += 64) {
      double out;
      if ((step & 1) == 0) {  // Z(angle): 0 alpha, 2 beta, 4 gamma
        const int inf = P.xinfo[e];
        const int l = inf & 255, mm = (inf >> 8) - l;
        const double* t = trig + 2 * (LL + 1) * (step >> 1);
        const int m = mm < 0 ? -mm : mm;
        const double self = src[e], other = src[l * l + l - mm];
        const double cm = t[2 * m], sm = t[2 * m + 1];
        out = (mm == 0) ? self : fma(cm, self, (mm > 0 ? sm : -sm) * other);
        if (step == 4) out *= P.gscale[e];
      } else {  // X^T (step 1) or X (step 3), ELL rows
        const size_t rowoff = ((size_t)(step == 1 ? ns : 0) + e) * XW;
        const double* val = P.xval + rowoff;
        const int* col = P.xcol + rowoff;
        out = 0.0;
#pragma unroll
        for (int t = 0; t < ((L >= 0) ? L / 2 + 1 : XW); ++t) out = fma(val[t], src[col[t]], out);
      }
      dst[e] = out;
    }
    wave_lds_sync();
  }
  // the rotated, scaled coefficients are now in v0
}

// The rotations as a kernel of their own (compiled orders), ONE LANE PER ROTATION.  Inside the contact kernel a
// rotation is a chain of five dependent table-load / LDS steps, ~10 000 cycles of latency for 85 instructions during
// which the wave holds its registers and LDS (a wave-per-rotation kernel measured 0.6 ms per launch at the headline
// however many waves were resident: round 2, profiles/r02_w_*).  Here a wave carries 64 rotations through the same five steps; the
// rotation is block diagonal in l, so a lane's block of 2l + 1 values lives in LDS as [element][lane] (conflict
// free) between the steps that gather (X^T, X) and in registers for those that do not (the Z turns; cos/sin(m angle) of
// the three angles sit in registers too), the X matrices are wave-uniform (scalar loads, SGPR operands) and every
// loop is wave-uniform: ~20 instructions per rotation.  The arithmetic and its order are those of cap_frame_rotate.
// Layout of the rotated vectors: TILES of 64 rotations (one wave of the rotation kernel), inside a tile block-major —
// [l-block][rotation][element of the block] — so that the 64 x (2l+1) doubles a wave produces for one l are contiguous
// and leave as full 512-byte wave stores.  (Rotation-major rows were written in 49 store instructions of scattered
// 8...104-byte runs per wave: WRITE_SIZE 1.3-1.5x the payload and a kernel bound by its own write pattern.)
// Element e = l^2 + r of rotation T sits at rot_index(L, T, l, r).  Measured (profiles/r03_u_ab_rottile.txt): rotation
// kernel 0.296 -> 0.220 ms at L = 6; the contact kernel's reads become 2L + 1 pieces per vector, which costs it more
// than the rotation kernel gains from L = 9 on (L = 12: +1.2 % per step) — there the rows stay rotation-major, padded
// to whole 64-byte lines.
__host__ __device__ constexpr bool rot_tiled(const int L) { return L <= 8; }
__host__ __device__ constexpr size_t rot_row_doubles(const int L) { return (size_t)(((L + 1) * (L + 1) + 7) & ~7); }
__host__ __device__ constexpr size_t rot_tile_doubles(const int L) { return (size_t)64 * (rot_tiled(L) ? (size_t)(L + 1) * (L + 1) : rot_row_doubles(L)); }
__host__ __device__ inline size_t rot_index(const int L, const int T, const int l, const int r)
{
  if (!rot_tiled(L)) return (size_t)T * rot_row_doubles(L) + (size_t)l * l + r;
  return (size_t)(T >> 6) * rot_tile_doubles(L) + (size_t)64 * l * l + (size_t)(T & 63) * (2 * l + 1) + r;
}
__host__ __device__ inline size_t rot_buffer_doubles(const int L, const size_t nrot) { return ((nrot + 63) / 64) * rot_tile_doubles(L); }
template <int L>
struct RotLaneLds {
  static constexpr int NB = 2 * L + 1;
  static constexpr int a() { return 0; }
  static constexpr int b() { return 0; }   // the second gather reads the block in place: a lane only ever touches its own column
  // L >= 9 (rows of the rotated vectors rotation-major in memory): the block's rows of X^T and of X wait in LDS behind
  // the column block, 2 (2l + 1)(L / 2 + 1) doubles
  static constexpr int xs() { return NB * 64; }
  static constexpr int bytes() { return 8 * (NB * 64 + (rot_tiled(L) ? 0 : 2 * NB * (L / 2 + 1))); }
};
// cos / sin(m angle), m = 1..L, of a lane's three Euler angles: registers (every index is a compile-time constant)
template <int L>
struct RotTrig {
  double c[3 * (L > 0 ? L : 1)], s[3 * (L > 0 ? L : 1)];
};
typedef const int __attribute__((address_space(4))) * ciptr;
__device__ __forceinline__ ciptr launder_uniform_i(const int* p)
{
  asm volatile("" : "+s"(p));
  return (ciptr)p;
}
template <int L, int LB>
__device__ __forceinline__ void rotate_lane_block(const PairParams& P, double* __restrict__ sm, const int lane,
                                                  const double* __restrict__ cre, double* __restrict__ rot,
                                                  const int task0, const int ntasks, const RotTrig<L>& T)
{
  constexpr int ns = (L + 1) * (L + 1), n = 2 * LB + 1, base = LB * LB, XW = L / 2 + 1, XN = LB / 2 + 1;
  // The X matrices, their column indices and the ring scale are the same for every lane: through constant-address-space
  // pointers they are SCALAR loads into SGPRs (an SGPR can be the multiplier of a v_fma_f64).  Through the plain global
  // pointers of the argument struct the compiler emits ~220 per-lane vector loads of them per wave (the kernel also
  // stores to global memory, so it may not assume the tables unchanged).
  // Measured on the tiled layout: L = 6 rotation kernel 0.220 -> 0.149 ms; at L = 12 (244 VGPRs, 1 100 scalar loads per
  // wave) the step gets 3.9 % slower, so from L = 9 on the plain pointers stay (profiles/r03_z_ab_rot_scalar.txt).
  const auto xval = [&] { if constexpr (rot_tiled(L)) return launder_uniform(P.xval); else return P.xval; }();
  const auto gsc = [&] { if constexpr (rot_tiled(L)) return launder_uniform(P.gscale); else return P.gscale; }();
  double* A = sm + RotLaneLds<L>::a() + lane;
  double* B = sm + RotLaneLds<L>::b() + lane;
  // L >= 9 (round 4): this block's rows of X^T and X are staged in LDS by the wave (contiguous in the ELL table: row =
  // base + r) and read as broadcasts at immediate offsets, with the compile-time columns of the small orders — instead
  // of two vector loads (value, column) and six integer instructions of address arithmetic per v_fma_f64 (4 353 of the
  // L = 12 kernel's ~8 000 vector instructions per wave were 32-bit integer, 1 067 were vector memory reads).  With
  // constant addresses the compiler forwards a lane's LDS stores to its own loads, so the block lives in registers
  // (230-254 of them: two waves per SIMD as before).  L = 12: 0.926 -> 0.752 ms per launch, L = 9: 0.488 -> 0.457
  // (profiles/r04_x_rot_kernel_times.txt).  Tried on top and dropped: the trig multiples by recurrence instead of the
  // 6 L-double table (0.83 ms at two waves per SIMD; capped at three waves the kernel spills and runs 0.95 ms).
  constexpr bool XLDS = !rot_tiled(L);
  const double* xs = sm + RotLaneLds<L>::xs();
  if constexpr (XLDS) {
    double* xw = sm + RotLaneLds<L>::xs();
    for (int i = lane; i < n * XW; i += 64) {
      xw[i] = P.xval[((size_t)ns + base) * XW + i];
      xw[n * XW + i] = P.xval[(size_t)base * XW + i];
    }
    wave_lds_sync();
  }
  // Z(alpha) on the way in: the pair (l, +m), (l, -m) turns by m alpha
  A[64 * LB] = cre[base + LB];
#pragma unroll
  for (int m = 1; m <= LB; ++m) {
    const double c = T.c[0 * L + m - 1], s = T.s[0 * L + m - 1];
    const double p = cre[base + LB + m], q = cre[base + LB - m];
    A[64 * (LB + m)] = fma(c, p, s * q);
    A[64 * (LB - m)] = fma(c, q, -(s * p));
  }
  // X^T: rows ns + e of the ELL table
  double xb[n];
#pragma unroll
  for (int r = 0; r < n; ++r) {
    const size_t ro = ((size_t)ns + base + r) * XW;
    double o = 0.0;
    // the columns of row (LB, r - LB) are known at compile time (sh_const::xpat_*): immediate LDS offsets, no index loads
    // (constants once the loops are unrolled)
    const int first = sh_const::xpat_first(LB, r - LB), count = sh_const::xpat_count(LB, r - LB);
#pragma unroll
    for (int t = 0; t < XN; ++t) {
      if constexpr (rot_tiled(L)) {
        if (t < count) o = fma(xval[ro + t], A[64 * (LB + first + 2 * t)], o);
      } else {
        if (t < count) o = fma(xs[r * XW + t], A[64 * (LB + first + 2 * t)], o);   // L >= 9: X from LDS
      }
    }
    xb[r] = o;
  }
  // Z(beta), in registers
  B[64 * LB] = xb[LB];
#pragma unroll
  for (int m = 1; m <= LB; ++m) {
    const double c = T.c[1 * L + m - 1], s = T.s[1 * L + m - 1];
    const double p = xb[LB + m], q = xb[LB - m];
    B[64 * (LB + m)] = fma(c, p, s * q);
    B[64 * (LB - m)] = fma(c, q, -(s * p));
  }
  // X
#pragma unroll
  for (int r = 0; r < n; ++r) {
    const size_t ro = ((size_t)base + r) * XW;
    double o = 0.0;
    const int first = sh_const::xpat_first(LB, r - LB), count = sh_const::xpat_count(LB, r - LB);
#pragma unroll
    for (int t = 0; t < XN; ++t) {
      if constexpr (rot_tiled(L)) {
        if (t < count) o = fma(xval[ro + t], B[64 * (LB + first + 2 * t)], o);
      } else {
        if (t < count) o = fma(xs[(n + r) * XW + t], B[64 * (LB + first + 2 * t)], o);   // L >= 9: X from LDS
      }
    }
    xb[r] = o;
  }
  if constexpr (rot_tiled(L)) {
    // Z(gamma) and the ring scale, in registers; then the block leaves through LDS in ROTATION-major order (lane's row of
    // n numbers at lane n: odd stride, the plain two passes of a 64-bit write), so that consecutive lanes read — and
    // store to global memory — consecutive elements with no index arithmetic at all: element idx = lane + 64 it of the
    // tile's block is LDS cell idx.  (Read back from the column layout it was a division, a multiply and an exec-masked
    // branch per store: 490 of the kernel's 1 755 vector instructions at L = 6.)
    double* A2 = sm + RotLaneLds<L>::a() + lane * n;
    A2[LB] = xb[LB] * gsc[base + LB];
  #pragma unroll
    for (int m = 1; m <= LB; ++m) {
      const double c = T.c[2 * L + m - 1], s = T.s[2 * L + m - 1];
      const double p = xb[LB + m], q = xb[LB - m];
      A2[LB + m] = fma(c, p, s * q) * gsc[base + LB + m];
      A2[LB - m] = fma(c, q, -(s * p)) * gsc[base + LB - m];
    }
    wave_lds_sync();
    const double* At = sm + RotLaneLds<L>::a() + lane;
    double* out = rot + (size_t)(task0 >> 6) * rot_tile_doubles(L) + 64 * base + lane;
    if (task0 + 64 <= ntasks) {   // a full tile (every workgroup but the last): wave-uniform
  #pragma unroll
      for (int it = 0; it < n; ++it) out[64 * it] = At[64 * it];
    } else {
  #pragma unroll
      for (int it = 0; it < n; ++it) {
        const int idx = lane + 64 * it;   // < 64 n
        // task0 is a multiple of 64 (one tile per workgroup): cell idx is rot_index(L, task0 + idx / n, LB, idx % n)
        if (task0 + idx / n < ntasks) out[64 * it] = At[64 * it];
      }
    }
    wave_lds_sync();
  } else {
    // L >= 9 (rotation-major rows in memory, 244 vector registers: two waves per SIMD): the block leaves transposed
    // through LDS from the column layout; compile-time columns and the lane-major block push these kernels past 256
    // registers — one wave per SIMD, L = 12 / n_q = 32 2 % slower (profiles/r03_zzzzz_ab_rot.txt)
    A[64 * LB] = xb[LB] * gsc[base + LB];
  #pragma unroll
    for (int m = 1; m <= LB; ++m) {
      const double c = T.c[2 * L + m - 1], s = T.s[2 * L + m - 1];
      const double p = xb[LB + m], q = xb[LB - m];
      A[64 * (LB + m)] = fma(c, p, s * q) * gsc[base + LB + m];
      A[64 * (LB - m)] = fma(c, q, -(s * p)) * gsc[base + LB - m];
    }
    wave_lds_sync();
    const double* At = sm + RotLaneLds<L>::a();
  #pragma unroll
    for (int it = 0; it < n; ++it) {
      const int idx = lane + 64 * it;   // < 64 n
      const int tk = idx / n, r = idx - tk * n;
      if (task0 + tk < ntasks) rot[(size_t)(task0 + tk) * rot_row_doubles(L) + base + r] = At[64 * r + tk];
    }
    wave_lds_sync();
  }
  if constexpr (LB < L) rotate_lane_block<L, LB + 1>(P, sm, lane, cre, rot, task0, ntasks, T);
}
template <int L>
__global__ void __launch_bounds__(64) pair_rotate_lane_kernel(const PairParams P, double* __restrict__ rot)
{
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_rl[];
  double* sm = (double*)smem_rl;
  const int lane = threadIdx.x;
  const int task0 = 2 * P.slot0 + blockIdx.x * 64, ntasks = 2 * P.npairs;   // slot0 is a multiple of 32: whole tiles
  const int task = task0 + lane;
  const int w = (task < ntasks ? task : ntasks - 1) >> 1, which = task & 1;
  const int* rid = P.rec_i + 4 * (size_t)w;
  const bool live = task < ntasks && rid[0] != 0;
  const int shape = live ? rid[1 + which] : 0;   // dead slots rotate shape 0 by the identity: nobody reads the result
  const double* eu = P.rec + (size_t)kRecStride * w + (which ? FR_EULERJ : FR_EULER);
  RotTrig<L> T;
  if constexpr (L >= 1) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const double c1 = live ? eu[2 * a] : 1.0, s1 = live ? eu[2 * a + 1] : 0.0;
      double cm = c1, sn = s1;
#pragma unroll
      for (int m = 1; m <= L; ++m) {
        T.c[a * L + m - 1] = cm;
        T.s[a * L + m - 1] = sn;
        const double c = fma(cm, c1, -(sn * s1)), s = fma(cm, s1, sn * c1);
        cm = c;
        sn = s;
      }
    }
  }
  rotate_lane_block<L, 0>(P, sm, lane, P.creal + (size_t)shape * ((L + 1) * (L + 1)), rot, task0, ntasks, T);
}

// Ring tables of rings k0 .. k0 + nrows - 1 from the rotated coefficients.
//
// Lanes are (ring, order class): G = 8, 4, 2 or 1 lanes per ring — as many as 64 lanes give the group's rows — and
// lane (kr, g) builds the orders m = g, g + G, g + 2G, ...  For one m the Legendre recurrence runs over n = m+1 .. L;
// all lanes step through n together (compile-time n for the compiled orders: every LDS and table offset is an
// immediate), a lane joins at n = m + 1 under the exec mask, and steps no lane of the pass needs (n <= the pass's
// smallest m) are skipped wave-uniformly.  Q_n lives in one of two registers by the parity of n, so a step updates
// the older value in place: 8 FP64 operations per step and no moves.  L = 6, n_q = 16: 8 steps in one pass, ~130
// vector instructions per ring group.  (Round 2 up to here: one (k, m) per lane and, for every lane, L steps each
// split by the divergent test t < m: ~200 instructions per 64 entries, 2 passes = ~400 per pair at the headline.)
// PRE (the JPT kernels, which have the registers): the recurrence constants of ALL steps of a
// pass are requested before the first step instead of inside each step's divergent branch — a pass then waits for
// one table load, not for one per step (six dependent ~1000-cycle round trips at L = 6).
// WPP = 2: `lane` is the thread index within the pair's two waves (0..127); with 128 lanes a group of <= 8 rings gets
// 16 lanes per ring — at L <= 15 one order per lane, a single pass.
// DENSE map (one wave per pair, (L + 1) x rows <= 64): lane = (ring, order), L + 1 lanes per ring — every (ring, order)
// of the group in ONE pass.  With power-of-two classes L = 4, n_q = 10 took two passes (orders 0-3, then order 4 alone
// with the whole pass overhead): 213 of that kernel's 1 150 instructions per pair.
// (JPT kernels only: in one forces-only body-frame kernel the extra map tips the register allocator into a spill.)
template <int L, int WPP, bool DENSE>
__device__ __forceinline__ void ring_lane_map(const int lane, const int nrows, int& krl, int& g, int& G, int& rpc, int& lg)
{
  const int lg1 = (nrows <= 8) ? 3 : (nrows <= 16) ? 2 : (nrows <= 32) ? 1 : 0;
  lg = lg1 + (WPP == 2 ? 1 : 0);   // log2 G: as many lanes per ring as the NT lanes give the group's rows; uniform
  G = 1 << lg;
  krl = lane >> lg;
  g = lane & (G - 1);
  rpc = (64 * WPP) >> lg;
  if constexpr (DENSE && L >= 1 && WPP == 1) {
    if ((L + 1) * nrows <= 64) {   // wave-uniform
      G = L + 1;
      krl = lane / (L + 1);
      g = lane - krl * (L + 1);
      rpc = 64 / (L + 1);
      lg = 0;
    }
  }
}

template <int L, bool PRE = false, int WPP = 1, bool DENSE = false>
__device__ __forceinline__ void cap_frame_rings(const PairParams& P, double* __restrict__ lw, const WaveLdsLayout& W,
                                                const int LL, const int lane, const int k0, const int nrows,
                                                const double hw, const double hm, const bool have_first = false)
{
  // have_first (PRE kernels): the Gauss-Legendre node of this lane's ring in the first pass of the first ring group
  // was requested at the start of the kernel and waits in the (empty) queue at lw[W.stash + lane]
  const double* ch = lw + W.v0;
  double* ring = lw + W.ring;
  int krl, g, G, rpc, lg;
  ring_lane_map<L, WPP, DENSE>(lane, nrows, krl, g, G, rpc, lg);
  for (int kr0 = 0; kr0 < nrows; kr0 += rpc) {
    const int kr = kr0 + krl;
    const bool row_ok = kr < nrows && krl < rpc;
    const double tk = (PRE && have_first && k0 == 0 && kr0 == 0) ? lw[W.stash + lane] : P.glt[k0 + (row_ok ? kr : 0)];
    const double mu = fma(hw, tk, hm);
    const double sig2 = fmax(0.0, fma(-mu, mu, 1.0));
    const double sig = sqrt_nr(sig2);
    double sp = 1.0, sigG = sig;   // sigma^g and sigma^G
    for (int t = 0; t < G - 1; ++t) {
      if (t < g) sp *= sig;
    }
    for (int t = 0; t < lg; ++t) sigG *= sigG;
    for (int m0 = 0; m0 <= LL; m0 += G) {
      const int m = m0 + g;
      const bool ok = row_ok && m <= LL;
      const int mc = ok ? m : 0;   // idle lanes read in bounds
      const double* rcm = P.rc + (sh_moff(LL, mc) - mc);   // a'_nm at rcm[n]
      const double* cp = ch + mc;                          // C_nm at cp[n^2 + n], C_n,-m at cm[n^2 + n]
      const double* cm = ch - mc;
      // Q_n and dQ_n/dmu in q[n & 1], d[n & 1]; start: Q_m = 1, Q_(m-1) = 0
      const bool modd = (mc & 1) != 0;
      double qe = modd ? 0.0 : 1.0, qo = modd ? 1.0 : 0.0, de = 0.0, dd = 0.0;
      // m = 0: the B sums read C_n0 again and are not stored (no select in the loop)
      double wa = cp[mc * mc + mc], wb = cm[mc * mc + mc], wad = 0.0, wbd = 0.0;
      constexpr int NPRE = (PRE && L >= 1) ? L : 1;
      double pa[NPRE];
      if constexpr (PRE && L >= 1) {
#pragma unroll
        for (int n = 1; n <= L; ++n) {
          if (n <= m0) continue;
          pa[n - 1] = rcm[n];   // every lane, whatever its m: the address is inside the table, the value unused
        }
#pragma unroll
        for (int n = 1; n <= L; ++n) {
          if (n <= m0) continue;
          asm volatile("" : "+v"(pa[n - 1]));   // keep the requests up here
        }
      }
#pragma unroll
      for (int n = 1; n <= ((L >= 0) ? L : LL); ++n) {
        if (n <= m0) continue;   // wave-uniform: no lane of this pass has m < n
        if (ok && n > m) {
          const double a = (PRE && L >= 1) ? pa[(PRE && L >= 1) ? n - 1 : 0] : rcm[n];
          const double ca = cp[n * n + n], cbm = cm[n * n + n];
          if (n & 1) {
            dd = fma(a, fma(mu, de, qe), -dd);
            qo = fma(a, mu * qe, -qo);
            wa = fma(ca, qo, wa); wb = fma(cbm, qo, wb); wad = fma(ca, dd, wad); wbd = fma(cbm, dd, wbd);
          } else {
            de = fma(a, fma(mu, dd, qo), -de);
            qe = fma(a, mu * qo, -qe);
            wa = fma(ca, qe, wa); wb = fma(cbm, qe, wb); wad = fma(ca, de, wad); wbd = fma(cbm, de, wbd);
          }
        }
      }
      if (ok) {
        // d/dmu [sigma^m W] = sigma^m (W' - m mu W / sigma^2)
        const double f = (m > 0) ? (double)m * mu * rcp_nr(sig2) : 0.0;
        double* o = ring + 4 * (kr * (LL + 1) + m);
        o[0] = sp * wa;
        o[2] = sp * fma(-f, wa, wad);
        if (m == 0) {
          o[1] = mu;   // B_k0 = 0: the slot carries mu_k
          o[3] = sig;  // dB_k0/dmu = 0: carries sigma_k
        } else {
          o[1] = sp * wb;
          o[3] = sp * fma(-f, wb, wbd);
        }
      }
      sp *= sigG;
    }
  }
  pair_sync<WPP>();
}

// Ring tables of the JPT kernels (round 4): HORNER EVALUATIONS of particle i's first-stage polynomials.
//
// jpoly_build leaves, for every order m and part (cos, sin), the polynomial PJ^i[2m + part](mu) with
//   r_i(mu, psi) = sum_m s_m [cos(m psi) PJ^i[2m](mu) + sin(m psi) PJ^i[2m + 1](mu)],   s_m = 1 (m even), sigma (m odd)
// (the host table folds (1 - mu^2)^floor(m / 2) into the polynomial: degree L for even m, L - 1 for odd m).  So
//   A_km = s_m PJ^i[2m](mu_k),   dA_km/dmu = s_m PJ^i[2m]'(mu_k)  [- (mu_k / sigma_k) PJ^i[2m](mu_k) for odd m],   B likewise:
// one lane per table entry (ring, order), value and derivative of both parts by Horner — 4L - 2 v_fma_f64 and L + 1
// ds_read_b128 (the two parts' coefficients are adjacent: 2 (L + 1) doubles) — no recurrence constants, no sigma^m, no
// division.  The associated-Legendre recurrence this replaces (cap_frame_rings: 8 FP64 operations per step, L - m steps
// per entry, a pass per order class) was 256 of the headline kernel's 1 826 vector instructions per pair, 59 % of them
// not FP64 (profiles/r04_d_headline_valu_sites.txt).
// Entry e = ring * (L + 1) + order IS the ring table's own index: the store needs no address arithmetic beyond 32 e.
// Lane map of cap_frame_rings_poly for a group of `nrows` rings on NT lanes: -1 = DENSE, one lane per table entry (ring,
// order), ceil(nrows (L + 1) / NT) passes; lg >= 1 = GROUPED, 2^lg lanes per ring, lane g of a ring takes the orders
// g, g + 2^lg, ... — one pass, and what an entry shares with the other orders of its ring (the Gauss node, mu, sigma,
// 1 / sigma: ~25 of a dense entry's ~69 vector instructions) is made once per lane.  Chosen by that instruction count.
__host__ __device__ inline int ring_poly_map(const int nrows, const int K, const int NT)
{
  const int nent = nrows * K;
  if (nent <= NT) return -1;
  int lg = 0;
  while ((NT >> (lg + 1)) >= nrows && (1 << lg) < K) ++lg;   // as many lanes per ring as one pass over the group allows
  if (lg < 1) return -1;
  const int E = (K + (1 << lg) - 1) >> lg, passes = (nent + NT - 1) / NT;
  return (25 + 44 * E < 69 * passes) ? lg : -1;
}

template <int L, int WPP, class PP = PairParams>
__device__ __forceinline__ void cap_frame_rings_poly(const PP& P, double* __restrict__ lw, const WaveLdsLayout& W,
                                                     const int lane, const int tid, const int k0, const int nrows,
                                                     const double hw, const double hm, const bool have_first)
{
  constexpr int K = L + 1, NT = 64 * WPP;
  const double* pi = lw + W.pi;
  double* ring = lw + W.ring;
  const int nent = nrows * K;
  const int lg = ring_poly_map(nrows, K, NT);   // wave-uniform
  // one table entry: value and mu-derivative of both parts of order m at (mu, sigma), stored at entry e
  auto entry = [&](const int m, const int e, const bool store, const double mu, const double sig, const double isig)
                   __attribute__((always_inline)) {
    const double* row = pi + (2 * K) * m;   // 16-byte aligned: 2K doubles per order, an aligned base
    v2d c[K];
#pragma unroll
    for (int t = 0; t < K; ++t) c[t] = lds2(row + 2 * t);
    // element j of the 2K doubles: cos-part coefficient of mu^p at j = p, sin-part at j = K + p
#define SHP_EL(j) c[(j) >> 1][(j) & 1]
    double pc = SHP_EL(L), ps = SHP_EL(K + L), dc = 0.0, ds = 0.0;
    if constexpr (L >= 1) {
      dc = pc;
      ds = ps;
      pc = fma(pc, mu, SHP_EL(L - 1));
      ps = fma(ps, mu, SHP_EL(K + L - 1));
#pragma unroll
      for (int q = L - 2; q >= 0; --q) {
        dc = fma(dc, mu, pc);
        ds = fma(ds, mu, ps);
        pc = fma(pc, mu, SHP_EL(q));
        ps = fma(ps, mu, SHP_EL(K + q));
      }
    }
#undef SHP_EL
    const bool odd = (m & 1) != 0;
    const double sm = odd ? sig : 1.0;            // s_m
    const double tm = odd ? -mu * isig : 0.0;     // d s_m / d mu
    const double A = sm * pc, dA = fma(tm, pc, sm * dc);
    double B = sm * ps, dB = fma(tm, ps, sm * ds);
    if (m == 0) {   // B_k0 = 0: the slots carry mu_k and sigma_k
      B = mu;
      dB = sig;
    }
    if (store) {
      double* o = ring + 4 * e;
      *(v2d*)__builtin_assume_aligned(o, 16) = v2d{A, B};
      *(v2d*)__builtin_assume_aligned(o + 2, 16) = v2d{dA, dB};
    }
  };
  if (lg >= 1) {
    // GROUPED: lane = (ring, g)
    const int G = 1 << lg, kr = tid >> lg, g = tid & (G - 1);
    const int krc = min(kr, nrows - 1);
    const double tk = (have_first && k0 == 0) ? lw[W.stash + lane] : P.glt[k0 + krc];
    const double mu = fma(hw, tk, hm);
    const double sig2 = max_raw(fma(-mu, mu, 1.0), 1e-300);
    const double isig = rsqrt_nr(sig2);
    const double sig = sig2 * isig;
    for (int m0 = 0; m0 < K; m0 += G) {   // wave-uniform trip count
      const int m = m0 + g;
      entry(min(m, K - 1), krc * K + m, kr < nrows && m < K, mu, sig, isig);
    }
  } else {
    for (int e0 = 0; e0 < nent; e0 += NT) {   // DENSE: wave-uniform passes
      const int e = e0 + tid;
      const int ec = min(e, nent - 1);   // idle lanes repeat the last entry and store nothing
      const int kr = (int)((unsigned)ec / (unsigned)K), m = ec - kr * K;
      // have_first: this lane's Gauss-Legendre node of the first pass of the first ring group was requested at the start
      // of the kernel and waits in the (empty) queue
      const double tk = (have_first && k0 == 0 && e0 == 0) ? lw[W.stash + lane] : P.glt[k0 + kr];
      const double mu = fma(hw, tk, hm);
      const double sig2 = max_raw(fma(-mu, mu, 1.0), 1e-300);
      const double isig = rsqrt_nr(sig2);
      const double sig = sig2 * isig;
      entry(m, e, e < nent, mu, sig, isig);
    }
  }
  pair_sync<WPP>();
}

// Layout of the cos/sin(m psi_l) table (host: upload_quadrature).  Up to L = 6 l-major: the orders of one azimuth are
// adjacent, a lane reads them with immediate offsets from one address (m-major costs a 64-bit address computation per
// order, ~10 VALU per slab).  Above, m-major: a lane's orders would span 16 (L - 1) > 128 bytes and every wave load
// would touch one cache line per lane (A/B at L = 12, n_q = 32: l-major +1.8 %), and at L = 7 the l-major form costs a spilled register.
__host__ __device__ constexpr bool trig_lmajor(int L) { return L >= 2 && L <= 6; }

// r_i (and its mu / psi derivatives) at ring row `row`, azimuth (c1, s1) = (cos psi, sin psi)
template <int L, bool GRAD>
__device__ __forceinline__ void ring_eval(const double* __restrict__ row, const int LL, const double c1, const double s1,
                                          const double* __restrict__ tr, const int tstride, double& r, double& rmu,
                                          double& rpsi)
{
  // cos/sin(m psi) of this lane's azimuth: compiled orders read them from the host-built table `tr`
  // (m = 2..L, 16 bytes per m, vector memory loads that cost no VALU slot); the run-time-order kernel keeps
  // the Chebyshev recurrence (4 FP64 operations per m).
  r = row[0];
  rmu = GRAD ? row[2] : 0.0;
  rpsi = 0.0;
  double cm = c1, sm = s1;
  const int lim = (L >= 0) ? L : LL;
#pragma unroll
  for (int m = 1; m <= lim; ++m) {
    if (L >= 2 && m >= 2) {
      cm = tr[trig_lmajor(L) ? 2 * (m - 2) : (m - 2) * tstride];
      sm = tr[(trig_lmajor(L) ? 2 * (m - 2) : (m - 2) * tstride) + 1];
    }
    const v2d ab = lds2(row + 4 * m);   // (A_km, B_km): one ds_read_b128
    const double A = ab[0], B = ab[1];
    r = fma(A, cm, r);
    r = fma(B, sm, r);
    if (GRAD) {
      const v2d dab = lds2(row + 4 * m + 2);
      rmu = fma(dab[0], cm, rmu);
      rmu = fma(dab[1], sm, rmu);
      const double dm = (double)m;
      rpsi = fma(dm * B, cm, rpsi);
      rpsi = fma(-dm * A, sm, rpsi);
    }
    if (L < 2 && m < lim) {
      const double c = fma(cm, c1, -(sm * s1)), s = fma(cm, s1, sm * c1);
      cm = c;
      sm = s;
    }
  }
}

// ---- particle j in the pair's COMMON frame ------------------------------------------------------------------------
// Every point at which a pair evaluates r_j — a cap node's surface point r_i u, or a point x_i + lambda u of the
// node's ray in the inner-radius search — lies in the half-plane through the line of centres that contains u: seen
// from x_j in the frame (e1, e2, c) it has the node's azimuth psi_l, and only its polar angle varies,
//   cos(theta_j) = (lambda mu_k - rho) / s,   sin(theta_j) = lambda sigma_k / s,   s^2 = lambda^2 - 2 lambda mu_k rho + rho^2.
// So particle j gets the treatment of particle i: its expansion is rotated into the common frame (the same
// cap_frame_rotate with M_j = [R_j^T e1, R_j^T e2, R_j^T c], whose Euler angles come with the pair record), where
//   r_j(mu, psi) = sum_m sigma^m [cos(m psi) Wc_m(mu) + sin(m psi) Ws_m(mu)],   sigma = sqrt(1 - mu^2).
// For a FIXED azimuth the even orders sum to a polynomial G_l(mu) of degree L (sigma^m = (1 - mu^2)^(m/2)) and the odd
// ones to sigma H_l(mu), H_l of degree L - 1:   r_j = G_l(mu_j) + sigma_j H_l(mu_j)   — 2L + 1 coefficients and 2L + 1
// v_fma_f64 per evaluation instead of (L+1)^2 coefficients and ~(L+1)^2 + 4L operations of a body-frame evaluation
// (L = 6: 13 against 69, and no direction in j's body frame: 14 more), kept in VGPRs across the inner-radius
// iterations (and across phase 1, where a lane's azimuth does not change when 2 n_q divides 64).  The azimuths
// psi_l and psi_(l + n_q) = psi_l + pi share a row: G is the same, H changes sign.
// Built per pair in two steps from the rotated, scaled vector v0 (both sparse matrix-vector products):
//   1. PJ[2m + part][k] = sum_n v0[n^2 + n +- m] E_nm[k]   (host table P.jval / P.jcol, ELL rows; sh_tables.cpp)
//   2. G_l[k] = sum_(m even) cos(m psi_l) PJ[2m][k] + sin(m psi_l) PJ[2m+1][k],  H_l likewise over the odd m.
// The first-stage rows of a lane (NP passes of 64 rows, XW entries each) and the cos/sin of its orders for the first
// 16 azimuths: constants of the launch, requested at the very start of the kernel so that their latency runs
// beside that of the pair's record (small orders only: 24 + 16 registers at L = 6).
template <int L>
struct JPolyPre {
  static constexpr int K = L + 1, NR = jpoly_rows(L) * K, XW = L / 2 + 1, NP = (NR + 63) / 64, NM = L / 2 + 1;
  static constexpr bool on = NP * XW <= 8;
  double val[on ? NP * XW : 1];
  int col[on ? NP * XW : 1];
  double cs[NM], sn[NM];
  __device__ __forceinline__ void fetch(const PairParams& P, const int lane, const int nq)
  {
    if constexpr (on) {
#pragma unroll
      for (int ps = 0; ps < NP; ++ps) {
        const int o = lane + 64 * ps;
        const size_t at = (size_t)(o < NR ? o : 0) * XW;
#pragma unroll
        for (int t = 0; t < XW; ++t) {
          val[ps * XW + t] = P.jval[at + t];
          col[ps * XW + t] = P.jcol[at + t];
        }
      }
    }
    const int l = lane & 15, par = (lane >> 4) & 1;
    const double* tj = P.trigj + (size_t)(l < nq ? l : 0) * (2 * (L + 2)) + 2 * par;
#pragma unroll
    for (int a = 0; a < NM; ++a) {
      cs[a] = tj[4 * a];
      sn[a] = tj[4 * a + 1];
    }
  }
};

// WPP = 2 (two waves per pair): both waves take rows of the first stage (stride 128) and azimuth passes of the second
// (wave h the passes h, h + 2, ...); `half` is the wave's index within the pair.
template <int L, int WPP = 1>
__device__ __forceinline__ void jpoly_build(const PairParams& P, double* __restrict__ lw, const WaveLdsLayout& W,
                                            const int lane, const int nq, const JPolyPre<L>& pre, const double glw_first,
                                            const int half = 0)
{
  // K powers per polynomial; the tables carry one order more than exist (m = L + 1: empty rows of PJ, a real
  // cos/sin pair) so that the azimuth stage below needs no guard on its reads
  constexpr int K = L + 1, NR = jpoly_rows(L) * K, NRI = jpoly_pi_doubles(L), XW = L / 2 + 1, RS = jpoly_row(L);
  // first stage for BOTH particles from one pass over the table rows: particle j's polynomials feed the azimuth stage
  // below, particle i's (the real orders only) are what the ring tables are evaluated from (cap_frame_rings_poly)
  const double* v0 = lw + W.v0;
  const double* v0i = lw + W.v0i;
  double* pj = lw + W.pj;
  double* pi = lw + W.pi;
  if constexpr (JPolyPre<L>::on && WPP == 1) {
#pragma unroll
    for (int ps = 0; ps < JPolyPre<L>::NP; ++ps) {
      const int o = lane + 64 * ps;
      double acc = 0.0, aci = 0.0;
#pragma unroll
      for (int t = 0; t < XW; ++t) {
        acc = fma(pre.val[ps * XW + t], v0[pre.col[ps * XW + t]], acc);
        aci = fma(pre.val[ps * XW + t], v0i[pre.col[ps * XW + t]], aci);
      }
      if (o < NR) pj[o] = acc;
      if (o < NRI) pi[o] = aci;
    }
  } else {
    for (int o = lane + 64 * half; o < NR; o += 64 * WPP) {
      const double* val = P.jval + (size_t)o * XW;
      const int* col = P.jcol + (size_t)o * XW;
      double acc = 0.0, aci = 0.0;
#pragma unroll
      for (int t = 0; t < XW; ++t) {
        acc = fma(val[t], v0[col[t]], acc);
        aci = fma(val[t], v0i[col[t]], aci);
      }
      pj[o] = acc;
      if (o < NRI) pi[o] = aci;
    }
  }
  pair_sync<WPP>();
  // the Gauss-Legendre weights go into the odd slot of the table's rows now that the rotated vectors are out of them
  for (int t = lane + 64 * half; t < nq; t += 64 * WPP) lw[W.glw + t * RS] = (t < 64 * WPP) ? glw_first : P.glw[t];
  // Azimuth stage.  Lanes are (azimuth l, parity of m, parity of k), 16 azimuths per pass: a lane loads the
  // cos/sin(m psi_l) of its orders m = par, par + 2, ... once and walks its powers k = kq, kq + 2, ...; every LDS
  // address is the lane's base plus an immediate.  G (par = 0) has the powers 0..L, H (par = 1) the powers 0..L-1.
  double* gh = lw + W.gh;
  constexpr int NM = L / 2 + 1;                  // orders of one parity (the last may be the empty order L + 1)
  const int par = (lane >> 4) & 1, kq = lane >> 5;
  const int kmax = L - par;
  const double* pjl = pj + (2 * par) * K + kq;   // PJ[2 (2a + par) + part][kq + 2 b] at pjl[(4 a + part) K + 2 b]
  for (int l0 = 16 * half; l0 < nq; l0 += 16 * WPP) {
    const int l = l0 + (lane & 15);
    const bool lok = l < nq;
    double cs[NM], sn[NM];
    if (l0 == 0) {   // wave-uniform: requested at the start of the kernel
#pragma unroll
      for (int a = 0; a < NM; ++a) {
        cs[a] = pre.cs[a];
        sn[a] = pre.sn[a];
      }
    } else {
      const double* tj = P.trigj + (size_t)(lok ? l : 0) * (2 * (L + 2)) + 2 * par;   // (cos, sin)(m psi_l) at tj[4a], tj[4a+1]
#pragma unroll
      for (int a = 0; a < NM; ++a) {
        cs[a] = tj[4 * a];
        sn[a] = tj[4 * a + 1];
      }
    }
    if (lok && kq == 0 && par == 1) {   // the row's own cos(psi_l), sin(psi_l): the first order of the odd lanes
      double* tw = jpoly_trig_sep(L) ? lw + W.tr + 2 * l : gh + l * RS + jpoly_trig(L);
      tw[0] = cs[0];
      tw[1] = sn[0];
    }
    // column of the power k in a row: G: L - k; H: 2L - k  (descending powers, Horner order)
    double* out = gh + (lok ? l : 0) * RS + (par ? 2 * L : L) - kq;
#pragma unroll
    for (int b = 0; b <= L / 2; ++b) {
      double acc = 0.0;
#pragma unroll
      for (int a = 0; a < NM; ++a) {
        acc = fma(cs[a], pjl[(4 * a) * K + 2 * b], acc);
        acc = fma(sn[a], pjl[(4 * a + 1) * K + 2 * b], acc);
      }
      if (lok && kq + 2 * b <= kmax) out[-2 * b] = acc;
    }
  }
  pair_sync<WPP>();
}

// r_j at polar angle (mu, sigma) of the common frame from a lane's row of the per-azimuth table; `sig` carries the
// sign of the azimuth's half (l >= n_q: -).  The row is read from LDS at every evaluation: the 64 lanes of a wave
// address at most n_q distinct rows (the hardware broadcasts), 7 ds_read_b128 at L = 6 beside ~25 v_fma_f64 — and the
// 2L + 1 coefficients do not sit in 4L + 2 registers through the node loops (held there they cost the kernel a wave
// per SIMD, and with the waves the cover for its dependent FP64 chains: 11 cycles from one v_fma_f64 to the next).
template <int L>
__device__ __forceinline__ double jpoly_eval(const double* __restrict__ row, const double mu, const double sig)
{
  // 2L + 1 coefficients in L + 1 aligned 16-byte pairs (the second half of the last pair is the ring weight)
  v2d c[L + 1];
#pragma unroll
  for (int t = 0; t <= L; ++t) c[t] = lds2(row + 2 * t);
  double g = c[0][0];
#pragma unroll
  for (int t = 1; t <= L; ++t) g = fma(g, mu, c[t >> 1][t & 1]);
  if constexpr (L >= 1) {
    double h = c[(L + 1) >> 1][(L + 1) & 1];
#pragma unroll
    for (int t = L + 2; t <= 2 * L; ++t) h = fma(h, mu, c[t >> 1][t & 1]);
    g = fma(sig, h, g);
  }
  return g;
}

// mu- and psi-derivative of r_i at a node of ring row `row` for the JPT kernels, from (cos psi, sin psi) alone: the
// higher orders by the angle-addition recurrence (4 v_fma_f64 per order), no table is read.
template <int L>
__device__ __forceinline__ void ring_grad_rec(const double* __restrict__ row, const double c1, const double s1, double& rmu,
                                              double& rpsi)
{
  rmu = row[2];
  rpsi = 0.0;
  // cos / sin((m + 1) psi) = 2 cos(psi) cos / sin(m psi) - cos / sin((m - 1) psi): ONE v_fma_f64 each (the angle
  // addition form costs two; the three-term form loses ~m^2 ulp, 1e-14 at L = 12, far inside the 1e-9 bar)
  double cm = c1, sm = s1, cp = 1.0, sp = 0.0;
  const double tc = c1 + c1;
#pragma unroll
  for (int m = 1; m <= L; ++m) {
    const v2d ab = lds2(row + 4 * m), dab = lds2(row + 4 * m + 2);   // two ds_read_b128 per order
    const double A = ab[0], B = ab[1], dm = (double)m;
    rmu = fma(dab[0], cm, rmu);
    rmu = fma(dab[1], sm, rmu);
    const double t = fma(B, cm, -(A * sm));   // three instructions per order (m B and m A as products of their own: four)
    rpsi = (m == 1) ? t : fma(dm, t, rpsi);
    if (m < L) {
      const double c = fma(tc, cm, -cp), s = fma(tc, sm, -sp);
      cp = cm;
      sp = sm;
      cm = c;
      sm = s;
    }
  }
}

template <bool B>
struct BoolC { static constexpr bool value = B; };
// r_i at a node of ring row `row` from (cos psi, sin psi), the same recurrence (a direct batch computes it a second time
// behind the inner-radius search instead of carrying it through, see DIRECT in pair_contact_kernel)
template <int L>
__device__ __forceinline__ double ring_value(const double* __restrict__ row, const double c1, const double s1)
{
  double r = row[0];
  double cm = c1, sm = s1, cp = 1.0, sp = 0.0;
  const double tc = c1 + c1;
#pragma unroll
  for (int m = 1; m <= L; ++m) {
    const v2d ab = lds2(row + 4 * m);
    r = fma(ab[0], cm, fma(ab[1], sm, r));
    if (m < L) {
      const double c = fma(tc, cm, -cp), s = fma(tc, sm, -sp);
      cp = cm;
      sp = sm;
      cm = c;
      sm = s;
    }
  }
  return r;
}

// Two evaluations from one pass over the row (phase 1: the two nodes of a lane's pair share it)
template <int L>
__device__ __forceinline__ void jpoly_eval2(const double* __restrict__ row, const double mua, const double siga,
                                            const double mub, const double sigb, double& ra, double& rb)
{
  v2d cc[L + 1];
#pragma unroll
  for (int t = 0; t <= L; ++t) cc[t] = lds2(row + 2 * t);
  const double c0 = cc[0][0];
  double ga = c0, gb = c0;
#pragma unroll
  for (int t = 1; t <= L; ++t) {
    const double c = cc[t >> 1][t & 1];
    ga = fma(ga, mua, c);
    gb = fma(gb, mub, c);
  }
  if constexpr (L >= 1) {
    const double h0 = cc[(L + 1) >> 1][(L + 1) & 1];
    double ha = h0, hb = h0;
#pragma unroll
    for (int t = L + 2; t <= 2 * L; ++t) {
      const double c = cc[t >> 1][t & 1];
      ha = fma(ha, mua, c);
      hb = fma(hb, mub, c);
    }
    ga = fma(siga, ha, ga);
    gb = fma(sigb, hb, gb);
  }
  ra = ga;
  rb = gb;
}

// WEIGHTED (SPEC §2.8): phase 1 keeps the residuals g~ of three consecutive slabs in registers, so that a
// node's azimuth and ring neighbours are a cross-lane read away, and queues every node with a positive
// covered fraction together with that fraction; phase 2 scales the node's weight by it.  n_q <= 32 (a ring
// neighbour is at most one slab away) and ring groups of at least two slabs' worth of rings: checked on the host.
// WPP = 2 (JPT kernels): two waves per pair, see pair_lds_layout2 — the workgroup is the pair, `half` the wave's half
// of the azimuths; every table build runs on 128 lanes and every hand-over between the waves is a workgroup barrier
// that BOTH waves reach the same number of times (the ring-group loop advances identically in both).
// Specialised instances (round 5).  n_q, the resident ring rows and the queue capacity are launch parameters of the
// per-azimuth kernels: every node's (ring, azimuth) comes out of a multiply-shift division by 2 n_q or n_q, every row
// address out of a multiplication by the row length, every ring-group bound out of a compare with the group size.
// With the three as compile-time constants the divisions become shifts and masks, the products immediates, the
// one-group case loses its group loop: -4.3 % at the headline with bitwise-equal results (profiles/r05_ab_nq_const.txt,
// an experiment build with the constants forced).  One instance per compiled order, for the (n_q, rows, queue) the
// host's rules pick at that order's BASELINE shape — PairSpec<L> — launched when the launch's parameters are exactly
// those (pair_spec_matches; option "spec" 0 keeps the general kernels, which every other (L, n_q) runs anyway).
template <int L> struct PairSpec { static constexpr int nq = 0, rr = 0, qc = 0, wpp = 1; };
template <> struct PairSpec<4> { static constexpr int nq = 10, rr = 10, qc = 128, wpp = 1; };    // configs[0]'s shape
template <> struct PairSpec<6> { static constexpr int nq = 16, rr = 16, qc = 172, wpp = 1; };    // configs[1], [2], [3]
template <> struct PairSpec<12> { static constexpr int nq = 32, rr = 12, qc = 148, wpp = 2; };   // configs[4]
template <int L>
inline bool pair_spec_matches(const PairParams& P)
{
  typedef PairSpec<(L >= 0 ? L : 0)> S;
  return L >= 0 && S::nq > 0 && P.spec && P.jpoly && !P.rule && P.nq == S::nq && P.ring_rows == S::rr && P.qcap == S::qc &&
         (P.split ? 2 : 1) == S::wpp && P.waves_per_block == 1;
}
inline bool pair_spec_matches_rt(const int L, const PairParams& P)
{
  return L == 4 ? pair_spec_matches<4>(P) : (L == 6 ? pair_spec_matches<6>(P) : (L == 12 ? pair_spec_matches<12>(P) : false));
}

template <int L, bool NEEDV, bool WEIGHTED = false, bool JPT = false, int WPP = 1, bool SPEC = false>
__global__ void __launch_bounds__(64 * kMaxWavesPerBlock, JPT ? SHP_JMIN_WAVES(L, NEEDV, WPP) : (WEIGHTED ? SHP_WMIN_WAVES(L) : SHP_MIN_WAVES(L, NEEDV))) pair_contact_kernel(const PairParams P)
{
  static_assert(!SPEC || (JPT && L >= 0 && !WEIGHTED && PairSpec<(L >= 0 ? L : 0)>::nq > 0 && PairSpec<(L >= 0 ? L : 0)>::wpp == WPP),
                "specialised instances: per-azimuth kernels of the orders PairSpec names");
  static_assert(WPP == 1 || (WPP == 2 && JPT && L >= 0 && !WEIGHTED), "two waves per pair: compiled-order JPT kernels only");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  int lane = threadIdx.x & 63;
  const int wib = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int w = P.slot0 + ((WPP == 2) ? (int)blockIdx.x : __builtin_amdgcn_readfirstlane((int)(blockIdx.x * P.waves_per_block)) + wib);
  const int half = (WPP == 2) ? wib : 0;      // wave-uniform
  const int tid = lane + 64 * half;           // lane within the pair's waves
  constexpr int NT = 64 * WPP;
  (void)tid;
  if (w >= P.npairs) return;
  const int LL = (L >= 0) ? L : P.lmax;
  typedef PairSpec<(L >= 0 ? L : 0)> Spec;
  const int nq = SPEC ? Spec::nq : P.nq;                    // (SPEC: compile-time constants, see PairSpec)
  const int P_ring_rows = SPEC ? Spec::rr : P.ring_rows;
  const int P_qcap = SPEC ? Spec::qc : P.qcap;
  // compiled orders: particle j from per-azimuth polynomials in the pair's common frame (jpoly_build above); the
  // run-time-order kernel keeps the body-frame evaluation sh_eval_rt
  constexpr bool JP = JPT && (L >= 0) && !WEIGHTED;
  constexpr int LJ = JP ? L : 0;
  // CARRY: the node (ring, azimuth, weight, mu, sigma) stays in registers across the inner-radius search where the kernel
  // has them to spare (see phase 2); DIRECT (every per-azimuth kernel): a slab whose inside nodes do not fit the queue
  // becomes a batch of its own
#ifndef SHP_CARRY_NODE
#define SHP_CARRY_NODE(L, WPP) ((L) >= 6 && !((L) == 9 && (WPP) == 1))
#endif
  constexpr bool CARRY = JP && SHP_CARRY_NODE(L, WPP);
#ifndef SHP_DIRECT
#define SHP_DIRECT(L) 1
#endif
  constexpr bool DIRECT = SHP_DIRECT(L) && JP && !WEIGHTED;
  constexpr int FRAME = JP ? kFrameJ : kFrame;   // doubles of the frame in LDS; FRM(slot): where a record slot sits in it
#define FRM(slot) (JP ? frj(slot) : (slot))
  WaveLdsLayout W = (WPP == 2) ? pair_lds_layout2(LL, P_ring_rows, nq, P_qcap)
                               : wave_lds_layout(LL, P_ring_rows, WEIGHTED, JP ? nq : 0, JP ? P_qcap : kQueue);
  if constexpr (WPP == 2) {   // this wave's queue
    W.qri += half * W.qstride;
    W.qrj += half * W.qstride;
    W.qp += half * W.qstride;
    W.park += half * W.qstride;
  }
  // The frame and ring tables are loop invariant: a plain LDS load would be
  // hoisted out of the node loops and pinned in VGPRs, which is what they are
  // in LDS to avoid.  Each loop iteration therefore re-derives its base pointer
  // from a byte offset laundered through an empty asm (an integer, so that the
  // compiler still sees an LDS address and emits ds_read, not flat loads).
  // The wave's LDS as an ABSOLUTE 32-bit LDS address in a scalar register: pointers made from it are an inttoptr, so an
  // address is one v_add with the scalar as an operand (through `smem_raw + offset` every re-derivation was a v_mov of
  // the offset and a v_add of the array's link-time address, 0: two issue slots, several times per slab and batch).
  typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
  const unsigned wave_off = (WPP == 2) ? 0u : (unsigned)(wib * P.wave_lds_bytes);
  const unsigned wave_abs = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_byte_t*)smem_raw + wave_off);
  // (the body-frame kernels are out of scalar registers: they keep the array-relative form)
#define SHP_LDS() (JP ? (double*)(lds_byte_t*)(size_t)launder_s32(wave_abs) : (double*)(smem_raw + launder_u32(wave_off)))
  double* lw = SHP_LDS();

  // the pair's record (pair_setup.hpp): scalar loads of the four ints, one coalesced vector load of the frame
  // Wave priority: the prologue, the table builds and the epilogue are chains of dependent memory / LDS round trips
  // with a few instructions in between; they issue ahead of the waves that are in their node loops (priority 0), so
  // that a pair's latency-bound stretches are as short as the memory system allows (-1.8 % at the headline).
  __builtin_amdgcn_s_setprio(3);
  const int* rid = P.rec_i + 4 * (size_t)w;
  // Everything the prologue reads from memory is requested before the slot's status is looked at (a scalar load of
  // its own: waiting for it first would put two memory round trips in a row at the start of every pair).  The
  // addresses do not depend on the status; a dead slot's rows are in bounds and never used.
  constexpr int NSL = (JPT && L >= 0 && !WEIGHTED) ? ((L + 1) * (L + 1) + NT - 1) / NT : 1;
  double vi[NSL], vj[NSL];
  JPolyPre<(JPT && L >= 0 && !WEIGHTED) ? L : 0> pre;
  const double recv = P.rec[(size_t)kRecStride * w + (lane < kRecUsed ? lane : 0)];
  if constexpr (JPT && L >= 0 && !WEIGHTED) {
    constexpr int ns = (L + 1) * (L + 1);
    // element e = l^2 + r of the slot's two rotations 2w (particle i) and 2w + 1 (particle j): adjacent in their tile
#pragma unroll
    for (int t = 0; t < NSL; ++t) {
      const int e = tid + NT * t;
      const int ec = e < ns ? e : 0;
      if constexpr (rot_tiled(L)) {
        // l = floor(sqrt(e)) from the bare v_sqrt_f32 (half an integer of margin against its last bits; sqrtf() is the
        // IEEE sequence, ~18 instructions); the address as a scalar tile base plus a 32-bit lane offset = rot_index()
        const int l = (int)__builtin_amdgcn_sqrtf((float)ec + 0.5f);
        const double* tile = P.rot + (size_t)((2 * w) >> 6) * rot_tile_doubles(L);
        const unsigned at = 64u * (unsigned)(l * l) + (unsigned)((2 * w) & 63) * (unsigned)(2 * l + 1) + (unsigned)(ec - l * l);
        vi[t] = tile[at];
        vj[t] = tile[at + (unsigned)(2 * l + 1)];
      } else {
        const size_t at = (size_t)(2 * w) * rot_row_doubles(L) + ec;
        vi[t] = P.rot[at];
        vj[t] = P.rot[at + rot_row_doubles(L)];
      }
    }
    pre.fetch(P, lane, nq);
  }
  double glt_first = 0.0, glw_first = 0.0;
  if constexpr (JPT && L >= 0 && !WEIGHTED) glw_first = P.glw[tid < nq ? tid : 0];   // the weight of ring `tid`, stored after the first stage
  if constexpr (JPT && L >= 0 && L <= 8 && !WEIGHTED && WPP == 1) {
    const int nr0 = P_ring_rows < nq ? P_ring_rows : nq;   // rings of the first group
    const int lg0 = ring_poly_map(nr0, L + 1, 64);             // cap_frame_rings_poly's lane map of that group's first pass
    const int kr = lg0 >= 1 ? (lane >> lg0) : lane / (L + 1);
    glt_first = P.glt[kr < nr0 ? kr : nr0 - 1];
  }
  const int status = rid[0];
  if (status == 0) return;   // bounding spheres apart (SPEC §2.1) or a shape index outside the table; wave-uniform
  const int si = rid[1], sj = rid[2];
  (void)si; (void)sj;
  // JPT: the pair's scalars as scalar loads of its record (SGPR operands), not LDS reads at the head of every loop
  // iteration — the kernel has the scalar registers now that no coefficient windows live in them
  double s_rho = 0.0, s_rj = 0.0, s_rj2 = 0.0, s_rho2 = 0.0, s_pj = 0.0, s_tol1 = 0.0, s_tol3 = 0.0, s_tiny = 0.0;
  if constexpr (JPT && L >= 0 && !WEIGHTED) {
    const cdptr rs = launder_uniform(P.rec + (size_t)kRecStride * w);
    s_rho = rs[FR_RHO];
    s_rj = rs[FR_RJ];
    s_rj2 = rs[FR_RJ2];
    s_rho2 = rs[FR_RHO2];
    s_pj = rs[FR_JPJ];
    s_tol1 = rs[FR_JTOL1];
    s_tol3 = rs[FR_JTOL3];
    s_tiny = rs[FR_JTINY];
  }
  const bool centre_in_bj = rid[3] != 0;  // rho < Rj
  if constexpr (JP) {
    if ((tid >= 12 && tid < 30) || (tid >= 36 && tid < kRecUsed)) lw[frj(tid)] = recv;
  } else {
    if (tid < kRecUsed) lw[tid] = recv;
  }
  if constexpr (JP && L <= 8 && WPP == 1) lw[W.stash + lane] = glt_first;
  if constexpr (JP) {
    // both rotated vectors, side by side in the (not yet built) rows of particle j's table: the first stage reads them
#pragma unroll
    for (int t = 0; t < NSL; ++t)
      if (tid + NT * t < (LJ + 1) * (LJ + 1)) {
        lw[W.v0 + tid + NT * t] = vj[t];
        lw[W.v0i + tid + NT * t] = vi[t];
      }
  }
  pair_sync<WPP>();
#if defined(SHP_ABL) && SHP_ABL == 1   // timing-only build: stop after the pair prologue
  asm volatile("" ::"v"(lw[lane & 31]));
  return;
#endif
  if constexpr (JP) {
    jpoly_build<LJ, WPP>(P, lw, W, lane, nq, pre, glw_first, half);
  } else {
    cap_frame_rotate<L>(P, lw, W, LL, si, lane);
  }
#if defined(SHP_ABL) && SHP_ABL == 4   // timing-only build: stop after the coefficient rotation
  asm volatile("" ::"v"(lw[(JP ? W.pi : W.v0) + lane]));
  return;
#endif

  const double* rc = P.rc;
  // particle j's coefficients (body-frame family): wave-uniform, fetched with scalar loads.  (Staged in LDS instead —
  // what north_star suggests — every term costs a broadcast ds_read: +40 % at L = 6, +87 % at L = 12, round 1 A/B.)
  const double* cwj = P.coef + (size_t)sj * P.cstride;
  const int lrt = P.lmax;
  (void)rc; (void)lrt;
  const double* fr = SHP_LDS();

  // SPEC §2.6: centre of i inside j (only possible when rho < Rj)
  bool centre_inside = false;
  if (NEEDV && centre_in_bj) {
    const double rho = fr[FRM(FR_RHO)];
    double rj;
    if constexpr (JP) {
      // x_i seen from x_j lies on the axis, opposite to c: mu = -1, sigma = 0, any azimuth
      rj = jpoly_eval<LJ>(fr + W.gh, -1.0, 0.0);
    } else {
      const double ir = rcp_nr(rho);
      rj = sh_eval<L>(rc, cwj, lrt, -fr[FR_DJ] * ir, -fr[FR_DJ + 1] * ir, -fr[FR_DJ + 2] * ir);
    }
    centre_inside = (rho - rj <= 0.0);
  }

  const int npsi = 2 * nq;
  const int Q = nq * npsi;
  // (integer products here go through the 24-bit multiplier — v_mul_u32_u24, full rate; v_mul_lo_u32 is a quarter-rate
  // instruction and the node loops had three to six of them per slab / batch)
  // p / npsi for 0 <= p < Q <= 2^15 as a multiply-shift: exact because
  // magic * npsi - 2^24 < npsi <= 256 < 2^24 / 2^15
  const unsigned magic = ((1u << 24) + (unsigned)npsi - 1u) / (unsigned)npsi;
  // lanes per ring in phase 1: the JPT kernels give a lane the node PAIR (k, l), (k, l + n_q) — the two azimuths of
  // a row of particle j's table, and r_i at both from one pass over the ring row (even orders + / - odd orders)
  // (two waves per pair: each wave half of them, the azimuths half n_q / 2 ... (half + 1) n_q / 2 - 1; n_q is even there)
  const int per_ring = JP ? nq / WPP : npsi;
  const unsigned magicr = ((1u << 24) + (unsigned)per_ring - 1u) / (unsigned)per_ring;
  const int nslabs = (nq * per_ring + 63) >> 6;
  const int rowlen = 4 * (LL + 1);

  // JPT kernels: the ring tables of the FIRST group are built here, before any of the seven sums exists — with all
  // rings resident (the common case) that is the only build of the pair, and nothing has to be parked around it (round
  // 3 parked two sums in the queue for every build: 128 doubles of LDS beside the polynomials the build reads)
  if constexpr (JP) {
    const int kend0 = (P_ring_rows < nq) ? P_ring_rows : nq;
    cap_frame_rings_poly<LJ, WPP>(P, SHP_LDS(), W, lane, tid, 0, kend0, fr[FRM(FR_HW)], fr[FRM(FR_HM)], L <= 8 && WPP == 1);
  }
  bool first_group = true;   // wave-uniform
  double aV = 0.0, aS0 = 0.0, aS1 = 0.0, aS2 = 0.0, aT0 = 0.0, aT1 = 0.0, aT2 = 0.0;
  int qhead = 0, qcount = 0, slab = 0;  // wave-uniform
  double wg1 = 0.0, wg2 = 0.0, wri1 = 0.0, wrj1 = 0.0;  // WEIGHTED: residuals of slabs t-1, t-2; r_i, r_j of slab t-1
  bool win1 = false;
  const bool aligned = (npsi <= 64) && ((64 % npsi) == 0);  // wave-uniform

  // Ring groups: the tables of P.ring_rows consecutive rings are resident at a time (all nq of
  // them unless that would starve the CU of waves); the queue is drained at the end of a group.
  // WEIGHTED: iteration t = slab computes slab t and weighs slab t - 1, so a group starts at the ring of the
  // still unweighed slab (its nodes are queued, and read their ring rows, only after the switch) and the last
  // group runs one iteration past the last slab.  The host sizes ring_rows so that every group advances.
  while (WEIGHTED ? (slab <= nslabs) : (slab < nslabs)) {
  const int sfirst = (WEIGHTED && slab > 0) ? slab - 1 : slab;
  const int k0 = (int)(((unsigned)(sfirst << 6) * magicr) >> 24);
  const int kend = (k0 + P_ring_rows < nq) ? k0 + P_ring_rows : nq;
  const int slab_end = (kend == nq) ? (WEIGHTED ? nslabs + 1 : nslabs) : ((kend * per_ring) >> 6);
  if (slab_end <= slab) return;  // cannot happen with the host's ring_rows; never spin
  __builtin_amdgcn_s_setprio(3);
  {
    double* lr = SHP_LDS();
    // the queue is empty between ring groups: four of the seven sums wait there while the ring tables are built
    // (eight registers the build has for its recurrences instead of spilling)
    if constexpr (JP) {
      if (!first_group) {   // later groups (ring tables in pieces): two sums wait in the empty queue meanwhile (they would be spilled otherwise)
        double* park = lr + W.park + lane;
        park[0] = aT2; park[64] = NEEDV ? aV : aS0;
        if constexpr (WPP == 2) __syncthreads();   // the other wave has left the node loops of the previous group: its rows may go
        cap_frame_rings_poly<LJ, WPP>(P, lr, W, lane, tid, k0, kend - k0, lr[FRM(FR_HW)], lr[FRM(FR_HM)], false);
        park = SHP_LDS() + W.park + lane;
        aT2 = park[0];
        if (NEEDV) aV = park[64]; else aS0 = park[64];
        wave_lds_sync();
      }
      first_group = false;
    } else {
      double* park = lr + W.qri + lane;
      park[0] = aT0; park[64] = aT1; park[128] = aT2; park[192] = NEEDV ? aV : aS0;
      cap_frame_rings<L>(P, lr, W, LL, lane, k0, kend - k0, lr[FR_HW], lr[FR_HM]);
      park = SHP_LDS() + W.qri + lane;
      aT0 = park[0]; aT1 = park[64]; aT2 = park[128];
      if (NEEDV) aV = park[192]; else aS0 = park[192];
      wave_lds_sync();
    }
#if defined(SHP_ABL) && SHP_ABL == 2   // timing-only build: stop after rotation + ring tables
    asm volatile("" ::"v"(lr[W.ring + lane]));
    return;
#endif
  }

  // Phase 2 as a lambda, instantiated twice by the kernels that take DIRECT batches: from the queue (DIR false), and on the
  // lanes' own nodes (dp, dri, drj; lanes mdir) when a slab's inside nodes do not fit the queue.
  auto phase2 = [&](auto dir_c, const int dp, const double dri, const double drj, const unsigned long long mdir)
                    __attribute__((always_inline)) {
    constexpr bool DIR = decltype(dir_c)::value;
    (void)dp; (void)dri; (void)drj; (void)mdir;
    wave_lds_sync();
    fr = SHP_LDS();
    const int cnt = DIR ? 64 : (qcount < 64 ? qcount : 64);
    const bool active = DIR ? lane_of(mdir) : lane < cnt;
    // (idle lanes repeat a queued node, with weight 0: the last one through a v_min in the per-azimuth kernels)
    // per-azimuth kernels: the batch is the LAST cnt entries (a stack: what is left stays at the front, appends and
    // reads need no wrap, and the capacity need not be a power of two); the others keep a ring of kQueue entries
    if constexpr (!DIR) qcount -= cnt;
    const int e = DIR ? 0 : (JP ? qcount + min(lane, cnt - 1) : ((qhead + (active ? lane : 0)) & (kQueue - 1)));
    if constexpr (!JP) qhead = (qhead + cnt) & (kQueue - 1);
#ifdef SHP_STATS
    if (lane == 0) atomicAdd(&P.dbg[4], 1ULL);
    if (active) atomicAdd(&P.dbg[7], 1ULL);
#endif
    int p;
    double ri;
    if constexpr (DIR) {   // the lanes' own nodes; an idle lane holds a node of the slab that is not inside j: finite numbers, weight 0
      p = dp;
      ri = dri;
    } else {
      p = ((const unsigned short*)(fr + W.qp))[e];   // Q = 2 nq^2 <= 2^15
      ri = fr[W.qri + e];
    }
    int k = (int)(umul_sel<JP>((unsigned)p, magic) >> 24);
    int l = p - mul_sel<JP>(k, npsi);
    double omi = active ? fr[FRM(FR_WSC)] * (JP ? fr[W.glw + mul_sel<JP>(k, jpoly_row(LJ))] : P.glw[k]) : 0.0;   // the node's plain weight
    bool outside = false;    // WEIGHTED: a node with g~ >= 0 has no ray segment inside j
    if (WEIGHTED) outside = !(fr[W.qw + e] > 0.0);
    double c1 = P.cpsi[l], s1 = P.spsi[l];
    double mu, sig;
    {
      const double* row = fr + W.ring + (k - k0) * rowlen;
      mu = row[1];
      sig = row[3];
    }

    double rin = 0.0;
    if (NEEDV) {
      // SPEC §2.6 inner radius, all lanes in lock step
      const double rj0 = DIR ? drj : fr[W.qrj + e];
      // the node's ray seen from x_j: compiled orders (axial, signed radial) = (lambda mu - rho, +-lambda sigma) in the
      // common frame; run-time-order kernel lambda u_j - d_j in j's body frame
      double uj0, uj1, uj2 = 0.0;
      const int ghrow_ = W.gh + mul_sel<JP>(l >= nq ? l - nq : l, jpoly_row(LJ));
      if constexpr (JP) {
        uj0 = mu;
        uj1 = (l >= nq) ? -sig : sig;
      } else {
        const double a1 = sig * c1, a2 = sig * s1;
        uj0 = fma(a1, fr[FR_BJ1], fma(a2, fr[FR_BJ2], mu * fr[FR_BJC]));
        uj1 = fma(a1, fr[FR_BJ1 + 1], fma(a2, fr[FR_BJ2 + 1], mu * fr[FR_BJC + 1]));
        uj2 = fma(a1, fr[FR_BJ1 + 2], fma(a2, fr[FR_BJ2 + 2], mu * fr[FR_BJC + 2]));
      }
      bool act = active && !centre_inside && !outside;
      // three most recent points: (xa,ga) oldest, (xb,gb), (lam,gl) newest
      double lo = 0.0, hi = ri, lam, xa = ri, ga, xb = ri, gb;
      {
        double bp, s2i;
        if constexpr (JP) {
          const double rho = s_rho;
          bp = mu * rho;   // u . d: d = rho c
          const double q0 = fma(ri, uj0, -rho), q1 = ri * uj1;
          s2i = fma(q0, q0, q1 * q1);
        } else {
          const double dj0 = fr[FR_DJ], dj1 = fr[FR_DJ + 1], dj2 = fr[FR_DJ + 2];
          bp = uj0 * dj0 + uj1 * dj1 + uj2 * dj2;
          const double q0 = fma(ri, uj0, -dj0), q1 = fma(ri, uj1, -dj1), q2 = fma(ri, uj2, -dj2);
          s2i = q0 * q0 + q1 * q1 + q2 * q2;
        }
        const double rho2l = JP ? s_rho2 : fr[FR_RHO2];
        if (!centre_in_bj) lo = bp - sqrt_nr1<JP>(JP ? fma(bp, bp, -s_pj) : fma(bp, bp, -(rho2l - fr[FR_RJ2])));
        lam = bp - sqrt_nr1<JP>(fma(bp, bp, -(rho2l - rj0 * rj0)));
        if (!(lam > lo && lam < hi)) lam = 0.5 * (lo + hi);
        ga = gb = sqrt_nr1<JP>(s2i) - rj0;
      }
      if constexpr (!JP) { if (!act) lam = ri; }   // (JP: an idle lane repeats a live node's search and is never read)
      // JPT: the byte address of the node's row of particle j's table, laundered (no instruction) at every iteration so
      // that the reads stay in the loop; re-deriving it from the wave's scalar LDS offset cost three vector instructions
      unsigned jrow_addr = wave_abs + 8u * (unsigned)ghrow_;
      // the lanes still searching, as a scalar mask: the votes and the loop's exit are scalar compares, the loop counter
      // a scalar register (as a lane predicate the exit counts as divergent: counter and tests become vector code)
      unsigned long long mact;
      if constexpr (JP) mact = centre_inside ? 0ULL : (DIR ? mdir : (cnt >= 64 ? ~0ULL : ((1ULL << cnt) - 1ULL)));   // scalar arithmetic
      else mact = wave_ballot(act);
      // One iterate of the search.  The three most recent points live in three (x, g) slots that trade roles from one
      // iterate to the next — (xa,ga) oldest, (xb,gb) middle, lam the point evaluated now, gc its residual — and the
      // loop below is written three iterates long, so that no slot is ever copied into another (as a shift of the
      // history the loop carried five 64-bit moves per iterate, each an issue slot beside the FP64 work).
      auto iterate = [&](double& xa, double& ga, const double xb, const double gb, const double lam, double& gc,
                         const bool have3) __attribute__((always_inline)) {
        if constexpr (!JP) fr = SHP_LDS();
#ifdef SHP_STATS
        if (lane == 0) atomicAdd(&P.dbg[5], 1ULL);
        if (lane_of(mact)) atomicAdd(&P.dbg[6], 1ULL);
#endif
        double y0, y1, y2 = 0.0, ss2;
        if constexpr (JP) {
          y0 = fma(lam, uj0, -s_rho);
          y1 = lam * uj1;
          ss2 = fma(y0, y0, y1 * y1);
        } else {
          y0 = fma(lam, uj0, -fr[FR_DJ]);
          y1 = fma(lam, uj1, -fr[FR_DJ + 1]);
          y2 = fma(lam, uj2, -fr[FR_DJ + 2]);
          ss2 = y0 * y0 + y1 * y1 + y2 * y2;
        }
        const bool z0 = !(ss2 > 0.0);
        const double iv = rsqrt_nr1(ss2);   // ss2 = 0: NaN in r_j and g, replaced in the rare branch below
        double rj;
        if constexpr (JP) {   // the node's row, read at every iteration
          jrow_addr = launder_u32(jrow_addr);
          rj = jpoly_eval<LJ>((const double*)(lds_byte_t*)(size_t)jrow_addr, y0 * iv, y1 * iv);
        } else {
          rj = sh_eval<L>(rc, cwj, lrt, y0 * iv, y1 * iv, y2 * iv);
        }
        const double Rjl = JP ? s_rj : fr[FR_RJ];
        double gl = ss2 * iv - rj;
        if (wave_any<JP>(z0)) {   // the point sits on x_j (measure zero): a wave-uniform branch, not two selects per iteration
          asm volatile("; rare: a point on x_j");   // ... which the volatile statement keeps a branch (no if-conversion)
          gl = z0 ? -Rjl : gl;
        }
        // The update has no divergent control flow: every lane goes through it, and a lane that is done (or never was
        // active) carries on with values nobody reads — its r_in is frozen by `act`.  (Nested conditionals cost a
        // copy of each loop-carried value per merge: 19 v_mov_b64 and 84 vector instructions per iteration beside
        // the 86 of the radius evaluation.)  The common case — the extrapolated point lies strictly inside the
        // bracket, which is not yet tiny — needs no decision at all: accepted value and next iterate are both `ext`;
        // everything else (fallback to the secant, bisection, clamping, non-finite values) is a wave-uniform branch.
        {
          const bool pos = gl >= 0.0;
          lo = pos ? lam : lo;
          hi = pos ? hi : lam;
          // have3: wave-uniform (false on the first iterate only)
          const double dbl = gb - gl;
          // extrapolation to g = 0: secant through two points on the first iterate, inverse
          // quadratic interpolation (one common denominator) through three afterwards
          double ext;
          if (!have3) {
            ext = fma(gl * (lam - xb), rcp_nr1(dbl), lam);
          } else {
            const double dab = ga - gb, dal = ga - gl;
            const double num = fma(xa * gb, gl * dbl, fma(lam * ga, gb * dab, -(xb * ga) * (gl * dal)));
            ext = num * rcp_nr1(dab * dal * dbl);
          }
          const bool inb = ext > lo && ext < hi;   // false for NaN and inf
          // accept the extrapolated point
          const bool accept = JP ? (fabs(gl) <= (have3 ? s_tol3 : s_tol1)) : (fabs(gl) <= (have3 ? SHP_TAU3 : 1e-7) * Rjl);
          const bool tiny = JP ? (hi - lo <= s_tiny) : (hi - lo <= 1e-14 * Rjl);
          double res = ext, nxt = ext;
          unsigned long long mstop = wave_ballot(accept);
          if (mask_any<JP>(mact & (wave_ballot(tiny) | ~(wave_ballot(ext > lo) & wave_ballot(ext < hi))))) {
#ifdef SHP_STATS
            if (lane == 0) atomicAdd(&P.dbg[8], 1ULL);
#endif
            // the general case, lane by lane with selects: the secant is the fallback of the interpolation, the
            // midpoint the fallback of both; an accepted point is clamped to the bracket
            double sec = ext, e2 = ext;
            if (have3) {
              const double s2 = fma(gl * (lam - xb), rcp_nr1(dbl), lam);
              const bool fin0 = fabs(ext) <= 1e300;
              sec = (!fin0 || !inb) ? s2 : ext;
              e2 = fin0 ? ext : s2;
            }
            const double mid = 0.5 * (lo + hi);
            double n2 = (e2 > lo && e2 < hi) ? e2 : sec;
            n2 = (n2 > lo && n2 < hi) ? n2 : mid;
            const double accv = (fabs(e2) <= 1e300) ? fmin(fmax(e2, lo), hi) : lam;
            res = accept ? accv : (tiny ? mid : n2);
            nxt = n2;
            mstop |= wave_ballot(tiny);
          }
          rin = lane_of(mact) ? res : rin;
          mact &= ~mstop;
          gc = gl;    // the point just evaluated stays where it is ...
          xa = nxt;   // ... and the next one takes the place of the oldest: nothing moves
        }
      };
      double gl0 = 0.0;
      if constexpr (JP) {
        if constexpr (SHP_PEEL(L)) {
        // From L = 6 on the first trip stands alone: its first iterate is the secant step (two points), and written apart
        // from the loop the oldest slot's initial value is dead — no copies of r_i and g(r_i) into it, no test of the trip
        // count (A/B profiles/r04_ap_ab_peel.txt: L = 9 / 16 -1.2 %, L = 12 / 32 -0.7 %, L = 6 / 32 -1.1 %, headline -0.2 %;
        // L <= 5 measured +1 % and keep the loop as it was)
        do {
          if (!mask_any(mact)) break;
          iterate(xa, ga, xb, gb, lam, gl0, false);
          if (!mask_any(mact)) break;
          iterate(xb, gb, lam, gl0, xa, ga, true);
          if (!mask_any(mact)) break;
          iterate(lam, gl0, xa, ga, xb, gb, true);
          for (int it = 3; it < 60; it = __builtin_amdgcn_readfirstlane(it + 3)) {   // (a scalar counter, said so)
            if (!mask_any(mact)) break;
            iterate(xa, ga, xb, gb, lam, gl0, true);
            if (!mask_any(mact)) break;
            iterate(xb, gb, lam, gl0, xa, ga, true);
            if (!mask_any(mact)) break;
            iterate(lam, gl0, xa, ga, xb, gb, true);
          }
        } while (false);
        } else {
        for (int it = 0; it < 60; it = __builtin_amdgcn_readfirstlane(it + 3)) {   // (a scalar counter, said so)
          if (!mask_any(mact)) break;
          iterate(xa, ga, xb, gb, lam, gl0, it >= 1);
          if (!mask_any(mact)) break;
          iterate(xb, gb, lam, gl0, xa, ga, true);
          if (!mask_any(mact)) break;
          iterate(lam, gl0, xa, ga, xb, gb, true);
        }
        }
      } else {
        // body-frame kernels: one iterate per trip and the history shifted — their registers are spoken for: three
        // copies of the iterate, each with a full evaluation, spill
        for (int it = 0; it < 60; it = __builtin_amdgcn_readfirstlane(it + 1)) {
          if (!mask_any<JP>(mact)) break;
          iterate(xa, ga, xb, gb, lam, gl0, it >= 1);
          const double nx = xa;
          xa = xb; ga = gb; xb = lam; gb = gl0; lam = nx;
        }
      }
      // a node outside j contributes exactly nothing (r^3 - r^3 under FMA contraction is a rounding residue, and
      // V^(m-1) turns a residue of 1e-22 into a visible force)
      // CARRY: the node (ring, azimuth, weight, mu, sigma) stays in registers across the root loop where the kernel has
      // them to spare; elsewhere it is looked up a second time below
      // (from L = 6 on: six registers; up to L = 5 they would cost the sixth wave per SIMD, the one-wave kernel of L = 9
      // its fifth.  A/B profiles/r04_n_ab_carry.txt: L = 6, 7, 8 / n_q = 16 -1.6 %, -1.8 %, -1.5 %)
      // the node's r_i behind the search: a queued node reads it again from its slot, a direct batch has no slot and computes
      // it a second time from the node's ring row
      auto ri_again = [&]() __attribute__((always_inline)) {
        const int lrow = l >= nq ? l - nq : l;
        const double* tgd = jpoly_trig_sep(LJ) ? fr + W.tr + 2 * lrow : fr + W.gh + mul_sel<JP>(lrow, jpoly_row(LJ)) + jpoly_trig(LJ);
        double cd = 1.0, sd = 0.0;
        if constexpr (LJ >= 1) {
          const v2d csd = lds2(tgd);
          const double sgd = (l >= nq) ? -1.0 : 1.0;
          cd = sgd * csd[0];
          sd = sgd * csd[1];
        }
        return ring_value<LJ>(fr + W.ring + (k - k0) * rowlen, cd, sd);
      };
      if constexpr (JP && !CARRY) {
        // The node is looked up a second time here (the root loop holds a row of particle j's table in 4L + 2
        // registers and has none to carry weight, psi, mu, sigma across), from LDS only: the batch's queue slots are
        // untouched until the next phase 1, r_i and the node index are read again from the slot instead of being
        // carried through the root loop (three registers become one; a direct batch carries the index)
        if constexpr (!DIR) {
          const int e2 = (int)launder_u32((unsigned)e);
          ri = fr[W.qri + e2];
          p = ((const unsigned short*)(fr + W.qp))[e2];
        }
        p = (int)launder_u32((unsigned)p);
        k = (int)(umul_sel<JP>((unsigned)p, magic) >> 24);
        l = p - mul_sel<JP>(k, npsi);
        omi = active ? fr[FRM(FR_WSC)] * fr[W.glw + mul_sel<JP>(k, jpoly_row(LJ))] : 0.0;
        const double* row = fr + W.ring + (k - k0) * rowlen;
        mu = row[1];
        sig = row[3];
        if constexpr (DIR) ri = ri_again();
      } else if constexpr (JP) {
        if constexpr (DIR) {
          ri = ri_again();
        } else {
          const int e2 = (int)launder_u32((unsigned)e);
          ri = fr[W.qri + e2];
        }
      }
      const double dv3 = (WEIGHTED && outside) ? 0.0 : ri * ri * ri - rin * rin * rin;
      if constexpr (!JP && WEIGHTED) {
        // The weighted kernels have three slabs of residuals in registers on top of the root finder's state: the
        // node (weight, psi, mu, sigma: nine registers) is looked up a second time here, through a copy of p the
        // compiler cannot see through, instead of being carried across the loop — that is what keeps them free of
        // spills.  The sharp kernels have the room (A/B: the second lookup costs them 1.5 %).
        p = (int)launder_u32((unsigned)p);
        k = (int)(umul_sel<JP>((unsigned)p, magic) >> 24);
        l = p - mul_sel<JP>(k, npsi);
        omi = active ? fr[FR_WSC] * P.glw[k] : 0.0;
        c1 = P.cpsi[l];
        s1 = P.spsi[l];
        const double* row = fr + W.ring + (k - k0) * rowlen;
        mu = row[1];
        sig = row[3];
      }
      aV = fma(omi * (1.0 / 3.0), dv3, aV);   // the volume keeps the node's plain weight (SPEC §2.8)
    }
    if (WEIGHTED) omi *= fabs(fr[W.qw + e]);

    // surface gradient of i at the node, in the cap frame:
    //   A = r^2 u + r sigma r_mu gamma^ - (r / sigma) r_psi psi^,
    //   u = (sigma c, sigma s, mu), gamma^ = (mu c, mu s, -sigma), psi^ = (-s, c, 0)
    fr = SHP_LDS();
    double r2, rmu, rpsi;
    if constexpr (JP) {
      const double sg = (l >= nq) ? -1.0 : 1.0;
      const int lrow = l >= nq ? l - nq : l;
      const double* tg = jpoly_trig_sep(LJ) ? fr + W.tr + 2 * lrow : fr + W.gh + mul_sel<JP>(lrow, jpoly_row(LJ)) + jpoly_trig(LJ);
      if constexpr (LJ >= 1) {
        const v2d cs1 = lds2(tg);
        c1 = sg * cs1[0];
        s1 = sg * cs1[1];
      }
      ring_grad_rec<LJ>(fr + W.ring + (k - k0) * rowlen, c1, s1, rmu, rpsi);
      (void)r2;
    } else {
      ring_eval<L, true>(fr + W.ring + (k - k0) * rowlen, LL, c1, s1, P.trig + (trig_lmajor(L) ? (size_t)P.trig_stride * l : (size_t)(2 * l)), P.trig_stride, r2, rmu, rpsi);
    }
    const double rad = ri * fma(ri, sig, rmu * sig * mu);   // r (r sigma + sigma mu r_mu): multiplies (c, s)
    const double tan_ = ri * rpsi * rcp_nr1(sig);          // (r / sigma) r_psi; sigma > 0 at Gauss-Legendre nodes
    const double A0 = fma(rad, c1, tan_ * s1);
    const double A1 = fma(rad, s1, -tan_ * c1);
    const double A2 = ri * fma(ri, mu, -(sig * sig) * rmu);
    aS0 = fma(omi, A0, aS0);
    aS1 = fma(omi, A1, aS1);
    aS2 = fma(omi, A2, aS2);
    // (r u) x A, cap frame
    const double wr = omi * ri;
    const double u0 = sig * c1, u1 = sig * s1;
    aT0 = fma(wr, u1 * A2 - mu * A1, aT0);
    aT1 = fma(wr, mu * A0 - u0 * A2, aT1);
    aT2 = fma(wr, u0 * A1 - u1 * A0, aT2);
    // the queue slots just read may be overwritten by the next phase 1
    wave_lds_sync();
  };
  __builtin_amdgcn_s_setprio(0);
  for (;;) {
    // ---------------------------------------------------------------- phase 1
    // classify slabs of 64 cap nodes until 64 inside nodes are queued
    if constexpr (WEIGHTED) {
    // iteration t: residuals of slab t (if any), then the weights of slab t - 1 from slabs t - 2, t - 1, t
    while (qcount < 64 && slab < slab_end) {
      fr = SHP_LDS();
      const int t = slab;
      ++slab;
      double g0 = 0.0, ri0 = 0.0, rj00 = 0.0;
      bool in0 = false;
      if (t < nslabs) {  // wave-uniform
        const int p = (t << 6) + lane;
        const bool valid = p < Q;
        const int k = valid ? (int)(umul_sel<JP>((unsigned)p, magic) >> 24) : 0;
        const int l = valid ? p - mul_sel<JP>(k, npsi) : 0;
        const double* row = fr + W.ring + (k - k0) * rowlen;
        const double mu = row[1], sig = row[3];
        const double c1 = P.cpsi[l], s1 = P.spsi[l];
        double ri, t0, t1;
        ring_eval<L, false>(row, LL, c1, s1, P.trig + (trig_lmajor(L) ? (size_t)P.trig_stride * l : (size_t)(2 * l)), P.trig_stride, ri, t0, t1);
        const double a1 = sig * c1, a2 = sig * s1;
        const double uj0 = fma(a1, fr[FR_BJ1], fma(a2, fr[FR_BJ2], mu * fr[FR_BJC]));
        const double uj1 = fma(a1, fr[FR_BJ1 + 1], fma(a2, fr[FR_BJ2 + 1], mu * fr[FR_BJC + 1]));
        const double uj2 = fma(a1, fr[FR_BJ1 + 2], fma(a2, fr[FR_BJ2 + 2], mu * fr[FR_BJC + 2]));
        const double q0 = fma(ri, uj0, -fr[FR_DJ]), q1 = fma(ri, uj1, -fr[FR_DJ + 1]),
                     q2 = fma(ri, uj2, -fr[FR_DJ + 2]);
        const double s2 = q0 * q0 + q1 * q1 + q2 * q2;
        const bool cand = valid && (s2 < fr[FR_RJ2]);
        const bool szero = !(s2 > 0.0);
        const double inv = rsqrt_nr1(fmax(s2, 1e-300));
        const double sN = s2 * inv;
        g0 = sN - fr[FR_RJ];  // outside B_j: the stand-in of SPEC §2.8 (>= 0)
        double rj0 = fr[FR_RJ];
        if (wave_any<false>(cand)) {  // wave-uniform
          const double rj0e = sh_eval<L>(rc, cwj, lrt, q0 * inv, q1 * inv, q2 * inv);
          if (!szero) rj0 = rj0e;
          if (cand) g0 = szero ? -rj0 : sN - rj0;
        }
        in0 = cand;
        ri0 = ri;
        rj00 = rj0;
      }
      if (t >= 1) {
        const int p1 = ((t - 1) << 6) + lane;
        const bool valid1 = p1 < Q;
        const int k1 = valid1 ? (int)(umul_sel<JP>((unsigned)p1, magic) >> 24) : 0;
        const int l1 = valid1 ? p1 - mul_sel<JP>(k1, npsi) : 0;
        double nb[3];
        if (aligned) {
          // rings do not straddle slabs (n_psi divides 64): the azimuth neighbours sit in slab t-1 itself and the
          // ring neighbour one ring further in slab t-1 or at the start of slab t (n_psi = 64: the same lane of
          // slab t, or of slab t-2 for the last ring) — 4 (2) cross-lane reads instead of 9
          const int base = lane & ~(npsi - 1);
          nb[0] = __shfl(wg1, base | ((lane + 1) & (npsi - 1)), 64);
          nb[1] = __shfl(wg1, base | ((lane - 1) & (npsi - 1)), 64);
          if (npsi == 64) {
            nb[2] = (k1 < nq - 1) ? g0 : wg2;
          } else {
            const int idx = lane + ((k1 < nq - 1) ? npsi : -npsi);
            const double v1 = __shfl(wg1, idx & 63, 64), v0 = __shfl(g0, idx & 63, 64);
            nb[2] = (idx < 64) ? v1 : v0;
          }
        } else {
          // neighbours as lane offsets within the three-slab window [t-2 | t-1 | t]
          const int o_lp = (l1 == npsi - 1) ? -(npsi - 1) : 1;
          const int o_lm = (l1 == 0) ? (npsi - 1) : -1;
          const int o_k = (k1 < nq - 1) ? npsi : -npsi;
#pragma unroll
          for (int a = 0; a < 3; ++a) {
            const int idx = lane + (a == 0 ? o_lp : (a == 1 ? o_lm : o_k));
            const int src = idx & 63;
            const double v2 = __shfl(wg2, src, 64), v1 = __shfl(wg1, src, 64), v0 = __shfl(g0, src, 64);
            nb[a] = (idx < 0) ? v2 : ((idx < 64) ? v1 : v0);
          }
        }
        const double Dl = 0.5 * fabs(nb[0] - nb[1]);
        const double Dk = (nq > 1) ? fabs(nb[2] - wg1) : 0.0;
        const double den = Dk + Dl;
        double wt = (wg1 < 0.0) ? 1.0 : 0.0;
        if (den > 0.0) wt = fmin(1.0, fmax(0.0, fma(-wg1, rcp_nr(den), 0.5)));
        const bool take = valid1 && win1 && (wt > 0.0);
        const unsigned long long m = wave_ballot(take);
        if (m != 0ULL) {
          if (take) {
            const int pos = (qhead + qcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                     __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) & (kQueue - 1);
            double* lq = (double*)fr;
            ((unsigned short*)(lq + W.qp))[pos] = (unsigned short)p1;
            lq[W.qri + pos] = wri1;
            lq[W.qrj + pos] = wrj1;
            lq[W.qw + pos] = (wg1 < 0.0) ? wt : -wt;  // the sign carries [g~ < 0] to phase 2 (no second opinion there)
          }
          qcount += __builtin_popcountll(m);
        }
      }
      wg2 = wg1;
      wg1 = g0;
      wri1 = ri0;
      wrj1 = rj00;
      win1 = in0;
    }
    } else if constexpr (JP) {
    // Queue append of the lanes flagged `in` (mask m_, prefix count).  A slab of node pairs may bring up to 128 inside
    // nodes to a queue that holds fewer than 64: with 128 ... 192 entries (queue_capacity) they do not always fit — a dense
    // slab of a deeply overlapping pair on top of a leftover.  Phase 2 then runs on the lanes' own nodes at once (DIRECT), a
    // second instance of the phase-2 lambda: every lane with an inside node keeps one of its two, the other — where both
    // are inside — is queued (at most 64 entries: they always fit), and the slab is consumed.  (SHP_DIRECT(L) = 0 is the
    // kernel before: the slab is NOT consumed, what is queued is drained as a (short) batch first and the slab is
    // classified again with the queue empty.  Until round 3 the second half waited in five registers that were live
    // through phase 2, which the kernel does not have.  Round 4, profiles/r04_ar_ab_direct.txt, r04_as_ab_direct2.txt:
    // headline -2.3 %, L = 7 / 16 -3.6 %, L = 8 / 20 -6.4 %, L = 6 / 32 -2.3 %, L = 2 / 16 -3.9 %, L = 5 / 24 -2.6 %; written
    // as ONE phase 2 with a second entry the same idea cost every kernel 2-14 registers and was dropped; so was filling
    // the queue with the first nodes of the slab and classifying it again for the rest — the number of batches per pair
    // does not change, r04_am_ab_queue2.txt.)
#define SHP_PUSH(m_, pn, rin_, rjn_)                                                                                  \
    {                                                                                                                  \
      if (m_ != 0ULL) {                                                                                                \
        if (lane_of(m_)) {                                                                                             \
          const int pos_ = qcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_ >> 32),                               \
                                                    __builtin_amdgcn_mbcnt_lo((unsigned)m_, 0u));                      \
          double* lq_ = SHP_LDS();                                                                                     \
          ((unsigned short*)(lq_ + W.qp))[pos_] = (unsigned short)(pn);                                                \
          lq_[W.qri + pos_] = (rin_);                                                                                  \
          lq_[W.qrj + pos_] = (rjn_);                                                                                  \
        }                                                                                                              \
        qcount += __builtin_popcountll(m_);                                                                            \
      }                                                                                                                \
    }
    while (qcount < 64 && slab < slab_end) {
      fr = SHP_LDS();
      const int pp = (slab << 6) + lane;   // node pair: ring k, azimuths l and l + n_q
      // idle lanes (past the last node pair: in the last slab only, whose group holds the last ring) take the last node
      // pair — a resident row, and one v_min instead of a compare, a masked region and two selects; mvalid drops them
      const int ppc = min(pp, nq * per_ring - 1);
      const int k = (int)(umul_sel<JP>((unsigned)ppc, magicr) >> 24);
      const int l = ppc - mul_sel<JP>(k, per_ring) + half * per_ring;
      const double* row = fr + W.ring + (k - k0) * rowlen;
      const v2d r01 = lds2(row);   // (A_k0, mu_k)
      const double mu = r01[1], sig = row[3];
      // r_i at the two azimuths: psi + pi changes the sign of the odd orders
      // cos/sin(m psi_l): the first order from the lane's row of particle j's table, the rest by angle addition
      const double* gr = fr + W.gh + mul_sel<JP>(l, jpoly_row(LJ));
      double re = r01[0], ro = 0.0;
      if constexpr (LJ >= 1) {
        const v2d cs1 = lds2(jpoly_trig_sep(LJ) ? fr + W.tr + 2 * l : gr + jpoly_trig(LJ));
        const double c1 = cs1[0], s1 = cs1[1];
        double cm = c1, sm = s1, cp = 1.0, sp = 0.0;   // three-term recurrence: one v_fma_f64 per cos / sin (ring_grad_rec)
        const double tc = c1 + c1;
#pragma unroll
        for (int m = 1; m <= LJ; ++m) {
          const v2d ab = lds2(row + 4 * m);
          const double A = ab[0], B = ab[1];
          if (m & 1) ro = fma(A, cm, fma(B, sm, ro));
          else re = fma(A, cm, fma(B, sm, re));
          if (m < LJ) {
            const double c = fma(tc, cm, -cp), s = fma(tc, sm, -sp);
            cp = cm;
            sp = sm;
            cm = c;
            sm = s;
          }
        }
      }
      const double ria = re + ro, rib = re - ro;
      const double rho = s_rho, rj2 = s_rj2;
      const double qa0 = fma(ria, mu, -rho), qa1 = ria * sig;
      const double qb0 = fma(rib, mu, -rho), qb1 = -rib * sig;
      const double sa2 = fma(qa0, qa0, qa1 * qa1), sb2 = fma(qb0, qb0, qb1 * qb1);
      // candidates, inside nodes: masks (scalar unit), not lane predicates
      // the valid lanes are the first (count - 64 slab) of the wave: the mask from scalar arithmetic (as a ballot of
      // `valid` it goes through a 0 / 1 value per lane)
      const int nvalid = nq * per_ring - (slab << 6);
      const unsigned long long mvalid = nvalid >= 64 ? ~0ULL : ((1ULL << nvalid) - 1ULL);
      const unsigned long long mca = wave_ballot(sa2 < rj2) & mvalid, mcb = wave_ballot(sb2 < rj2) & mvalid;
      if ((mca | mcb) == 0ULL) {   // wave-uniform: all 128 nodes miss B_j
#ifdef SHP_STATS   // a slab of this family is 128 nodes: counted as two, so that the counters compare across families
        if (lane == 0) atomicAdd(&P.dbg[0], 2ULL);
#endif
        ++slab;
        continue;
      }
      const bool za = !(sa2 > 0.0), zb = !(sb2 > 0.0);
      const unsigned long long mza = wave_ballot(za), mzb = wave_ballot(zb);   // once: the compares' own scalar pairs
      // no clamp of sa2, sb2 (two v_max_f64 each under IEEE mode): a node on x_j leaves NaN in r_j, replaced below
      const double inva = rsqrt_nr1(sa2), invb = rsqrt_nr1(sb2);
      double rjae, rjbe;   // one pass over the lane's row of particle j's table serves both nodes
      jpoly_eval2<LJ>(fr + W.gh + mul_sel<JP>(l, jpoly_row(LJ)), qa0 * inva, qa1 * inva, qb0 * invb, qb1 * invb, rjae, rjbe);
      const double Rjl = s_rj;
      double rja = rjae, rjb = rjbe;
      if (mask_any(mza | mzb)) {   // a node on x_j: measure zero; the volatile statement keeps this a branch
        asm volatile("; rare: a node on x_j");
        rja = za ? Rjl : rjae;
        rjb = zb ? Rjl : rjbe;
      }
      const unsigned long long ma = mca & (mza | wave_ballot(sa2 * inva < rja));
      const unsigned long long mb = mcb & (mzb | wave_ballot(sb2 * invb < rjb));
      const int pa = mul_sel<JP>(k, npsi) + l;
#ifdef SHP_STATS
      if (qcount + __builtin_popcountll(ma) + __builtin_popcountll(mb) > W.qcap) {
        if (lane == 0) atomicAdd(&P.dbg[9], 1ULL);
      }
#endif
      if (qcount + __builtin_popcountll(ma) + __builtin_popcountll(mb) > W.qcap) {   // wave-uniform; qcount > 0 here
        if constexpr (DIRECT) {
          // every lane with an inside node keeps one of its two — the second where both are inside, the first of those is
          // queued: at most 64 go to a queue that holds fewer than 64 — and phase 2 runs on the lanes' own nodes at once
          const unsigned long long mboth = ma & mb;
          SHP_PUSH(mboth, pa, ria, rja);
          const bool second = lane_of(mb);
          ++slab;
#ifdef SHP_STATS
          if (lane == 0) atomicAdd(&P.dbg[0], 2ULL);
          if (lane_of(mca)) atomicAdd(&P.dbg[1], 1ULL);
          if (lane_of(mcb)) atomicAdd(&P.dbg[1], 1ULL);
          if (lane == 0) atomicAdd(&P.dbg[2], 2ULL);
          if (lane_of(ma)) atomicAdd(&P.dbg[3], 1ULL);
          if (lane_of(mb)) atomicAdd(&P.dbg[3], 1ULL);
          if (lane == 0) atomicAdd(&P.dbg[10], 1ULL);
          if (lane == 0 && P.dbg[15]) atomicAdd(&P.dbg[64 + w], (unsigned long long)(__builtin_popcountll(ma) + __builtin_popcountll(mb)));   // per-slot inside-node counts (tools/halfwave_sim.py)
#endif
          phase2(BoolC<true>{}, second ? pa + nq : pa, second ? rib : ria, second ? rjb : rja, ma | mb);
          continue;
        }
        break;
      }
#ifdef SHP_STATS
      if (lane == 0) atomicAdd(&P.dbg[0], 2ULL);
      if (lane_of(mca)) atomicAdd(&P.dbg[1], 1ULL);
      if (lane_of(mcb)) atomicAdd(&P.dbg[1], 1ULL);
      if (lane == 0) atomicAdd(&P.dbg[2], 2ULL);
      if (lane_of(ma)) atomicAdd(&P.dbg[3], 1ULL);
      if (lane_of(mb)) atomicAdd(&P.dbg[3], 1ULL);
      if (lane == 0 && P.dbg[15]) atomicAdd(&P.dbg[64 + w], (unsigned long long)(__builtin_popcountll(ma) + __builtin_popcountll(mb)));
#endif
      ++slab;
      SHP_PUSH(ma, pa, ria, rja);
      SHP_PUSH(mb, pa + nq, rib, rjb);
    }
#undef SHP_PUSH
    } else {
    while (qcount < 64 && slab < slab_end) {
      fr = SHP_LDS();
      const int p = (slab << 6) + lane;
      ++slab;
      const bool valid = p < Q;
      const int k = valid ? (int)(umul_sel<JP>((unsigned)p, magic) >> 24) : 0;
      const int l = valid ? p - mul_sel<JP>(k, npsi) : 0;
      const double* row = fr + W.ring + (k - k0) * rowlen;
      const double mu = row[1], sig = row[3];
      const double c1 = P.cpsi[l], s1 = P.spsi[l];
      double ri, t0, t1;
      ring_eval<L, false>(row, LL, c1, s1, P.trig + (trig_lmajor(L) ? (size_t)P.trig_stride * l : (size_t)(2 * l)), P.trig_stride, ri, t0, t1);
      // the surface point seen from x_j, in j's body frame
      const double a1 = sig * c1, a2 = sig * s1;
      const double uj0 = fma(a1, fr[FR_BJ1], fma(a2, fr[FR_BJ2], mu * fr[FR_BJC]));
      const double uj1 = fma(a1, fr[FR_BJ1 + 1], fma(a2, fr[FR_BJ2 + 1], mu * fr[FR_BJC + 1]));
      const double uj2 = fma(a1, fr[FR_BJ1 + 2], fma(a2, fr[FR_BJ2 + 2], mu * fr[FR_BJC + 2]));
      const double q0 = fma(ri, uj0, -fr[FR_DJ]), q1 = fma(ri, uj1, -fr[FR_DJ + 1]),
                   q2 = fma(ri, uj2, -fr[FR_DJ + 2]);
      const double s2 = q0 * q0 + q1 * q1 + q2 * q2;
      const bool cand = valid && (s2 < fr[FR_RJ2]);
#ifdef SHP_STATS
      if (lane == 0) atomicAdd(&P.dbg[0], 1ULL);
      if (cand) atomicAdd(&P.dbg[1], 1ULL);
      { const bool a_ = __any(cand); if (lane == 0 && a_) atomicAdd(&P.dbg[2], 1ULL); }
#endif
      if (!wave_any<false>(cand)) continue;  // wave-uniform: the whole 64-node slab misses B_j

      // s == 0 (the node sits on x_j) is inside by definition; clamping s2 keeps that lane
      // finite without a select per component (its direction is then the zero vector)
      const bool szero = !(s2 > 0.0);
      const double inv = rsqrt_nr1(fmax(s2, 1e-300));
      const double rj0e = sh_eval<L>(rc, cwj, lrt, q0 * inv, q1 * inv, q2 * inv);
      const double rj0 = szero ? fr[FR_RJ] : rj0e;
      // SPEC: inside iff s < r_j (s == 0 is inside); s = s2 / sqrt(s2)
      const bool inside = cand && (szero || s2 * inv < rj0);
      const unsigned long long m = wave_ballot(inside);
#ifdef SHP_STATS
      if (inside) atomicAdd(&P.dbg[3], 1ULL);
#endif
      if (m == 0ULL) continue;
      if (inside) {
        const int pos = (qhead + qcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))) & (kQueue - 1);
        double* lq = (double*)fr;
        ((unsigned short*)(lq + W.qp))[pos] = (unsigned short)p;
        lq[W.qri + pos] = ri;
        lq[W.qrj + pos] = rj0;
      }
      qcount += __builtin_popcountll(m);
    }
    }
    if (qcount == 0) break;   // the group's slabs are classified and its queue is drained
#if defined(SHP_ABL) && SHP_ABL == 3   // timing-only build: phase 1 only, the queue is discarded
    qhead = (qhead + qcount) & (kQueue - 1);
    qcount = 0;
    continue;
#endif

    // ---------------------------------------------------------------- phase 2 (the lambda in front of the loop)
    phase2(BoolC<false>{}, 0, 0.0, 0.0, 0ULL);
  }
  }  // ring groups

  // ---------------------------------------------------------------- epilogue
  // Round 1: seven butterfly reductions (42 v_add_f64 + 84 ds_bpermute), then lane 0 alone rotated the sums to the
  // space frame, applied the force law and issued 12 scalar atomics — ~250 vector instructions per pair of which
  // ~150 ran with ONE active lane (an instruction costs its issue slots whatever the lane count): 7.4 % of the kernel
  // (ablation A=5).  Now: the 64 x 7 partial sums are transposed through LDS (the ring tables and the queue are
  // dead) and added in two levels of 8; the force law runs with one COMPONENT per lane — lanes 0-2 the force, 3-5
  // the torque — so the rotation is 3 FMAs instead of 18 and the scatter is ONE 6-lane global_atomic_add_f64 per
  // atom (f[3i..3i+2] and torque[3i..3i+2] are contiguous: 2 memory-side operations per atom instead of 6).
  __builtin_amdgcn_s_setprio(3);
  lane = fresh_lane();
  if constexpr (WPP == 2) __syncthreads();   // both waves are through their node loops: the shared tables are dead
  {
    // [7][kRedStride] partial sums | [56] | [7] totals | [6] force components; two waves per pair: one such block each
    double* red = SHP_LDS() + ((WPP == 2 && half) ? (int)(P.wave_lds_bytes >> 3) - kRedPerWave : FRAME);
    red[0 * kRedStride + lane] = aS0; red[1 * kRedStride + lane] = aS1; red[2 * kRedStride + lane] = aS2;
    red[3 * kRedStride + lane] = aT0; red[4 * kRedStride + lane] = aT1; red[5 * kRedStride + lane] = aT2;
    red[6 * kRedStride + lane] = NEEDV ? aV : 0.0;
    wave_lds_sync();
    double* part = red + 7 * kRedStride;
    if (lane < 56) {   // lane = (v, seg): the entries seg, seg + 8, ... of sum v
      const double* src = red + (lane >> 3) * kRedStride + (lane & 7);
      part[lane] = ((src[0] + src[8]) + (src[16] + src[24])) + ((src[32] + src[40]) + (src[48] + src[56]));
    }
    wave_lds_sync();
    double* tot = part + 56;
    if (lane < 7) {
      const double* src = part + 8 * lane;
      tot[lane] = ((src[0] + src[1]) + (src[2] + src[3])) + ((src[4] + src[5]) + (src[6] + src[7]));
    }
    wave_lds_sync();
    if constexpr (WPP == 2) {   // wave 0 adds the other half's totals and finishes the pair
      __syncthreads();
      if (half != 0) return;
      if (lane < 7) tot[lane] += SHP_LDS()[(int)(P.wave_lds_bytes >> 3) - kRedPerWave + 7 * kRedStride + 56 + lane];
      wave_lds_sync();
    }
  }
  if (lane >= 6) return;
  fr = SHP_LDS();
#undef SHP_LDS
  const double* tot = fr + FRAME + 7 * kRedStride + 56;
  double* fcomp = (double*)tot + 7;
  const int comp = (lane >= 3) ? lane - 3 : lane;   // 0..2
  const bool is_t = lane >= 3;                        // lanes 3-5: torque components
  // rotate the cap-frame integrals to the space frame (columns e1, e2, c): this lane's component of S_n or T_n
  const double a0 = tot[is_t ? 3 : 0], a1 = tot[is_t ? 4 : 1], a2 = tot[is_t ? 5 : 2];
  const double val = fr[FRM(FR_E1) + comp] * a0 + fr[FRM(FR_E2) + comp] * a1 + fr[FRM(FR_C) + comp] * a2;
  const double aVt = tot[6];

  LateParams& E = *late_params();   // the first explicit argument starts the kernarg segment
  if (E.pair_out) {
    double* o = E.pair_out + 7 * (size_t)w;
    o[1 + lane] = val;
    if (lane == 0) o[0] = aVt;
  }
  // touched: V > 0, or (forces only) any component of S_n non-zero
  const bool touched = NEEDV ? (aVt > 0.0) : (wave_ballot(!is_t && val != 0.0) != 0ULL);
  // statistics go through a byte per slot, summed by count_flags_kernel: one
  // atomic per pair on a shared counter costs more than the whole kernel
  if (E.flags && lane == 0) E.flags[w] = touched ? 2 : 1;
  if (!touched) return;

  // SPEC §2.7 force law
  // operands looked up (and the types range-checked) by the set-up kernel
  const int* ij = (const int*)(fr + FRM(FR_IJ));
  const int i = ij[0], j = ij[1];
  const double knij = fr[FRM(FR_KN)], mij = fr[FRM(FR_EXPO)];
  const double vm1 = (mij == 1.0) ? 1.0 : pow_quarter(aVt, mij - 1.0);  // V^(m-1)
  const double pn = knij * mij * vm1;
  const double Fm = -pn * val;   // lanes 0-2: F_i; lanes 3-5: tau_i
  fcomp[lane] = Fm;
  wave_lds_sync();
  double* const det = E.pair_ft;   // deterministic mode: the pair's numbers are written once, a gather adds them in list order
  if (det) det[12 * (size_t)w + lane] = Fm;
  else atomicAdd((is_t ? E.torque : E.f) + 3 * (size_t)i + comp, Fm);
  const bool applyj = E.newton_pair || j < E.nlocal;
  if (applyj) {
    // F_j = -F_i ;  tau_j = -tau_i - d x F_j : component c needs d and F_j at c + 1, c + 2
    const int c1 = (comp == 2) ? 0 : comp + 1, c2 = (comp == 0) ? 2 : comp - 1;
    double vj = -Fm;
    if (is_t) vj -= fr[FRM(FR_D) + c1] * (-fcomp[c2]) - fr[FRM(FR_D) + c2] * (-fcomp[c1]);
    if (det) det[12 * (size_t)w + 6 + lane] = vj;
    else atomicAdd((is_t ? E.torque : E.f) + 3 * (size_t)j + comp, vj);
  }
  // Global energy / virial tally (thermo steps): each of the six lanes owns ONE virial component and stores it, with
  // lane 0 adding the energy, into the pair's own 64-byte row of a per-slot buffer; tally_reduce_kernel (det_kernels.hpp)
  // adds the rows in slot order.  Round 3 had lane 0 issue 1 + 6 atomics on the same seven addresses for every touching
  // pair (~3.5 M same-address atomics per launch at the headline) and a sum whose last bits changed from run to run.
  if ((E.eflag || E.vflag) && E.pair_ev) {
    const double share = E.newton_pair ? 1.0 : (0.5 + (j < E.nlocal ? 0.5 : 0.0));
    double* row = E.pair_ev + 8 * (size_t)w;
    if (E.vflag) {
      // ev_tally_xyz with del = x_i - x_j = -d and the force on i: xx yy zz xy xz yz = d_a F_b, (a, b) per lane
      const int a = (lane < 3) ? lane : ((lane == 5) ? 1 : 0);
      const int b = (lane < 3) ? lane : ((lane == 3) ? 1 : 2);
      row[1 + lane] = share * (-fr[FRM(FR_D) + a]) * fcomp[b];
    }
    if (E.eflag && lane == 0) row[0] = share * knij * (vm1 * aVt);
  }
  if (lane != 0) return;
  // the per-atom tallies (only when asked for) stay with lane 0
  if (E.eatom || E.vatom) {
    const double F0 = fcomp[0], F1 = fcomp[1], F2 = fcomp[2];
    const double d0 = fr[FRM(FR_D)], d1 = fr[FRM(FR_D) + 1], d2 = fr[FRM(FR_D) + 2];
    // ev_tally_xyz per-atom part: half of the pair's energy / virial to each atom this rank tallies for
    const bool owni = E.newton_pair || i < E.nlocal;
    if (E.eatom) {
      const double eh = 0.5 * knij * (vm1 * aVt);
      if (owni) atomicAdd(&E.eatom[i], eh);
      if (applyj) atomicAdd(&E.eatom[j], eh);
    }
    if (E.vatom) {
      const double v[6] = {0.5 * (-d0) * F0, 0.5 * (-d1) * F1, 0.5 * (-d2) * F2,
                           0.5 * (-d0) * F1, 0.5 * (-d0) * F2, 0.5 * (-d1) * F2};
      for (int a = 0; a < 6; ++a) {
        if (owni) atomicAdd(&E.vatom[6 * (size_t)i + a], v[a]);
        if (applyj) atomicAdd(&E.vatom[6 * (size_t)j + a], v[a]);
      }
    }
  }
}

// Host-callable launcher, one per compiled order (pair_kernels_L*.hip).
typedef void (*pair_launch_fn)(const PairParams&, bool needv, hipStream_t, hipEvent_t wait_before_contact);
// Register / LDS footprint of the kernel that launch would pick (occupancy evidence for bench.py).
typedef hipError_t (*pair_attr_fn)(bool needv, bool weighted, hipFuncAttributes*, bool jpoly, bool split, bool spec);

// Orders for which the two-waves-per-pair kernels are compiled (they pay where one wave's private tables starve the CU
// of waves: large L with large n_q; the host's rule is use_split in shpair_api.hip)
__host__ __device__ constexpr bool split_compiled(int L) { return L >= 7; }

template <int L>
hipError_t pair_contact_attributes(bool needv, bool weighted, hipFuncAttributes* a, bool jpoly = false, bool split = false,
                                   bool spec = false)
{
  if constexpr (L >= 0) {
    if constexpr (PairSpec<L>::nq > 0) {
      if (spec && jpoly && !weighted && (split ? 2 : 1) == PairSpec<L>::wpp)
        return needv ? hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, true, false, true, PairSpec<L>::wpp, true>)
                     : hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, false, false, true, PairSpec<L>::wpp, true>);
    }
  }
  if constexpr (split_compiled(L)) {
    if (split && jpoly && !weighted)
      return needv ? hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, true, false, true, 2>)
                   : hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, false, false, true, 2>);
  }
  if (weighted) {
    if constexpr (L >= 0) return hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, true, true>);
    return hipErrorInvalidValue;
  }
  if constexpr (L >= 0) {
    if (jpoly)
      return needv ? hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, true, false, true>)
                   : hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, false, false, true>);
  }
  return needv ? hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, true>)
               : hipFuncGetAttributes(a, (const void*)pair_contact_kernel<L, false>);
}

template <typename K>
static inline void launch_contact_one(K kern, const dim3 grid, const dim3 block, const size_t lds, hipStream_t st,
                                      const PairParams& P)
{
  if (lds > 65536) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, grid, block, lds, st, P);
}

// wait_before_contact (nullable): an event the CONTACT kernel waits for, not the rotation kernel in front of it — the
// host-pointer entry point uploads f and torque on a second stream beside the set-up and rotation kernels.
template <int L>
void launch_pair_contact(const PairParams& P, bool needv, hipStream_t st, hipEvent_t wait_before_contact = nullptr)
{
  const int nslots = P.npairs - P.slot0;   // the launch covers the slots [slot0, npairs)
  if (nslots <= 0) return;
  const int wpb = P.waves_per_block;
  const dim3 grid((nslots + wpb - 1) / wpb), block(64 * wpb);
  const size_t lds = (size_t)wpb * P.wave_lds_bytes;
  if (P.rule) {
    // SPEC §2.8; one instantiation (with the volume path) serves both force laws
    if (wait_before_contact) (void)hipStreamWaitEvent(st, wait_before_contact, 0);
    if constexpr (L >= 0) launch_contact_one(pair_contact_kernel<L, true, true>, grid, block, lds, st, P);
    return;
  }
  if constexpr (L >= 0) {
    if (P.jpoly) {
      // both particles' coefficient rotations, one lane each, then the contact kernel that reads them
      hipLaunchKernelGGL((pair_rotate_lane_kernel<L>), dim3((2 * (unsigned)nslots + 63) / 64), dim3(64),
                         RotLaneLds<L>::bytes(), st, P, const_cast<double*>(P.rot));
      if (wait_before_contact) (void)hipStreamWaitEvent(st, wait_before_contact, 0);
      if constexpr (PairSpec<L>::nq > 0) {
        if (pair_spec_matches<L>(P)) {   // the order's BASELINE shape: n_q, ring rows and queue capacity are compile-time constants
          constexpr int SW = PairSpec<L>::wpp;
          const dim3 gs(SW == 2 ? nslots : (int)grid.x), bs(64 * SW);
          const size_t ls = SW == 2 ? (size_t)P.wave_lds_bytes : lds;
          if (needv) launch_contact_one(pair_contact_kernel<L, true, false, true, SW, true>, gs, bs, ls, st, P);
          else launch_contact_one(pair_contact_kernel<L, false, false, true, SW, true>, gs, bs, ls, st, P);
          return;
        }
      }
      if constexpr (split_compiled(L)) {
        if (P.split) {   // two waves per pair: the workgroup is the pair
          const dim3 grid2(nslots), block2(128);
          if (needv) launch_contact_one(pair_contact_kernel<L, true, false, true, 2>, grid2, block2, (size_t)P.wave_lds_bytes, st, P);
          else launch_contact_one(pair_contact_kernel<L, false, false, true, 2>, grid2, block2, (size_t)P.wave_lds_bytes, st, P);
          return;
        }
      }
      if (needv) launch_contact_one(pair_contact_kernel<L, true, false, true>, grid, block, lds, st, P);
      else launch_contact_one(pair_contact_kernel<L, false, false, true>, grid, block, lds, st, P);
      return;
    }
  }
  if (wait_before_contact) (void)hipStreamWaitEvent(st, wait_before_contact, 0);
  if (needv) launch_contact_one(pair_contact_kernel<L, true>, grid, block, lds, st, P);
  else launch_contact_one(pair_contact_kernel<L, false>, grid, block, lds, st, P);
}

}  // namespace shp
