/*
 * shpair.h — C ABI of the MI355X-native `pair_style sh` contact path.
 *
 * Drop-in boundary for the SPHERHARM pair style (BASELINE.json north_star:
 * "keeping LAMMPS's pair_style / compute() plugin API ... calling hand-written
 * HIP kernels through a thin C-ABI layer").
 *
 * Reference citations: the reference mount holds only /root/reference/README.md:1
 * ("SPHERHARM Package to simulate complex shaped granular particles"); the
 * PairSH sources these entry points stand in for are ABSENT FROM MOUNT
 * (SURVEY.md §0, §8b), so each entry point names the LAMMPS `Pair` virtual it
 * serves instead of a file:line.  The LAMMPS-side adapter that binds them is
 * lammps-spherharm_amd/lammps/pair_sh.{h,cpp}; see INTEGRATION.md.
 *
 * Conventions: plain pointers and sizes, no C++/torch types, no exceptions.
 * Every function returns 0 (SHPAIR_OK) or a negative SHPAIR_E* code;
 * shpair_strerror() names it and shpair_last_error() gives the detail string
 * of the last failure on a context.  One context per rank/GPU; a context is
 * not re-entrant.  There is NO CPU fallback: without a usable HIP device
 * shpair_create() fails with SHPAIR_ENODEV.
 */
#ifndef SHPAIR_H
#define SHPAIR_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SHPAIR_OK 0
#define SHPAIR_EINVAL -1   /* bad argument                              */
#define SHPAIR_ENODEV -2   /* no HIP device / device id out of range    */
#define SHPAIR_EHIP -3     /* a HIP runtime call failed                 */
#define SHPAIR_ESTATE -4   /* call order: shapes/coeffs/neighbours unset */
#define SHPAIR_ENOMEM -5
#define SHPAIR_ELMAX -6    /* lmax or nq above the compiled limits      */

#define SHPAIR_MAX_LMAX 20 /* L <= 12 run unrolled kernels, 13..20 the loop kernel */
#define SHPAIR_MAX_NQ 128
#define SHPAIR_NEIGHMASK 0x1FFFFFFF /* LAMMPS NEIGHMASK: low 29 bits of a neighbour entry */

typedef struct shpair_ctx shpair_ctx;

/* Per-call statistics of the last compute (shpair_get_stats). */
typedef struct shpair_stats {
  long long n_candidates;   /* pairs in the half list                         */
  long long n_contact;      /* pairs whose bounding spheres overlap (metric)  */
  long long n_touching;     /* pairs with a non-zero force this call          */
  double kernel_ms;         /* device time of the pair kernels (hipEvents)    */
  double total_ms;          /* device time incl. staging copies, 0 if unknown */
} shpair_stats;

/* ---- lifetime ------------------------------------------------------------ */

/* PairSH::PairSH(LAMMPS*) — allocate a context bound to HIP device `device_id`
 * (one non-blocking stream of its own). */
int shpair_create(shpair_ctx **out, int device_id);
/* PairSH::~PairSH() */
void shpair_destroy(shpair_ctx *ctx);
const char *shpair_strerror(int code);
const char *shpair_last_error(const shpair_ctx *ctx);
/* Library identification: "shpair <version> gfx950". */
const char *shpair_version(void);

/* ---- setup: PairSH::settings() / coeff() / init_style() / init_one() ------ */

/* PairSH::settings(): `pair_style sh <nq>` — Gauss order of the cap
 * quadrature (Q = 2 nq^2 nodes per pair), 1 <= nq <= SHPAIR_MAX_NQ. */
int shpair_settings(shpair_ctx *ctx, int nq);

/* Sizes the tables: `ntypes` LAMMPS atom types (1-based in pair_coeff) and
 * `nshapes` shape-table entries (0-based `shtype`).  Clears earlier shapes
 * and coefficients. */
int shpair_set_ntypes(shpair_ctx *ctx, int ntypes, int nshapes);

/* One entry of the per-shape table the reference's atom style owns
 * (coefficients a_nm, m >= 0, n-major: anm[2k]=Re, anm[2k+1]=Im,
 * k = n(n+1)/2+m; (lmax+1)(lmax+2) doubles; docs/SPEC.md §1).
 * rmax > 0 sets the bounding radius, rmax <= 0 asks for the default
 * (1.01 x grid maximum).  Caller keeps ownership of anm. */
int shpair_set_shape(shpair_ctx *ctx, int ishape, int lmax, const double *anm, double rmax);

/* PairSH::coeff(): `pair_coeff I J kn exponent` for ONE (itype, jtype)
 * (1-based; the adapter expands wildcards and mirrors i<->j). */
int shpair_set_coeff(shpair_ctx *ctx, int itype, int jtype, double kn, double exponent);

/* PairSH::init_one(i,j) support: bounding radius of a shape. */
int shpair_get_rmax(const shpair_ctx *ctx, int ishape, double *rmax);

/* Stateless host helpers (no device needed): radius of a shape in body-frame
 * unit direction u[3], and the default bounding radius. */
int shpair_shape_radius(int lmax, const double *anm, const double *u, double *r);
int shpair_shape_default_rmax(int lmax, const double *anm, double *rmax);

/* ---- neighbour list: consumed after Neighbor::build() --------------------- */

/* LAMMPS NeighList layout (half list): inum, ilist[inum], numneigh[] and
 * firstneigh[] indexed by ATOM index i = ilist[ii].  Entries are masked with
 * SHPAIR_NEIGHMASK.  Host pointers; copied. */
int shpair_set_neighbors(shpair_ctx *ctx, int inum, const int *ilist, const int *numneigh,
                         const int *const *firstneigh);
/* Same list in CSR form: offsets[inum+1] into jlist. Host pointers; copied. */
int shpair_set_neighbors_csr(shpair_ctx *ctx, int inum, const int *ilist, const int *offsets,
                             const int *jlist);

/* The CSR list already resident in HBM (a device-resident host such as a KOKKOS build): device
 * pointers, expanded on the device, asynchronous on `stream` (NULL = HIP null stream).  npairs =
 * offsets[inum] must be passed by the caller (it is not read back); max_atom_index = the largest
 * atom index the list can contain (normally nlocal + nghost - 1), used for the stale-list check. */
int shpair_set_neighbors_device(shpair_ctx *ctx, int inum, const int *ilist_dev, const int *offsets_dev,
                                const int *jlist_dev, int npairs, int max_atom_index, void *stream);

/* ---- PairSH::compute(eflag, vflag) --------------------------------------- */

/* Host-pointer form, LAMMPS layout: x[nall][3], quat[nall][4] (w,x,y,z),
 * type[nall] (1-based), shtype[nall] (0-based), nall = nlocal + nghost.
 * ADDS into f[nall][3] and torque[nall][3] (LAMMPS' force_clear() zeroes them).
 * eng_vdwl / virial[6] (xx,yy,zz,xy,xz,yz) are ADDED to when eflag / vflag
 * are non-zero (may be NULL otherwise). Blocks until the result is on the host. */
int shpair_compute(shpair_ctx *ctx, int nlocal, int nghost, const double *x, const double *quat,
                   const int *type, const int *shtype, int newton_pair, int eflag, int vflag,
                   double *f, double *torque, double *eng_vdwl, double *virial);

/* Optional, for hosts that call shpair_compute() every step with the SAME arrays (LAMMPS: atom->x, f, torque and the
 * per-atom vectors stay where they are until atom->nmax grows): page-lock a caller-owned array (hipHostRegister), so
 * that the per-call copies are direct DMA at PCIe rate (16 MB per call at 100k atoms: ~0.25 ms) instead of the
 * runtime's staged copy of pageable memory, and the upload of f / torque overlaps the set-up kernels.  CONTRACT: the
 * range [ptr, ptr + bytes) stays allocated until shpair_unpin_host(ptr), a shpair_pin_host() of the same ptr with another
 * length, or shpair_destroy(); when the host reallocates an array (nmax changed) it unpins the old pointer — also if the
 * memory is already gone — and pins the new one.  Pinning is never required: unpinned arrays take the staged path. */
int shpair_pin_host(shpair_ctx *ctx, void *ptr, size_t bytes);
int shpair_unpin_host(shpair_ctx *ctx, void *ptr);

/* Device-pointer form: all arrays already resident in HBM (the measured
 * path). Same layout and ADD semantics; ev_dev (nullable) is 7 doubles on the
 * device: [0] += energy, [1..6] += virial.  stream: the hipStream_t to launch
 * on; NULL is HIP's null stream (what the arrays' producer normally used), and
 * shpair_get_stream() returns the context's own stream if that is wanted.
 * Asynchronous: returns after enqueueing. */
int shpair_compute_device(shpair_ctx *ctx, int nlocal, int nghost, const double *x_dev,
                          const double *quat_dev, const int *type_dev, const int *shtype_dev,
                          int newton_pair, int eflag, int vflag, double *f_dev, double *torque_dev,
                          double *ev_dev, void *stream);

/* Options. key = "force_volume" (1: always run the overlap-volume root finder,
 * even when exponent == 1 and eflag == 0), "timing" (1: bracket the pair
 * kernels with hipEvents for shpair_get_stats), "count" (1: count contact
 * pairs every call), "variant" (1: force the run-time-order loop kernel), "rule" (0: sharp inside
 * test, the default; 1: the covered-fraction weights of docs/SPEC.md §2.8 — `pair_style sh <nq> rule weighted`;
 * needs lmax <= 12 and nq <= 32), "ring_rows" (> 0:
 * override the number of quadrature rings whose tables are LDS resident at a time; tuning), "jpoly" (which kernel
 * family evaluates the neighbour's radius for the compiled orders lmax <= 12, sharp rule: 1 = per-azimuth
 * polynomials in the pair's common frame, 0 = body-frame Horner evaluation, -1 (default) = whichever the library's
 * measured rule picks for (lmax, nq); same results to rounding, ~1e-14 relative), "split" (1: two waves per pair — the
 * pair's tables are shared by a 128-lane workgroup, each wave integrates half of the azimuths — for the "jpoly" family
 * at lmax >= 7 and even nq; 0: one wave per pair; -1 (default): two where one wave's private tables would leave a CU
 * fewer than 16 waves), "deterministic" (1: bitwise reproducible forces and torques — each pair's force and torque
 * are written once into a per-slot buffer and added per atom in list order by a gather pass through a reverse index
 * that is rebuilt on the device whenever a list is installed, instead of hardware FP64 atomics whose order of
 * arrival varies from run to run (last-bit differences, ~1e-16 relative per add); costs one more pass and 96 bytes per
 * list slot; the ghost reverse sums of shstep / shhalo follow the option (no atomics, fixed order), so device-resident
 * trajectories are reproducible too, on one rank and on several; 0 (default): atomics.  The GLOBAL energy / virial
 * tallies are bitwise reproducible in both modes — per-slot rows added in slot order — the per-atom tallies keep their
 * atomics), "halo_overlap" (1: device-built lists are partitioned — slots whose two atoms are owned first, slots with a
 * ghost behind them, each in list order — and shhalo_run_device runs the forward exchange of a step on a stream of its
 * own beside the pair kernels of the owned-only slots; 2: the reverse exchange is hidden as well — half of the owned-only
 * slots run beside the forward exchange, the ghost slots follow it, the other half runs beside the reverse exchange,
 * whose unpack uses the same FP64 atomics (atomic accumulation only: with "deterministic" 2 behaves as 1); 0, the default: the
 * exchanges and the pair kernels follow each other on the caller's stream; same forces, another order of the per-atom
 * sums), "spec" (default 1: a launch whose order, n_q, ring rows and queue capacity are those of a specialised instance — the
 * BASELINE shapes L = 4 / n_q = 10, L = 6 / n_q = 16, L = 12 / n_q = 32 — runs that instance, in which the three are
 * compile-time constants: same arithmetic, bitwise-equal results, fewer index instructions; 0: always the general kernels),
 * "halo_stream_priority" (1: that second stream is one at the highest stream priority — a hardware queue of its
 * own whatever other streams the process has, its few workgroups dispatched ahead of the pair kernels' backlog; 0, the
 * default: an ordinary stream; takes effect at the next shhalo_run_device),
 * "waves_per_block" (tuning: waves per workgroup of the one-wave contact kernels, default 1), "queue_slack"
 * (diagnostic, default 1: the node queue of the "jpoly" kernels takes what the wave's LDS layout leaves of its last
 * 1 280-byte allocation granule, up to 192 entries; 0: 128 entries).
 * Memory: the contact path keeps per-slot scratch in HBM — a 320-byte record and, for the "jpoly" family, two rotated
 * coefficient vectors of (lmax+1)^2 doubles each (rounded up to 8 from lmax = 9 on): 1.1 KB per list slot at lmax = 6
 * (0.7 GB at 100k particles / 580k pairs, ~7 GB at 1 M), 3.0 KB at lmax = 12 (1.7 GB at 100k); +96 bytes per slot in
 * the deterministic mode; +64 bytes per slot for the tally rows of thermo steps. */
int shpair_set_option(shpair_ctx *ctx, const char *key, int value);

/* Static footprint of the pair kernel the last compute launched (occupancy evidence): registers per lane, LDS
 * per wave (one wave = one pair = one workgroup), and the resident waves those allow on a gfx950 CU (4 SIMDs,
 * 512 VGPRs per lane and SIMD, 160 KiB LDS). */
typedef struct shpair_kernel_info {
  int lmax, compiled_order;   /* compiled_order 0: the run-time-order loop kernel */
  int vgprs, scratch_bytes;
  int lds_bytes_per_wave, ring_rows;
  int waves_per_simd_vgpr;    /* limit from registers */
  int waves_per_cu_lds;       /* limit from LDS (allocated in granules of 1 280 bytes) */
  int waves_per_cu;           /* min(4 x waves_per_simd_vgpr, waves_per_cu_lds) */
  int family;                 /* 0: particle j evaluated in its body frame (Horner, scalar-fed coefficients);
                                 1: from per-azimuth polynomials in the pair's common frame (option "jpoly") */
  int waves_per_pair;         /* 1: one wave = one pair = one workgroup; 2: two waves share a pair's tables (option
                                 "split"); lds_bytes_per_wave is then the pair's LDS / 2 */
  int needv, weighted;        /* the instance's other two template arguments: overlap-volume root finder compiled in;
                                 covered-fraction rule.  (lmax, needv, weighted, family, waves_per_pair) name the
                                 pair_contact_kernel instance that ran — profiles/pmc_traffic.json is keyed to its code */
  int queue_entries;          /* entries of a wave's node queue: 128, or for the "jpoly" family 128 + what the layout leaves
                                 of its last LDS granule (at most 192; option "queue_slack") */
  int specialised;            /* 1: the instance with n_q, ring rows and queue capacity as compile-time constants ran (the order's
                                 BASELINE shape: L = 4 / n_q = 10, L = 6 / n_q = 16, L = 12 / n_q = 32; option "spec") */
} shpair_kernel_info;
int shpair_get_kernel_info(shpair_ctx *ctx, shpair_kernel_info *out);

/* Blocks until the last compute finished, then fills `out`. */
int shpair_get_stats(shpair_ctx *ctx, shpair_stats *out);

/* Optional per-pair diagnostics of the next computes: pair_out_dev holds
 * 7 doubles per half-list entry (V, S_n[3], T_n[3], docs/SPEC.md §2), device
 * memory owned by the caller; NULL disables. */
int shpair_set_pair_output(shpair_ctx *ctx, double *pair_out_dev);

/* Per-atom tallies (Pair::eatom / vatom, filled as LAMMPS' ev_tally_xyz does: half of a pair's energy
 * and virial to each of its atoms this rank tallies for).  Device form: eatom_dev[nall] and
 * vatom_dev[nall][6] (xx,yy,zz,xy,xz,yz) are ADDED to by every following shpair_compute_device(); NULL
 * switches either off.  A non-NULL eatom makes the overlap-volume root finder run. */
int shpair_set_peratom_output(shpair_ctx *ctx, double *eatom_dev, double *vatom_dev);
/* Host form for shpair_compute(): host arrays of the same shapes, staged through the device. */
int shpair_set_peratom_host(shpair_ctx *ctx, double *eatom, double *vatom);

/* The box's FP64 ceilings, measured instead of trusted (SURVEY.md §8d asks for a v_fma_f64 microbenchmark):
 * mode 0 = independent v_fma_f64 chains on every SIMD (the vector peak the pair kernel is priced against),
 * mode 1 = v_mfma_f64_16x16x4_f64 alone, mode 2 = both side by side on every SIMD (4 waves each).
 * Runs on the context's device for about target_ms per launch (best of 5) and blocks.  TFLOP/s of the waves
 * running each loop; either output may be NULL. */
int shpair_fp64_peak(shpair_ctx *ctx, int mode, double target_ms, double *valu_tflops, double *mfma_tflops);

/* The context's own non-blocking stream (a hipStream_t), used by
 * shpair_compute() for its staging copies and kernels. */
int shpair_get_stream(shpair_ctx *ctx, void **stream);

/* Wait for everything enqueued on the context's stream. */
int shpair_synchronize(shpair_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* SHPAIR_H */
