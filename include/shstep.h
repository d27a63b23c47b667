/*
 * shstep.h — C ABI of the steps either side of the `pair_style sh` contact path (SURVEY.md §8f rows 2
 * and 4): the rigid-body integrator of SH particles, gravity/viscous body forces, and — for a host
 * that keeps its atoms resident in HBM — periodic ghosts and the binned half neighbour list built on
 * the device.  Same library (libshpair.so), same context and conventions as include/shpair.h.
 *
 * Reference citations: the fork's `fix nve/sh`-style integrator and its neighbour code are ABSENT
 * FROM MOUNT (/root/reference/README.md:1 is the whole reference), so each entry point
 * names the stock LAMMPS interface it serves (Fix::initial_integrate / final_integrate / post_force,
 * Comm::borders / forward_comm / reverse_comm, Neighbor::build / check_distance) instead of a
 * file:line.  The algorithm is docs/SPEC.md Part II.  LAMMPS-side adapter of the integrator:
 * lammps-spherharm_amd/lammps/fix_nve_sh.{h,cpp}.
 *
 * All *_device functions take device pointers, enqueue on `stream` (NULL = HIP's null stream) and
 * return without waiting unless stated otherwise.  There is no CPU fallback.
 */
#ifndef SHSTEP_H
#define SHSTEP_H

#include "shpair.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- rigid-body properties (SPEC §5) -------------------------------------- */

/* Stateless host helper: out[10] = V, c[3], J_c (xx,yy,zz,xy,xz,yz) of a shape at unit density,
 * body frame (what the reference's atom style would hold per shape). */
int shstep_shape_mass_props(int lmax, const double *anm, double *out);

/* Density of shape `ishape` (default 1).  After shpair_set_shape(). */
int shstep_set_density(shpair_ctx *ctx, int ishape, double rho);

/* mass, centre of mass (body frame) and inertia about it (xx,yy,zz,xy,xz,yz) of a shape; any output
 * may be NULL.  What a LAMMPS fix needs for rmass / `compute erotate`. */
int shstep_get_body(const shpair_ctx *ctx, int ishape, double *mass, double *com, double *inertia);

/* ---- Fix::initial_integrate / final_integrate (SPEC §6) ------------------- */

/* phase 0: initial_integrate (half kick, drift, quaternion update); phase 1: final_integrate (half
 * kick).  Particles with (mask[i] & groupbit) == 0 are untouched.  x, v, angmom: [n][3]; quat: [n][4]
 * (w,x,y,z); v is the velocity of the centre of mass, angmom is about it, space frame. */
int shstep_nve_device(shpair_ctx *ctx, int phase, int nlocal, double dt, double *x_dev, double *v_dev,
                      double *quat_dev, double *angmom_dev, const double *f_dev, const double *torque_dev,
                      const int *shtype_dev, const int *mask_dev, int groupbit, void *stream);

/* Host-pointer form for a CPU-resident LAMMPS (stages through the device, blocks). */
int shstep_nve(shpair_ctx *ctx, int phase, int nlocal, double dt, double *x, double *v, double *quat,
               double *angmom, const double *f, const double *torque, const int *shtype, const int *mask,
               int groupbit);

/* Verlet::force_clear: zeroes f and torque of nall atoms (owned + ghost) in one launch on `stream`. */
int shstep_force_clear_device(shpair_ctx *ctx, int nall, double *f_dev, double *torque_dev, void *stream);

/* Fix::post_force of `fix gravity` + `fix viscous` in one pass: f += m g - gamma_t v,
 * torque += s x (m g - gamma_t v) - gamma_r omega  (s = R c: both act at the centre of mass). */
int shstep_post_force_device(shpair_ctx *ctx, int nlocal, const double *gravity3, double gamma_t, double gamma_r,
                             const double *v_dev, const double *quat_dev, const double *angmom_dev,
                             const int *shtype_dev, const int *mask_dev, int groupbit, double *f_dev,
                             double *torque_dev, void *stream);

/* `compute ke` / `compute erotate` / gravitational potential: ADDS into out3_dev[0..2] (device):
 * sum 1/2 m v^2, sum 1/2 omega.L, sum -m g.(x + s). */
int shstep_energies_device(shpair_ctx *ctx, int nlocal, const double *gravity3, const double *x_dev,
                           const double *v_dev, const double *quat_dev, const double *angmom_dev,
                           const int *shtype_dev, const int *mask_dev, int groupbit, double *out3_dev,
                           void *stream);

/* ---- Domain / Comm / Neighbor for a device-resident host (SPEC §7) --------- */

/* Orthogonal box, per-dimension periodic flags, neighbour skin. */
int shstep_set_box(shpair_ctx *ctx, const double *lo3, const double *hi3, const int *periodic3, double skin);

/* Domain::pbc + Comm::borders: wraps the owned rows of x into the box and appends the periodic images
 * as ghost rows nlocal .. nlocal+nghost-1 of x, quat, type, shtype and tag (arrays sized for nmax
 * rows; tag may be NULL = tag is the row index).  Blocks (nghost is read back).  Fails with
 * SHPAIR_ENOMEM if nlocal + nghost > nmax (nghost is still returned). */
int shstep_borders_device(shpair_ctx *ctx, int nlocal, int nmax, double *x_dev, double *quat_dev, int *type_dev,
                          int *shtype_dev, int *tag_dev, int *nghost, void *stream);

/* Comm::forward_comm: ghost x = owner x + shift, ghost quat = owner quat. */
int shstep_forward_device(shpair_ctx *ctx, double *x_dev, double *quat_dev, void *stream);
/* Comm::reverse_comm: owner f, torque += ghost f, torque. */
int shstep_reverse_device(shpair_ctx *ctx, double *f_dev, double *torque_dev, void *stream);

/* Neighbor::build: bins owned + ghost particles and builds the SPEC §7 half list on the device, installs
 * it as the context's neighbour list (as shpair_set_neighbors_device would) and records x for the
 * rebuild test.  tag may be NULL (row index; ghosts then use their owner's index).  Blocks (npairs is
 * read back). */
int shstep_neighbor_build_device(shpair_ctx *ctx, int nlocal, int nghost, const double *x_dev, const int *shtype_dev,
                                 const int *tag_dev, int *npairs, void *stream);

/* Neighbor::check_distance: *rebuild = 1 if an owned particle moved more than skin/2 since the last
 * build.  Blocks (one flag is read back). */
int shstep_neighbor_check_device(shpair_ctx *ctx, int nlocal, const double *x_dev, int *rebuild, void *stream);

/* Copies the current device-built list to the host in CSR form: offsets[nlocal+1], jlist[npairs]
 * (either may be NULL).  Blocks. */
int shstep_copy_neighbors(shpair_ctx *ctx, int *offsets, int *jlist);

/* ---- the whole loop, for a host that owns nothing but the arrays ----------- */

/* Device pointers and scalars of one rank's particles; arrays sized for nmax rows (owned + ghosts) except
 * v, angmom, mask (nlocal rows). */
typedef struct shstep_arrays {
  int nlocal, nmax;
  double *x, *v, *quat, *angmom, *f, *torque;
  int *type, *shtype, *mask;
  int groupbit;
  double dt;
  double gravity[3], gamma_t, gamma_r; /* all zero: no post_force pass */
  int check_every;                      /* rebuild test every this many steps (>= 1), neigh_modify every N check yes */
} shstep_arrays;

/* Verlet::run for nsteps: initial_integrate -> [rebuild test -> borders + neighbour build] -> forward ->
 * clear -> pair compute -> reverse -> post_force -> final_integrate, entirely on `stream` (must not be
 * NULL when use_graph is set: the legacy null stream cannot be captured).  On entry the ghosts / list
 * of the current positions must exist (shstep_borders_device + shstep_neighbor_build_device) and f, torque
 * must hold their forces (as after Verlet::setup); *nghost is the current ghost count and is updated.
 * use_graph != 0: the launches of a step are replayed from two captured hipGraphs (re-captured after every
 * rebuild) instead of being issued one by one — the launch-bound regime of small systems.
 * Returns after the last step is enqueued and the stream is idle (blocks).  *rebuilds (nullable) counts
 * the list rebuilds. */
int shstep_run_device(shpair_ctx *ctx, const shstep_arrays *a, int nsteps, int use_graph, int *nghost, int *rebuilds,
                      void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHSTEP_H */
