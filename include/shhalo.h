/*
 * shhalo.h — C ABI of the N > 1 path of `pair_style sh` (BASELINE.json configs[3]): LAMMPS'
 * spatial domain decomposition around PairSH::compute — Comm::exchange (atom migration), Comm::borders (ghost
 * selection), Comm::forward_comm (x + quaternion of ghosts) and Comm::reverse_comm (force + torque of ghosts
 * summed into their owners) — for a host whose atoms are resident in HBM, one rank and one shpair context per
 * GPU, on RCCL point-to-point over xGMI.  Same library (libshpair.so) and conventions as include/shpair.h
 * (0 or a negative SHPAIR_E* code; shhalo_last_error() gives the detail).
 *
 * Reference citations: the fork's Comm code is stock LAMMPS and ABSENT FROM MOUNT (/root/reference/README.md:1 is
 * the whole reference), so each entry point names the LAMMPS interface it serves.
 *
 * MI355X-first shape: bricks of a px x py x pz grid; every rank talks to each of its <= 26
 * geometric neighbours (7 distinct peers on a 2x2x2 node, one per xGMI link) DIRECTLY: per step and direction of
 * travel ONE ncclGroupStart .. ncclSend/ncclRecv per peer .. ncclGroupEnd on the caller's stream, no host wait,
 * one pack and one unpack kernel whatever the number of peers — instead of LAMMPS' three staged x/y/z swaps.
 * The plan (who sends which rows to whom, and where they land) is rebuilt on the device at every reneighbouring;
 * the same plan logic is exported as pure host functions (shhalo_plan_*) so that it can be tested without a GPU.
 *
 * Errors across ranks: exchange / borders / run are collective calls.  Rank-local failures that are found between
 * two exchanges — a lost atom, nmax too small for the arrivals or the ghosts, a type or shape index outside its
 * table — are AGREED ON with one max-all-reduce of an error word before any further message is posted, so that every
 * rank returns a non-zero code from the same call (the failing rank its own code and message, the others
 * SHPAIR_ESTATE naming it) and none is left waiting inside ncclRecv, which has no timeout.  What is covered: the
 * failures of a reneighbouring (shhalo_exchange_device, shhalo_borders_device, the list build — inside
 * shhalo_run_device too) and the pair / step kernels' error bits, which shhalo_run_device reads once at the end of a
 * call.  NOT covered: argument and state errors of the per-step calls inside shhalo_run_device's loop (null arrays, a
 * list that does not match the atoms) and of shhalo_forward_device / shhalo_reverse_device called directly — the
 * contract is the same call sequence with like arguments on every rank, so these are raised by every rank in the same
 * step or are a caller's bug; an all-reduce per step to agree on them would cost every step ~1 % for nothing.
 * Any non-zero return of any rank is fatal for the communicator: destroy the contexts (a HIP or RCCL call that failed in the middle of a
 * step — SHPAIR_EHIP — cannot be agreed on, the peers are already inside their exchanges; treat it like a lost rank).
 *
 * Transports: RCCL (the product; librccl is bound at run time with dlopen, so a single-GPU host does not need
 * it), and an in-process hub that moves the same messages between the contexts of several host THREADS of one
 * process with device copies — for rehearsing N ranks on fewer than N GPUs (tests) and for self-periodic
 * single-rank runs.  Memory: every array handed in must be ordinary device memory (hipMalloc): the force
 * accumulation uses hardware FP64 atomics, which fine-grained / host-coherent allocations do not support.
 */
#ifndef SHHALO_H
#define SHHALO_H

#include <stddef.h>
#include "shpair.h"
#include "shstep.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SHHALO_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */

typedef struct shhalo_ctx shhalo_ctx;
typedef struct shhalo_hub shhalo_hub;

/* ---- the plan, as pure host functions (no device, no transport) ------------------------------------------ */

/* One rank's view of the brick decomposition.  Direction code of a step s = (sx, sy, sz) in {-1,0,1}^3:
 * (sz+1)*9 + (sy+1)*3 + (sx+1); 13 is the rank itself. */
typedef struct shhalo_geometry {
  int grid[3], coord[3], rank, nranks;
  int periodic[3];
  double lo[3], hi[3];   /* the global orthogonal box */
  double blo[3], bhi[3]; /* this rank's brick [blo, bhi) */
  double cut;            /* ghost cutoff: 2 max Rmax + skin */
  int peer[27];          /* rank a direction leads to; -1 for none (open boundary) and for code 13 */
  double shift[27][3];   /* added to the position of a row sent in that direction (a box length where the hop
                            crosses a periodic boundary) */
} shhalo_geometry;

/* Most cubic px >= py >= pz factorisation of nranks (LAMMPS' default processor-grid idea). */
int shhalo_proc_grid(int nranks, int grid[3]);

/* Rank <-> brick: rank = (ix*py + iy)*pz + iz.  Fails with SHPAIR_EINVAL if a decomposed brick edge is shorter
 * than `cut` or an undecomposed periodic edge shorter than 2 cut. */
int shhalo_plan_geometry(const int grid[3], const double lo[3], const double hi[3], const int periodic[3], double cut,
                         int rank, shhalo_geometry *out);

/* Comm::exchange, decision part: wraps x[n][3] into the periodic box in place and returns the owning rank of every
 * row (the brick it lies in; coordinates outside an open boundary belong to the outermost brick). */
int shhalo_plan_owner(const shhalo_geometry *g, int n, double *x, int *owner);

/* Comm::borders, decision part: bit c of mask[i] is set iff owned row i becomes a ghost of the neighbour in
 * direction code c:  x_d >= bhi_d - cut for s_d = +1,  x_d < blo_d + cut for s_d = -1, and the direction has a peer. */
int shhalo_plan_ghost_mask(const shhalo_geometry *g, int n, const double *x, unsigned *mask);

/* Message layout from the per-direction counts.  All arrays are indexed by MY direction code (27 entries).
 * Send rows are ordered by (peer rank, my code); what arrives through my direction c was sent by peer[c] with
 * ITS code 26 - c, and the ghost rows are ordered by (peer rank, sender's code) — so that the rows exchanged
 * with one peer are contiguous on both sides and travel as one message.  The rank is its own peer where a
 * periodic dimension is not decomposed; those blocks are copied locally. */
typedef struct shhalo_layout {
  int send_off[27], send_cnt[27]; /* rows of the send buffer */
  int recv_off[27], recv_cnt[27]; /* ghost rows (relative to nlocal) */
  int nsend, nghost;
  int npeers;                     /* distinct REMOTE peers, ascending rank */
  int peer_rank[26];
  int peer_send_off[26], peer_send_cnt[26];
  int peer_recv_off[26], peer_recv_cnt[26];
} shhalo_layout;
int shhalo_plan_layout(const shhalo_geometry *g, const int send_cnt[27], const int recv_cnt[27], shhalo_layout *out);

/* ---- contexts and transports ------------------------------------------------------------------------------ */

/* ncclGetUniqueId(): rank 0 calls this and hands the bytes to every rank with whatever the host has
 * (MPI_Bcast in LAMMPS, a torch.distributed broadcast in bench.py). */
int shhalo_get_unique_id(unsigned char id[SHHALO_UNIQUE_ID_BYTES]);

/* One context per rank, bound to the rank's shpair context (whose shapes must be set: the ghost cutoff is
 * 2 max Rmax + skin).  Sets the shpair context's neighbour-build box to the brick plus the ghost shell.
 * RCCL form: collective over all ranks (ncclCommInitRank). */
int shhalo_create_rccl(shhalo_ctx **out, shpair_ctx *sp, const unsigned char id[SHHALO_UNIQUE_ID_BYTES], int rank,
                       int nranks, const int grid[3], const double lo[3], const double hi[3], const int periodic[3],
                       double skin);
/* In-process form: the ranks are host threads of one process sharing a hub (created for nranks; each rank
 * creates its context from its own thread or from one thread before the rank threads start).  nranks = 1 needs
 * no hub (NULL): every exchange is then a local copy (a self-periodic single rank). */
int shhalo_hub_create(shhalo_hub **out, int nranks);
void shhalo_hub_destroy(shhalo_hub *hub);
int shhalo_create_local(shhalo_ctx **out, shpair_ctx *sp, shhalo_hub *hub, int rank, int nranks, const int grid[3],
                        const double lo[3], const double hi[3], const int periodic[3], double skin);
/* Host-staged form: the bytes between ranks travel through functions of the CALLER — MPI_Isend / MPI_Irecv /
 * MPI_Allreduce in a LAMMPS host without GPU-aware MPI or RCCL, torch.distributed (gloo) in bench.py --transport
 * staged.  Per exchange the library copies its packed send buffers to page-locked host memory, waits for the stream,
 * calls `exchange` once with every message of the exchange (at most one send and one receive per peer; it returns when
 * all of them are complete; 0 = success), and copies the received bytes up.  `allreduce` combines n host values over
 * all ranks in place: kind 0 = max of int32, 1 = sum of double.  Same plan, same kernels, same loop as the other
 * transports; a host wait per exchange instead of none — the way to run N ranks where RCCL cannot (and the fallback
 * bench.py takes, and says it took, when ncclCommInitRank fails on any rank).  Collective only through the callbacks. */
typedef int (*shhalo_exchange_fn)(void *user, int nsend, const int *send_peer, void *const *send_ptr,
                                  const size_t *send_bytes, int nrecv, const int *recv_peer, void *const *recv_ptr,
                                  const size_t *recv_bytes);
typedef int (*shhalo_allreduce_fn)(void *user, void *data, int n, int kind);
int shhalo_create_staged(shhalo_ctx **out, shpair_ctx *sp, shhalo_exchange_fn exchange, shhalo_allreduce_fn allreduce,
                         void *user, int rank, int nranks, const int grid[3], const double lo[3], const double hi[3],
                         const int periodic[3], double skin);
void shhalo_destroy(shhalo_ctx *h);
const char *shhalo_last_error(const shhalo_ctx *h);
int shhalo_get_geometry(const shhalo_ctx *h, shhalo_geometry *out);

/* Device pointers of one rank's particles; every array holds nmax rows (owned rows first, then ghosts; v, angmom,
 * mask are only used for owned rows but must be able to take nmax rows, because owned atoms arrive by migration). */
typedef struct shhalo_arrays {
  int nlocal, nmax;
  double *x, *v, *quat, *angmom, *f, *torque;
  int *type, *shtype, *mask, *tag;  /* tag: global ids, unique over all ranks */
} shhalo_arrays;

/* Comm::exchange: wraps the owned rows into the periodic box, sends those that left the brick to their new owner
 * (one message per peer) and takes in the arrivals; a->nlocal is updated.  Blocks (counts are read back).
 * Fails with SHPAIR_ESTATE if an atom left the brick AND its 26 neighbours (lost atom), SHPAIR_ENOMEM if nmax is
 * too small — on EVERY rank, before any row travels (see "Errors across ranks" above). */
int shhalo_exchange_device(shhalo_ctx *h, shhalo_arrays *a, void *stream);

/* Comm::borders: selects the ghosts for all 26 directions, exchanges counts, builds the send lists on the device
 * and fills the ghost rows nlocal .. nlocal+nghost-1 of x, quat, type, shtype, tag.  Blocks.  The plan stays
 * valid until the next call. */
int shhalo_borders_device(shhalo_ctx *h, const shhalo_arrays *a, int *nghost, void *stream);

/* Neighbor::build of the bound shpair context over the brick plus its ghost shell (shstep_neighbor_build_device with
 * the global ids as tags), collective like the two calls above: a shape index outside the table found by one rank's
 * build fails the call on every rank.  Blocks. */
int shhalo_neighbor_build_device(shhalo_ctx *h, const shhalo_arrays *a, int nghost, int *npairs, void *stream);

/* Comm::forward_comm: owners' x (+ shift) and quat -> the ghost rows of every neighbour.  Enqueues one pack
 * kernel, one RCCL group and one unpack kernel on `stream`; does not wait. */
int shhalo_forward_device(shhalo_ctx *h, double *x_dev, double *quat_dev, void *stream);
/* Comm::reverse_comm: ghost rows of f and torque -> added into their owners' rows.  Enqueue only. */
int shhalo_reverse_device(shhalo_ctx *h, double *f_dev, double *torque_dev, void *stream);

/* Neighbor::decide over all ranks: *rebuild = 1 if an owned atom of ANY rank moved more than skin/2 since the
 * last list build of the bound shpair context (one max-all-reduce of a flag; blocks for the read-back). */
int shhalo_check_rebuild_device(shhalo_ctx *h, int nlocal, const double *x_dev, int *rebuild, void *stream);

/* Thermo sums: in-place sum over all ranks of n doubles on the device.  Enqueue only. */
int shhalo_allreduce_sum_device(shhalo_ctx *h, double *data_dev, int n, void *stream);

/* Transport self-test: one exchange() of the context's transport in which this rank sends `nbytes` bytes to ITSELF and
 * receives them (RCCL: ncclGroupStart, ncclRecv from self, ncclSend to self, ncclGroupEnd — point-to-point calls whose
 * peer is the caller are legal inside one group), plus an in-place max / sum all-reduce, all checked against what was
 * sent.  The halo loop never sends to self (periodic self-images are local copies), so on a one-GPU box this is the only
 * way the ncclSend / ncclRecv binding executes at all.  Not collective beyond the all-reduce (every rank calls it or none).
 * Returns 0, or SHPAIR_ESTATE with the mismatch in shhalo_last_error. */
int shhalo_transport_selftest(shhalo_ctx *h, int nbytes, void *stream);

/* Counters since creation (host values, no synchronisation). */
typedef struct shhalo_stats {
  int nranks_transport;        /* what the transport reports (ncclCommCount for RCCL) */
  int npeers;                  /* distinct remote peers of the current plan */
  int nsend_rows, nghost_rows; /* current plan */
  long long rebuilds, migrated_out, migrated_in;
  long long forward_bytes_per_step, reverse_bytes_per_step; /* bytes this rank sends to remote peers */
  int transport;               /* 0 local, 1 RCCL, 2 host-staged */
  int rccl_version;            /* ncclGetVersion, 0 for the local transport */
} shhalo_stats;
int shhalo_get_stats(const shhalo_ctx *h, shhalo_stats *out);

/* Verlet::run over all ranks for nsteps (every rank calls it with the same nsteps and check_every):
 * initial_integrate -> [every check_every steps: rebuild test over all ranks -> exchange + borders + neighbour
 * build] -> forward -> clear -> pair compute -> reverse -> post_force (gravity / viscous, if any is non-zero) ->
 * final_integrate, all on `stream`; the host only waits at the rebuild tests.  On entry the plan, ghosts and list of
 * the current positions must exist (shhalo_exchange_device + shhalo_borders_device +
 * shstep_neighbor_build_device with tags) and f, torque must hold their forces (as after Verlet::setup).
 * a->nlocal and *nghost are updated.  kernel_ms (nullable): sum of the pair-kernel times of the steps (hipEvent pairs
 * recorded on `stream` around each slot range of a step — with "halo_overlap" there are up to three — so the waits
 * for the exchange between the ranges are not in it).  Blocks until the last step is done. */
typedef struct shhalo_run_params {
  double dt;
  int groupbit;
  double gravity[3], gamma_t, gamma_r;
  int check_every;
  int eflag_last; /* 1: the last step tallies energy / virial into ev_dev (7 doubles, ADDED to) */
  double *ev_dev;
} shhalo_run_params;
int shhalo_run_device(shhalo_ctx *h, shhalo_arrays *a, const shhalo_run_params *p, int nsteps, int *nghost,
                      int *rebuilds, double *kernel_ms, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SHHALO_H */
