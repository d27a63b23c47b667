/* ----------------------------------------------------------------------
   pair_sh.cpp — LAMMPS Pair adapter over the shpair C ABI (include/shpair.h).

   Input script (everything else of the run is unmodified LAMMPS):

     pair_style sh <nq> [device <id>] [shapes <file> ...]
     pair_coeff I J <kn> <exponent>

   Per-atom data, looked up once per compute():
     orientation  atom->extract("quat")  or custom d2_quat   (nall x 4, w x y z)
     shape index  atom->extract("shtype") or custom i_shtype (0-based)
   Shape tables: the files named after `shapes`, one per shape index, text:
     [line 1: lmax ;] then one line per coefficient: n m Re(a_nm) Im(a_nm) — m >= 0, or the whole range -n..n of a
     real radius (load_shapes below)
   (the reference's own shape-file format is unknown: its reader is absent
   from the mount).

   Reference: PairSH of LAMMPS-SPHERHARM is ABSENT FROM MOUNT; written against
   the stock LAMMPS Pair interface.
------------------------------------------------------------------------- */

#include "pair_sh.h"

#include "sh_lammps_compat.h"    // which LAMMPS API generation (SHPAIR_LAMMPS_VERSION / SHPAIR_LMP_* switches)

#include "atom.h"
#include "comm.h"
#include "error.h"
#include "force.h"
#include "memory.h"
#include "neigh_list.h"
#include "neighbor.h"
#include "update.h"
#include "utils.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "shpair.h"

using namespace LAMMPS_NS;

/* ---------------------------------------------------------------------- */

PairSH::PairSH(LAMMPS *lmp) :
    Pair(lmp), ctx(nullptr), nq(16), device(-1), rule(0), nshapes(0), kn(nullptr), exponent(nullptr), maxrad(0.0),
    last_neigh_build(-1), quat_comm(nullptr), quat_is_custom(0)
{
  // Orientations kept in a custom per-atom array (fix property/atom d2_quat 4 ghost yes) reach the ghost atoms only at
  // Comm::borders(), i.e. at reneighbourings, while the integrator turns the owners every step: the pair style
  // forwards them itself (Comm::forward_comm(this), 4 doubles per ghost) at the top of compute().  An atom style
  // that carries the quaternion (atom->extract("quat")) is expected to pack it in its own pack_comm, as LAMMPS'
  // aspherical atom styles do.
  for (int k = 0; k < 6; k++) {
    pinned_ptr[k] = nullptr;
    pinned_bytes[k] = 0;
  }
  comm_forward = 4;
  single_enable = 0;
  restartinfo = 0;
  no_virial_fdotr_compute = 1;    // the virial comes back from the device, tallied per pair
  manybody_flag = 0;
}

PairSH::~PairSH()
{
  if (ctx) shpair_destroy(ctx);    // also releases the page locks
  if (allocated) {
    memory->destroy(setflag);
    memory->destroy(cutsq);
    memory->destroy(kn);
    memory->destroy(exponent);
  }
}

/* page-lock one of LAMMPS' per-atom arrays for the per-step copies; a failure is not an error (the copies are then
   staged by the HIP runtime, as for any pageable memory) */

void PairSH::pin(int slot, void *ptr, size_t bytes)
{
  if (pinned_ptr[slot] == ptr && pinned_bytes[slot] == bytes) return;
  if (pinned_ptr[slot]) shpair_unpin_host(ctx, pinned_ptr[slot]);    // the old array: LAMMPS has reallocated it
  pinned_ptr[slot] = nullptr;
  pinned_bytes[slot] = 0;
  if (ptr && bytes && shpair_pin_host(ctx, ptr, bytes) == SHPAIR_OK) {
    pinned_ptr[slot] = ptr;
    pinned_bytes[slot] = bytes;
  }
}

void PairSH::check(int rc, const char *what)
{
  if (rc == SHPAIR_OK) return;
  char msg[640];
  snprintf(msg, sizeof(msg), "pair sh: %s failed: %s (%s)", what, shpair_strerror(rc),
           ctx ? shpair_last_error(ctx) : "no context");
  error->all(FLERR, msg);
}

/* ----------------------------------------------------------------------
   pair_style sh <nq> [device <id>] [rule sharp|weighted] [shapes f1 f2 ...]
------------------------------------------------------------------------- */

void PairSH::settings(int narg, char **arg)
{
  if (narg < 1) error->all(FLERR, "Illegal pair_style sh command: pair_style sh <nq> [device id] [shapes files...]");
  nq = atoi(arg[0]);
  if (nq < 1 || nq > SHPAIR_MAX_NQ) error->all(FLERR, "pair_style sh: quadrature order out of range");
  shape_files.clear();
  int iarg = 1;
  while (iarg < narg) {
    if (strcmp(arg[iarg], "device") == 0) {
      if (iarg + 2 > narg) error->all(FLERR, "Illegal pair_style sh command: device needs an id");
      device = atoi(arg[iarg + 1]);
      iarg += 2;
    } else if (strcmp(arg[iarg], "rule") == 0) {
      if (iarg + 2 > narg) error->all(FLERR, "Illegal pair_style sh command: rule needs sharp or weighted");
      if (strcmp(arg[iarg + 1], "sharp") == 0) rule = 0;
      else if (strcmp(arg[iarg + 1], "weighted") == 0) rule = 1;
      else error->all(FLERR, "Illegal pair_style sh command: rule must be sharp or weighted");
      iarg += 2;
    } else if (strcmp(arg[iarg], "shapes") == 0) {
      ++iarg;
      while (iarg < narg && strcmp(arg[iarg], "device") != 0 && strcmp(arg[iarg], "rule") != 0)
        shape_files.emplace_back(arg[iarg++]);
      if (shape_files.empty()) error->all(FLERR, "Illegal pair_style sh command: shapes needs file names");
    } else
      error->all(FLERR, "Illegal pair_style sh command: unknown keyword");
  }
  if (!ctx) {
    // one rank per GPU: default device = rank within the node
    int dev = device;
    if (dev < 0) {
      // the rank within the node, from whichever launcher started the run: Open MPI, MVAPICH2, MPICH / Hydra (what
      // this image ships), Intel MPI, PALS, Slurm, Flux
      static const char *const vars[] = {"OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK", "MPI_LOCALRANKID",
                                         "PMI_LOCAL_RANK", "PALS_LOCAL_RANKID", "SLURM_LOCALID", "FLUX_TASK_LOCAL_ID",
                                         "LOCAL_RANK"};
      const char *lr = nullptr;
      for (const char *v : vars)
        if ((lr = getenv(v)) != nullptr && *lr) break;
      dev = (lr && *lr) ? atoi(lr) : 0;
    }
    const int rc = shpair_create(&ctx, dev);
    if (rc != SHPAIR_OK) {
      char msg[256];
      snprintf(msg, sizeof(msg), "pair sh: cannot open HIP device %d: %s", dev, shpair_strerror(rc));
      error->all(FLERR, msg);    // no CPU fallback by design
    }
  }
  check(shpair_settings(ctx, nq), "shpair_settings");
  check(shpair_set_option(ctx, "rule", rule), "shpair_set_option(rule)");
}

void PairSH::allocate()
{
  allocated = 1;
  const int n = atom->ntypes;
  memory->create(setflag, n + 1, n + 1, "pair:setflag");
  memory->create(cutsq, n + 1, n + 1, "pair:cutsq");
  memory->create(kn, n + 1, n + 1, "pair:kn");
  memory->create(exponent, n + 1, n + 1, "pair:exponent");
  for (int i = 0; i <= n; i++)
    for (int j = 0; j <= n; j++) {
      setflag[i][j] = 0;
      kn[i][j] = 0.0;
      exponent[i][j] = 1.0;
    }
}

/* ----------------------------------------------------------------------
   shape tables -> shpair_set_shape()
------------------------------------------------------------------------- */

void PairSH::load_shapes()
{
  if (shape_files.empty()) error->all(FLERR, "pair sh: no shape tables: add `shapes <file> ...` to pair_style sh");
  nshapes = (int) shape_files.size();
  check(shpair_set_ntypes(ctx, atom->ntypes, nshapes), "shpair_set_ntypes");
  maxrad = 0.0;
  for (int s = 0; s < nshapes; s++) {
    FILE *fp = fopen(shape_files[s].c_str(), "r");
    if (!fp) error->one(FLERR, "pair sh: cannot open shape file");
    // line based: `#` starts a comment, blank lines are skipped.  Data lines are `n m Re Im` for any subset of the
    // coefficients (the others are zero).  An optional FIRST data line with a single integer names lmax (this repo's
    // writer puts it there); without it lmax is the largest n in the file.  m may be negative: files that list the whole
    // range -n..n (a common layout of SH coefficient tables; [PRIOR]: the reference's own format is unknown, its reader
    // is absent from the mount) are accepted when they describe a REAL radius, a_{n,-m} = (-1)^m conj(a_{n,m}) — checked
    // to 1e-9 of the largest coefficient where both are listed; a coefficient listed only at -m fills +m.  Anything
    // else is an error, not an end of file.
    int lmax = -1;
    bool first = true;
    struct Entry { int n, m; double re, im; };
    std::vector<Entry> ent;
    char line[512];
    while (fgets(line, sizeof(line), fp)) {
      if (char *hash = strchr(line, '#')) *hash = '\0';
      char *p = line;
      while (*p == ' ' || *p == '\t' || *p == '\r' || *p == '\n') ++p;
      if (*p == '\0') continue;
      int n, m;
      double re, im;
      char extra;
      if (first) {
        first = false;
        int l0;
        if (sscanf(p, "%d %c", &l0, &extra) == 1) {
          if (l0 < 0 || l0 > SHPAIR_MAX_LMAX) {
            fclose(fp);
            error->one(FLERR, "pair sh: bad lmax in shape file");
          }
          lmax = l0;
          continue;
        }
      }
      if (sscanf(p, "%d %d %lf %lf %c", &n, &m, &re, &im, &extra) != 4) {
        fclose(fp);
        error->one(FLERR, "pair sh: malformed line in shape file (expected: n m Re Im)");
      }
      if (n < 0 || n > SHPAIR_MAX_LMAX || m < -n || m > n || (lmax >= 0 && n > lmax)) {
        fclose(fp);
        error->one(FLERR, "pair sh: (n, m) out of range in shape file");
      }
      ent.push_back({n, m, re, im});
    }
    fclose(fp);
    if (lmax < 0) {
      if (ent.empty()) error->one(FLERR, "pair sh: empty shape file");
      for (const Entry &e : ent) lmax = e.n > lmax ? e.n : lmax;
    }
    std::vector<double> anm((size_t) (lmax + 1) * (lmax + 2), 0.0);
    std::vector<char> have((size_t) (lmax + 1) * (lmax + 2) / 2, 0);
    double amax = 0.0;
    for (const Entry &e : ent) amax = fmax(amax, fmax(fabs(e.re), fabs(e.im)));
    for (const Entry &e : ent) {    // m >= 0 first
      if (e.m < 0) continue;
      if (e.m == 0 && fabs(e.im) > 1e-9 * amax) error->one(FLERR, "pair sh: a_n0 must be real in shape file");
      const int k = e.n * (e.n + 1) / 2 + e.m;
      anm[2 * k] = e.re;
      anm[2 * k + 1] = e.m == 0 ? 0.0 : e.im;
      have[k] = 1;
    }
    for (const Entry &e : ent) {    // then the mirror half, checked against it
      if (e.m >= 0) continue;
      const int mp = -e.m;
      const int k = e.n * (e.n + 1) / 2 + mp;
      const double sgn = (mp & 1) ? -1.0 : 1.0;
      const double re = sgn * e.re, im = -sgn * e.im;    // (-1)^m conj(a_{n,-m})
      if (have[k]) {
        if (fabs(anm[2 * k] - re) > 1e-9 * amax || fabs(anm[2 * k + 1] - im) > 1e-9 * amax)
          error->one(FLERR, "pair sh: shape file does not describe a real radius: a_{n,-m} != (-1)^m conj(a_{n,m})");
      } else {
        anm[2 * k] = re;
        anm[2 * k + 1] = im;
        have[k] = 1;
      }
    }
    check(shpair_set_shape(ctx, s, lmax, anm.data(), 0.0), "shpair_set_shape");
    double r = 0.0;
    check(shpair_get_rmax(ctx, s, &r), "shpair_get_rmax");
    if (r > maxrad) maxrad = r;
  }
}

/* ----------------------------------------------------------------------
   pair_coeff I J kn exponent
------------------------------------------------------------------------- */

void PairSH::coeff(int narg, char **arg)
{
  if (narg != 4) error->all(FLERR, "Incorrect args for pair coefficients: pair_coeff I J kn exponent");
  if (!allocated) allocate();
  int ilo, ihi, jlo, jhi;
#if SHPAIR_LMP_FORCE_BOUNDS
  force->bounds(FLERR, arg[0], atom->ntypes, ilo, ihi);
  force->bounds(FLERR, arg[1], atom->ntypes, jlo, jhi);
#else
  utils::bounds(FLERR, arg[0], 1, atom->ntypes, ilo, ihi, error);
  utils::bounds(FLERR, arg[1], 1, atom->ntypes, jlo, jhi, error);
#endif
  const double k = atof(arg[2]);
  const double e = atof(arg[3]);
  if (k < 0.0) error->all(FLERR, "pair_coeff sh: kn must be >= 0");
  if (e < 1.0) error->all(FLERR, "pair_coeff sh: exponent must be >= 1");
  int count = 0;
  for (int i = ilo; i <= ihi; i++)
    for (int j = (jlo > i ? jlo : i); j <= jhi; j++) {
      kn[i][j] = kn[j][i] = k;
      exponent[i][j] = exponent[j][i] = e;
      setflag[i][j] = 1;
      count++;
    }
  if (count == 0) error->all(FLERR, "Incorrect args for pair coefficients");
}

/* ---------------------------------------------------------------------- */

void PairSH::init_style()
{
  if (!ctx) error->all(FLERR, "pair sh: pair_style sh was not processed");
  int custom = 0;
  if (!sh_lammps::find_quat(atom, custom)) error->all(FLERR, sh_lammps::quat_requirement());
  if (!sh_lammps::find_shtype(atom))
    error->all(FLERR, "pair sh requires a per-atom shape index (atom_style spherharm, or fix property/atom i_shtype)");
  load_shapes();    // also sizes the coefficient tables, so pair_coeff values go in afterwards
  for (int i = 1; i <= atom->ntypes; i++)
    for (int j = 1; j <= atom->ntypes; j++) {
      if (!setflag[i][j] && !setflag[j][i]) error->all(FLERR, "All pair coeffs are not set");
      check(shpair_set_coeff(ctx, i, j, kn[i][j], exponent[i][j]), "shpair_set_coeff");
    }
#if SHPAIR_LMP_NEIGH_REQUEST
  neighbor->request(this, instance_me);    // default request: half list
#else
  neighbor->add_request(this);    // default request: half list, newton follows the run
#endif
  last_neigh_build = -1;
}

/* cutoff of a type pair: the two largest bounding spheres may touch */

double PairSH::init_one(int i, int j)
{
  if (setflag[i][j] == 0 && setflag[j][i] == 0) error->all(FLERR, "All pair coeffs are not set");
  kn[j][i] = kn[i][j];
  exponent[j][i] = exponent[i][j];
  return 2.0 * maxrad;
}

/* ----------------------------------------------------------------------
   the hot path: one call into the HIP library per timestep
------------------------------------------------------------------------- */

void PairSH::compute(int eflag, int vflag)
{
#if SHPAIR_LMP_EV_SETUP
  if (eflag || vflag) ev_setup(eflag, vflag);
  else evflag = vflag_fdotr = eflag_global = vflag_global = eflag_atom = vflag_atom = 0;
#else
  ev_init(eflag, vflag);
#endif

  const int nlocal = atom->nlocal;
  const int nall = nlocal + atom->nghost;

  // neighbour list: re-upload only after a rebuild
  if (neighbor->lastcall != last_neigh_build) {
    check(shpair_set_neighbors(ctx, list->inum, list->ilist, list->numneigh, list->firstneigh), "shpair_set_neighbors");
    last_neigh_build = neighbor->lastcall;
  }

  // per-atom orientation and shape index
  double **quat = sh_lammps::find_quat(atom, quat_is_custom);
  if (!quat) error->one(FLERR, "pair sh: per-atom quaternions disappeared");
  // Collective: every rank calls it whenever the orientations live in a custom property — also a rank without
  // ghosts (or without atoms), whose owned atoms may be ghosts of a neighbour that waits for them.  Comm handles
  // zero-length swaps.
  if (quat_is_custom) {
    quat_comm = quat;
#if SHPAIR_LMP_FORWARD_COMM_PAIR
    comm->forward_comm_pair(this);
#else
    comm->forward_comm(this);
#endif
  }
  int *shtype = sh_lammps::find_shtype(atom);
  if (!shtype) error->one(FLERR, "pair sh: per-atom shape index disappeared");

  double eng = 0.0, vir[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double *x0 = nall ? atom->x[0] : nullptr;
  double *q0 = nall ? quat[0] : nullptr;
  double *f0 = nall ? atom->f[0] : nullptr;
  double *t0 = nall ? atom->torque[0] : nullptr;
  if (nall) {
    // the arrays keep their place until atom->nmax grows: page-locked once, copied by DMA every step
    const size_t rows = (size_t) (atom->nmax > nall ? atom->nmax : nall);
    // When anything moved (atom->nmax grew), ALL six old registrations go first: x, f and torque have the same size, the
    // allocator may hand one array another's old address, and a slot-by-slot update would then find "same pointer, same
    // bytes" on a stale entry and later unpin the registration its neighbour believes it holds.
    {
      void *want[6] = {x0, q0, f0, t0, atom->type, shtype};
      const size_t wbytes[6] = {rows * 3 * sizeof(double), rows * 4 * sizeof(double), rows * 3 * sizeof(double),
                                rows * 3 * sizeof(double), rows * sizeof(int), rows * sizeof(int)};
      bool moved = false;
      for (int k = 0; k < 6; k++) moved = moved || pinned_ptr[k] != want[k] || pinned_bytes[k] != wbytes[k];
      if (moved)
        for (int k = 0; k < 6; k++) {
          if (pinned_ptr[k]) shpair_unpin_host(ctx, pinned_ptr[k]);
          pinned_ptr[k] = nullptr;
          pinned_bytes[k] = 0;
        }
    }
    pin(0, x0, rows * 3 * sizeof(double));
    pin(1, q0, rows * 4 * sizeof(double));
    pin(2, f0, rows * 3 * sizeof(double));
    pin(3, t0, rows * 3 * sizeof(double));
    pin(4, atom->type, rows * sizeof(int));
    pin(5, shtype, rows * sizeof(int));
  }
  // per-atom tallies (compute pe/atom, stress/atom): Pair's own eatom / vatom arrays, sized by ev_init
  check(shpair_set_peratom_host(ctx, (eflag_atom && nall) ? eatom : nullptr, (vflag_atom && nall) ? vatom[0] : nullptr),
        "shpair_set_peratom_host");
  check(shpair_compute(ctx, nlocal, atom->nghost, x0, q0, atom->type, shtype, force->newton_pair, eflag_global ? 1 : 0,
                       vflag_global ? 1 : 0, f0, t0, &eng, vir),
        "shpair_compute");

  if (eflag_global) eng_vdwl += eng;
  if (vflag_global)
    for (int a = 0; a < 6; a++) virial[a] += vir[a];
}

/* ----------------------------------------------------------------------
   forward communication of the orientations (see the constructor)
------------------------------------------------------------------------- */

int PairSH::pack_forward_comm(int n, int *list, double *buf, int /*pbc_flag*/, int * /*pbc*/)
{
  int m = 0;
  for (int i = 0; i < n; i++) {
    const double *q = quat_comm[list[i]];
    buf[m++] = q[0];
    buf[m++] = q[1];
    buf[m++] = q[2];
    buf[m++] = q[3];
  }
  return m;
}

void PairSH::unpack_forward_comm(int n, int first, double *buf)
{
  int m = 0;
  for (int i = first; i < first + n; i++) {
    double *q = quat_comm[i];
    q[0] = buf[m++];
    q[1] = buf[m++];
    q[2] = buf[m++];
    q[3] = buf[m++];
  }
}

void *PairSH::extract(const char *str, int &dim)
{
  dim = 2;
  if (strcmp(str, "kn") == 0) return (void *) kn;
  if (strcmp(str, "exponent") == 0) return (void *) exponent;
  dim = 0;
  if (strcmp(str, "nq") == 0) return (void *) &nq;
  if (strcmp(str, "ctx") == 0) return (void *) ctx;    // shared with fix nve/sh
  return nullptr;
}
