/* ----------------------------------------------------------------------
   fix_nve_sh.cpp — see fix_nve_sh.h.  Per step the fix hands LAMMPS' own
   per-atom arrays (x, v, angmom, f, torque, mask; quat and shtype as for
   pair_style sh) to shstep_nve(), which stages them through the GPU; a
   device-resident host calls shstep_nve_device() instead.
   v is the velocity of the centre of mass and angmom the angular momentum
   about it (space frame), x the SH origin the pair style reads.
------------------------------------------------------------------------- */

#include "fix_nve_sh.h"

#include "sh_lammps_compat.h"

#include "atom.h"
#include "error.h"
#include "force.h"
#include "pair.h"
#include "update.h"

#include "shstep.h"

#include <cstdlib>
#include <cstring>
#include <string>

using namespace LAMMPS_NS;
using namespace FixConst;

FixNVESH::FixNVESH(LAMMPS *lmp, int narg, char **arg) : Fix(lmp, narg, arg), ctx(nullptr), dtv(0.0)
{
  if (narg < 3) error->all(FLERR, "Illegal fix nve/sh command");
  int iarg = 3;
  while (iarg < narg) {
    if (strcmp(arg[iarg], "density") == 0) {
      iarg++;
      while (iarg < narg) {
        char *end = nullptr;
        const double rho = strtod(arg[iarg], &end);
        if (end == arg[iarg] || *end != '\0') break;
        if (!(rho > 0.0)) error->all(FLERR, "fix nve/sh: density must be > 0");
        density.push_back(rho);
        iarg++;
      }
      if (density.empty()) error->all(FLERR, "fix nve/sh: density needs at least one value");
    } else
      error->all(FLERR, "Illegal fix nve/sh command");
  }
  time_integrate = 1;
}

int FixNVESH::setmask()
{
  return INITIAL_INTEGRATE | FINAL_INTEGRATE;
}

void FixNVESH::check(int rc, const char *what)
{
  if (rc == SHPAIR_OK) return;
  std::string msg = std::string("fix nve/sh: ") + what + ": " + shpair_strerror(rc);
  if (ctx) msg += std::string(" — ") + shpair_last_error(ctx);
  error->all(FLERR, msg.c_str());
}

void FixNVESH::init()
{
  dtv = update->dt;
  Pair *pair = force->pair_match("sh", 0);
  int dim = 0;
  ctx = pair ? (shpair_ctx *) pair->extract("ctx", dim) : nullptr;
  if (!ctx) error->all(FLERR, "fix nve/sh requires pair_style sh (it owns the shape tables and the device context)");
  if (!atom->angmom || !atom->torque) error->all(FLERR, "fix nve/sh requires per-atom angmom and torque");
  for (size_t s = 0; s < density.size(); s++) check(shstep_set_density(ctx, (int) s, density[s]), "shstep_set_density");
}

void FixNVESH::reset_dt()
{
  dtv = update->dt;
}

void FixNVESH::step(int phase)
{
  int nlocal = atom->nlocal;
  if (igroup == atom->firstgroup) nlocal = atom->nfirst;
  if (nlocal == 0) return;
  int custom = 0;
  double **quat = sh_lammps::find_quat(atom, custom);
  int *shtype = sh_lammps::find_shtype(atom);
  if (!quat || !shtype) error->one(FLERR, "fix nve/sh: per-atom quaternions / shape index not found");
  check(shstep_nve(ctx, phase, nlocal, dtv, atom->x[0], atom->v[0], quat[0], atom->angmom[0], atom->f[0], atom->torque[0],
                   shtype, atom->mask, groupbit),
        phase == 0 ? "shstep_nve (initial_integrate)" : "shstep_nve (final_integrate)");
}

void FixNVESH::initial_integrate(int /*vflag*/)
{
  step(0);
}

void FixNVESH::final_integrate()
{
  step(1);
}
