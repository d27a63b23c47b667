/* -*- c++ -*- ----------------------------------------------------------
   fix_nve_sh.h — LAMMPS-side adapter of the MI355X rigid-body integrator
   of SH particles (include/shstep.h, docs/SPEC.md §6).

     fix ID group nve/sh [density rho_1 ... rho_nshapes]

   Velocity-Verlet translation of the centre of mass plus the Richardson
   quaternion update of LAMMPS' aspherical integrators, with the full
   (non-diagonal) inertia tensor and centre-of-mass offset of each SH shape.
   Shares the device context of the run's `pair_style sh` (PairSH::extract
   "ctx"), which owns the shape tables.

   The reference's own integrator fix is ABSENT FROM MOUNT
   (/root/reference/README.md:1 is the whole mount): written against the
   stock LAMMPS `Fix` interface, not derived from it.
------------------------------------------------------------------------- */

#ifdef FIX_CLASS
// clang-format off
FixStyle(nve/sh,FixNVESH);
// clang-format on
#else

#ifndef LMP_FIX_NVE_SH_H
#define LMP_FIX_NVE_SH_H

#include "fix.h"

#include <vector>

struct shpair_ctx;

namespace LAMMPS_NS {

class FixNVESH : public Fix {
 public:
  FixNVESH(class LAMMPS *, int, char **);
  int setmask() override;
  void init() override;
  void initial_integrate(int) override;
  void final_integrate() override;
  void reset_dt() override;

 protected:
  struct shpair_ctx *ctx;
  double dtv;
  std::vector<double> density;    // per shape; empty = leave the context's values (default 1)

  void step(int phase);
  void check(int rc, const char *what);
};

}    // namespace LAMMPS_NS

#endif
#endif
