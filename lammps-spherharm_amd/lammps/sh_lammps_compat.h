/* -*- c++ -*- ----------------------------------------------------------
   sh_lammps_compat.h — which LAMMPS API generation pair_sh.cpp and
   fix_nve_sh.cpp are compiled against.

   LAMMPS renamed or moved every one of the few calls the adapters make
   between 2019 and 2022, each at a different date.  One switch per call:

     SHPAIR_LMP_EV_SETUP           Pair::ev_setup(eflag, vflag)            instead of Pair::ev_init(eflag, vflag)
     SHPAIR_LMP_FORCE_BOUNDS       Force::bounds(FLERR, str, nmax, lo, hi) instead of utils::bounds(FLERR, str, 1, nmax, lo, hi, error)
     SHPAIR_LMP_FIND_CUSTOM_2ARG   Atom::find_custom(name, flag)           instead of Atom::find_custom(name, flag, cols):
                                   such a tree has NO 2-d custom per-atom arrays (fix property/atom d2_quat 4), so the
                                   orientations must come from the atom style (atom->extract("quat")); the shape
                                   index may still be a custom integer vector (fix property/atom i_shtype)
     SHPAIR_LMP_NEIGH_REQUEST      Neighbor::request(this, instance_me)    instead of Neighbor::add_request(this)
     SHPAIR_LMP_FORWARD_COMM_PAIR  Comm::forward_comm_pair(this)           instead of Comm::forward_comm(this)

   Give them one by one (-DSHPAIR_LMP_NEIGH_REQUEST=1), or name the tree:

     -DSHPAIR_LAMMPS_VERSION=yyyymmdd     the date in the tree's src/version.h, as a number
                                          (stable_29Sep2021 -> 20210929)

   from which every switch that was not given is derived.  With neither, the adapters are written for current LAMMPS
   (2022-06 and later).  -DSHPAIR_LAMMPS_OLD_API (rounds 2-4) still means what it meant: Force::bounds +
   Neighbor::request + Comm::forward_comm_pair.

   [PRIOR] Every date below is the builder's recollection of stock LAMMPS' history.  There is no LAMMPS tree in this
   image and none in /root/reference (README.md:1 is the whole mount): the dates are unverified, the adapters have
   only met lammps/stub/lammps_stub.h, which models these five generations.  If a tree near one of the dates does not
   compile, flip that one switch by hand — each is independent of the others.
------------------------------------------------------------------------- */

#ifndef SH_LAMMPS_COMPAT_H
#define SH_LAMMPS_COMPAT_H

#ifdef SHPAIR_LAMMPS_OLD_API
#ifndef SHPAIR_LMP_FORCE_BOUNDS
#define SHPAIR_LMP_FORCE_BOUNDS 1
#endif
#ifndef SHPAIR_LMP_NEIGH_REQUEST
#define SHPAIR_LMP_NEIGH_REQUEST 1
#endif
#ifndef SHPAIR_LMP_FORWARD_COMM_PAIR
#define SHPAIR_LMP_FORWARD_COMM_PAIR 1
#endif
#endif

#ifdef SHPAIR_LAMMPS_VERSION
#ifndef SHPAIR_LMP_EV_SETUP
#define SHPAIR_LMP_EV_SETUP (SHPAIR_LAMMPS_VERSION < 20190329)            /* [PRIOR] Pair::ev_init: spring 2019 */
#endif
#ifndef SHPAIR_LMP_FORCE_BOUNDS
#define SHPAIR_LMP_FORCE_BOUNDS (SHPAIR_LAMMPS_VERSION < 20200821)        /* [PRIOR] utils::bounds: August 2020 */
#endif
#ifndef SHPAIR_LMP_FIND_CUSTOM_2ARG
#define SHPAIR_LMP_FIND_CUSTOM_2ARG (SHPAIR_LAMMPS_VERSION < 20210730)    /* [PRIOR] custom per-atom arrays: July 2021 */
#endif
#ifndef SHPAIR_LMP_NEIGH_REQUEST
#define SHPAIR_LMP_NEIGH_REQUEST (SHPAIR_LAMMPS_VERSION < 20220324)       /* [PRIOR] Neighbor::add_request: March 2022 */
#endif
#ifndef SHPAIR_LMP_FORWARD_COMM_PAIR
#define SHPAIR_LMP_FORWARD_COMM_PAIR (SHPAIR_LAMMPS_VERSION < 20220504)   /* [PRIOR] Comm::forward_comm(Pair *): spring 2022 */
#endif
#endif

#ifndef SHPAIR_LMP_EV_SETUP
#define SHPAIR_LMP_EV_SETUP 0
#endif
#ifndef SHPAIR_LMP_FORCE_BOUNDS
#define SHPAIR_LMP_FORCE_BOUNDS 0
#endif
#ifndef SHPAIR_LMP_FIND_CUSTOM_2ARG
#define SHPAIR_LMP_FIND_CUSTOM_2ARG 0
#endif
#ifndef SHPAIR_LMP_NEIGH_REQUEST
#define SHPAIR_LMP_NEIGH_REQUEST 0
#endif
#ifndef SHPAIR_LMP_FORWARD_COMM_PAIR
#define SHPAIR_LMP_FORWARD_COMM_PAIR 0
#endif

#include "atom.h"

namespace sh_lammps {

/* Per-atom orientations [nall][4] (w x y z): the atom style's own array, or — where the tree has 2-d custom per-atom
   arrays — `fix property/atom d2_quat 4 ghost yes`.  is_custom tells the pair style that it has to forward the
   owners' values to the ghosts itself. */
static inline double **find_quat(LAMMPS_NS::Atom *atom, int &is_custom)
{
  is_custom = 0;
  double **quat = (double **) atom->extract("quat");
#if !SHPAIR_LMP_FIND_CUSTOM_2ARG
  int flag = 0, cols = 0;
  int idx;
  if (!quat && (idx = atom->find_custom("quat", flag, cols)) >= 0 && flag == 1 && cols == 4) {
    quat = atom->darray[idx];
    is_custom = 1;
  }
#endif
  return quat;
}

/* Per-atom shape index (0-based): the atom style's array, or `fix property/atom i_shtype`. */
static inline int *find_shtype(LAMMPS_NS::Atom *atom)
{
  int *shtype = (int *) atom->extract("shtype");
  if (shtype) return shtype;
  int flag = 0, idx;
#if SHPAIR_LMP_FIND_CUSTOM_2ARG
  if ((idx = atom->find_custom("shtype", flag)) >= 0 && flag == 0) shtype = atom->ivector[idx];
#else
  int cols = 0;
  if ((idx = atom->find_custom("shtype", flag, cols)) >= 0 && flag == 0 && cols == 0) shtype = atom->ivector[idx];
#endif
  return shtype;
}

static inline const char *quat_requirement()
{
#if SHPAIR_LMP_FIND_CUSTOM_2ARG
  return "pair sh requires per-atom quaternions from the atom style (atom->extract(\"quat\"), e.g. atom_style spherharm): this "
         "LAMMPS predates 2-d custom per-atom arrays, fix property/atom cannot hold them";
#else
  return "pair sh requires per-atom quaternions (atom_style spherharm, or fix property/atom d2_quat 4 ghost yes)";
#endif
}

}    // namespace sh_lammps

#endif
