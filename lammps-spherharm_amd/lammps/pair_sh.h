/* -*- c++ -*- ----------------------------------------------------------
   pair_sh.h — LAMMPS-side adapter of the MI355X `pair_style sh` path.

   Drop this file and pair_sh.cpp into LAMMPS' src/ (or load it through the
   PLUGIN package) and link libshpair.so: the class keeps
   LAMMPS' Pair contract (settings / coeff / init_style / init_one / compute)
   and forwards the per-step work to the C ABI of include/shpair.h.

   The reference's own PairSH (pair_sh.cpp of LAMMPS-SPHERHARM) is ABSENT FROM
   MOUNT (/root/reference/README.md:1 is the whole mount), so this adapter is
   written against the stock LAMMPS `Pair` interface, not copied or derived
   from it, and cannot cite its lines.
------------------------------------------------------------------------- */

#ifdef PAIR_CLASS
// clang-format off
PairStyle(sh,PairSH);
PairStyle(sh/hip,PairSH);
// clang-format on
#else

#ifndef LMP_PAIR_SH_H
#define LMP_PAIR_SH_H

#include "pair.h"

#include <cstddef>
#include <string>
#include <vector>

struct shpair_ctx;

namespace LAMMPS_NS {

class PairSH : public Pair {
 public:
  PairSH(class LAMMPS *);
  ~PairSH() override;
  void compute(int, int) override;
  void settings(int, char **) override;
  void coeff(int, char **) override;
  void init_style() override;
  double init_one(int, int) override;
  void *extract(const char *, int &) override;
  int pack_forward_comm(int, int *, double *, int, int *) override;
  void unpack_forward_comm(int, int, double *) override;

 protected:
  struct shpair_ctx *ctx;
  int nq;                               // pair_style sh <nq>
  int device;                           // HIP device (default: local rank of the node communicator)
  int rule;                             // 0 sharp inside test, 1 covered-fraction weights (docs/SPEC.md §2.8)
  std::vector<std::string> shape_files; // pair_style ... shapes f1 f2 ...
  int nshapes;
  double **kn, **exponent;              // [ntypes+1][ntypes+1], as pair_coeff sets them
  double maxrad;                        // largest bounding radius over all shapes
  bigint last_neigh_build;              // neighbor->lastcall of the list already uploaded
  double **quat_comm;                   // the array forward communication packs from / unpacks into
  int quat_is_custom;                   // orientation comes from fix property/atom: ghosts must be refreshed here

  // host arrays page-locked for the per-step copies (shpair_pin_host): x, quat, f, torque, type, shtype — re-pinned
  // when LAMMPS reallocates them (the pointer or atom->nmax changed)
  void *pinned_ptr[6];
  size_t pinned_bytes[6];
  void pin(int slot, void *ptr, size_t bytes);

  void allocate();
  void load_shapes();
  void check(int rc, const char *what);
};

}    // namespace LAMMPS_NS

#endif
#endif
