// lammps_stub.h — COMPILE-CHECK AND TEST SCAFFOLD, NOT LAMMPS.
//
// There are no LAMMPS headers in this image, so the adapter
// pair_sh.{h,cpp} cannot be compiled against the real thing here.  This file
// declares, hand-written from the public LAMMPS developer documentation, only
// the members the adapter touches, with just enough behaviour behind them for
// tests/lammps_host to drive PairSH::settings/coeff/init_style/compute on a
// synthetic bed.  Nothing here is shipped or used outside that test; with a
// real LAMMPS tree the adapter includes the real headers instead.
//
// SHPAIR_STUB_GEN selects which GENERATION of the LAMMPS API the stub offers — and ONLY that one, so that an adapter
// compiled with the wrong switches (sh_lammps_compat.h) fails to compile here as it would against the real tree
// ([PRIOR]: the dates are recollections, unverified; tests/test_lammps_adapter.py compiles and runs all five):
//   4 (default)  2022-06 and later : Pair::ev_init, utils::bounds, Atom::find_custom(name, flag, cols) + 2-d custom
//                                    arrays, Neighbor::add_request(this), Comm::forward_comm(Pair *)
//   3            2021-07 .. 2022-03: as 4, but Neighbor::request(this, instance_me) and Comm::forward_comm_pair(this)
//   2            2020-08 .. 2021-07: as 3, but Atom::find_custom(name, flag) — no 2-d custom arrays
//   1            2019-04 .. 2020-08: as 2, but Force::bounds(FLERR, str, nmax, lo, hi)
//   0            before 2019-03    : as 1, but Pair::ev_setup(eflag, vflag) (no ev_init)
#pragma once
#ifndef SHPAIR_STUB_GEN
#define SHPAIR_STUB_GEN 4
#endif
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#define FLERR __FILE__, __LINE__

namespace LAMMPS_NS {

typedef int64_t bigint;

class Error {
 public:
  [[noreturn]] void all(const char *file, int line, const char *msg)
  {
    fprintf(stderr, "ERROR: %s (%s:%d)\n", msg, file, line);
    exit(1);
  }
  [[noreturn]] void one(const char *file, int line, const char *msg) { all(file, line, msg); }
};

class Memory {
 public:
  template <typename T> T **create(T **&a, int n1, int n2, const char *)
  {
    T *data = (T *) calloc((size_t) n1 * n2, sizeof(T));
    a = (T **) malloc(sizeof(T *) * n1);
    for (int i = 0; i < n1; i++) a[i] = data + (size_t) i * n2;
    return a;
  }
  template <typename T> void destroy(T **&a)
  {
    if (!a) return;
    free(a[0]);
    free(a);
    a = nullptr;
  }
};

class Atom {
 public:
  int ntypes = 1, nlocal = 0, nghost = 0;
  int nmax = 0;    // rows the per-atom arrays are allocated for (Atom::nmax)
  double **x = nullptr, **f = nullptr, **torque = nullptr, **v = nullptr, **angmom = nullptr;
  int *type = nullptr, *mask = nullptr;
  int firstgroup = -1, nfirst = 0;
  int **iarray = nullptr;
  int **ivector = nullptr;      // custom per-atom int vectors
  std::map<std::string, void *> extractable;
  std::vector<std::string> custom_names;
  std::vector<int> custom_flag, custom_cols;
  void *extract(const char *name)
  {
    auto it = extractable.find(name);
    return it == extractable.end() ? nullptr : it->second;
  }
#if SHPAIR_STUB_GEN >= 3
  double ***darray = nullptr;   // custom per-atom double arrays (fix property/atom d2_name N)
  int find_custom(const char *name, int &flag, int &cols)
  {
    for (size_t i = 0; i < custom_names.size(); i++)
      if (custom_names[i] == name) {
        flag = custom_flag[i];
        cols = custom_cols[i];
        return (int) i;
      }
    return -1;
  }
#else
  // before the 2021 custom arrays: vectors only (i_name, d_name), two arguments
  int find_custom(const char *name, int &flag)
  {
    for (size_t i = 0; i < custom_names.size(); i++)
      if (custom_names[i] == name && custom_cols[i] == 0) {
        flag = custom_flag[i];
        return (int) i;
      }
    return -1;
  }
#endif
};

class Pair;
class Force {
 public:
  int newton_pair = 1;
  Pair *pair = nullptr;
  Pair *pair_match(const char *, int) { return pair; }
#if SHPAIR_STUB_GEN <= 1
  void bounds(const char *, int, char *str, int nmax, int &nlo, int &nhi)
  {
    if (strcmp(str, "*") == 0) {
      nlo = 1;
      nhi = nmax;
    } else
      nlo = nhi = atoi(str);
  }
#endif
};

class NeighList {
 public:
  int inum = 0;
  int *ilist = nullptr, *numneigh = nullptr;
  int **firstneigh = nullptr;
};

class Neighbor {
 public:
  bigint lastcall = 0;
  int nrequest = 0;
#if SHPAIR_STUB_GEN >= 4
  void add_request(class Pair *) { nrequest++; }
#else
  int request(void *, int) { return nrequest++; }
#endif
};

// Comm::forward_comm(Pair *) as LAMMPS does it for a pair style with comm_forward > 0: pack the owners' values of the
// send list, unpack them into the ghost rows.  Here one swap whose send list is the owner of every ghost.
class Comm {
 public:
  int me = 0, nprocs = 1;
  std::vector<int> ghost_owner;   // test scaffold: owner row of ghost nlocal + g
  class Atom *atom_for_comm = nullptr;
  int forward_calls = 0;
#if SHPAIR_STUB_GEN >= 4
  inline void forward_comm(class Pair *pair) { do_forward(pair); }
#else
  inline void forward_comm_pair(class Pair *pair) { do_forward(pair); }     // the name before 2022
#endif

 private:
  inline void do_forward(class Pair *pair);
};
class Update {
 public:
  bigint ntimestep = 0;
  double dt = 0.005;
};

class LAMMPS {
 public:
  Memory *memory = new Memory;
  Error *error = new Error;
  Atom *atom = new Atom;
  Force *force = new Force;
  Neighbor *neighbor = new Neighbor;
  Comm *comm = new Comm;
  Update *update = new Update;
  LAMMPS() { comm->atom_for_comm = atom; }
};

class Pointers {
 public:
  explicit Pointers(LAMMPS *p) :
      lmp(p), memory(p->memory), error(p->error), atom(p->atom), force(p->force), neighbor(p->neighbor),
      comm(p->comm), update(p->update)
  {
  }
  virtual ~Pointers() = default;

 protected:
  LAMMPS *lmp;
  Memory *&memory;
  Error *&error;
  Atom *&atom;
  Force *&force;
  Neighbor *&neighbor;
  Comm *&comm;
  Update *&update;
};

#if SHPAIR_STUB_GEN >= 2
namespace utils {
inline void bounds(const char *, int, const char *str, int nmin, int nmax, int &nlo, int &nhi, Error *)
{
  if (strcmp(str, "*") == 0) {
    nlo = nmin;
    nhi = nmax;
  } else
    nlo = nhi = atoi(str);
}
}    // namespace utils
#endif

class Pair : protected Pointers {
 public:
  double eng_vdwl = 0.0, eng_coul = 0.0;
  double virial[6] = {0, 0, 0, 0, 0, 0};
  int allocated = 0;
  int **setflag = nullptr;
  double **cutsq = nullptr;
  int single_enable = 1, restartinfo = 1, no_virial_fdotr_compute = 0, manybody_flag = 0;
  int instance_me = 0;
  NeighList *list = nullptr;
  int eflag_either = 0, eflag_global = 0, eflag_atom = 0, vflag_either = 0, vflag_global = 0, vflag_atom = 0;
  int evflag = 0, vflag_fdotr = 0;
  double *eatom = nullptr, **vatom = nullptr;    // per-atom tallies, (re)sized by ev_init as in LAMMPS
  int maxeatom = 0, maxvatom = 0;

  explicit Pair(LAMMPS *p) : Pointers(p) {}
  virtual void compute(int, int) = 0;
  virtual void settings(int, char **) = 0;
  virtual void coeff(int, char **) = 0;
  virtual void init_style() {}
  virtual double init_one(int, int) { return 0.0; }
  virtual void *extract(const char *, int &) { return nullptr; }
  int comm_forward = 0;    // doubles per atom in forward communication
  virtual int pack_forward_comm(int, int *, double *, int, int *) { return 0; }
  virtual void unpack_forward_comm(int, int, double *) {}
#if SHPAIR_STUB_GEN >= 1
  void ev_init(int eflag, int vflag) { ev_setup_impl(eflag, vflag); }
#else
  void ev_setup(int eflag, int vflag) { ev_setup_impl(eflag, vflag); }     // before 2019: called only when a flag is set
#endif

 private:
  void ev_setup_impl(int eflag, int vflag)
  {
    // LAMMPS bit convention: 1 = global, 2 = per-atom
    eflag_either = eflag ? 1 : 0;
    vflag_either = vflag ? 1 : 0;
    eflag_global = eflag & 1;
    vflag_global = vflag & 1;
    eflag_atom = (eflag & 2) ? 1 : 0;
    vflag_atom = (vflag & 2) ? 1 : 0;
    evflag = eflag || vflag;
    const int nall = atom->nlocal + atom->nghost;
    if (eflag_atom) {
      if (nall > maxeatom) {
        free(eatom);
        eatom = (double *) malloc(sizeof(double) * nall);
        maxeatom = nall;
      }
      for (int i = 0; i < nall; i++) eatom[i] = 0.0;
    }
    if (vflag_atom) {
      if (nall > maxvatom) {
        memory->destroy(vatom);
        memory->create(vatom, nall, 6, "pair:vatom");
        maxvatom = nall;
      }
      for (int i = 0; i < nall; i++)
        for (int a = 0; a < 6; a++) vatom[i][a] = 0.0;
    }
    eng_vdwl = eng_coul = 0.0;
    for (double &v : virial) v = 0.0;
  }

 public:
};

inline void Comm::do_forward(Pair *pair)
{
  ++forward_calls;
  const int ng = (int) ghost_owner.size();
  if (ng == 0 || pair->comm_forward <= 0) return;
  std::vector<double> buf((size_t) ng * pair->comm_forward);
  int pbc[6] = {0, 0, 0, 0, 0, 0};
  const int n = pair->pack_forward_comm(ng, ghost_owner.data(), buf.data(), 0, pbc);
  if (n != ng * pair->comm_forward) {
    fprintf(stderr, "stub Comm: pack_forward_comm returned %d, expected %d\n", n, ng * pair->comm_forward);
    exit(1);
  }
  pair->unpack_forward_comm(ng, atom_for_comm ? atom_for_comm->nlocal : 0, buf.data());
}

namespace FixConst {
enum { INITIAL_INTEGRATE = 1 << 0, POST_FORCE = 1 << 4, FINAL_INTEGRATE = 1 << 5 };
}

class Fix : protected Pointers {
 public:
  int igroup = 0, groupbit = 1;
  int time_integrate = 0;
  Fix(LAMMPS *p, int, char **) : Pointers(p) {}
  virtual int setmask() = 0;
  virtual void init() {}
  virtual void initial_integrate(int) {}
  virtual void final_integrate() {}
  virtual void reset_dt() {}
};

}    // namespace LAMMPS_NS
