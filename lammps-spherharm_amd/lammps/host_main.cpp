// host_main.cpp — TEST SCAFFOLD: a minimal C++ host that drives the PairSH
// adapter the way LAMMPS would (pair_style -> pair_coeff -> init -> compute)
// on a bed read from a plain text file, and writes forces/torques back.
// Built against stub/ (no LAMMPS in this image); used by
// tests/test_lammps_adapter.py on the GPU box.  Not part of the product.
//
// bed file:  nlocal nghost ntypes newton eflag
//            per atom: x y z qw qx qy qz type shtype
//            inum ; per row: i n j1 .. jn
// usage: lammps_host <bed> <out> <nq> <kn> <exponent> <shape files...>
// With LAMMPS_HOST_GHOST_OWNERS=<file of nghost owner rows> the quaternions are registered as a CUSTOM per-atom
// array (fix property/atom d2_quat 4 ghost yes) instead of an atom-style array and the stub Comm forwards owners'
// values to the ghosts when the pair style asks for it: the bed file may then hold stale ghost orientations.
// With LAMMPS_HOST_SHTYPE_CUSTOM=1 the shape index is a custom integer vector (fix property/atom i_shtype).
// With LAMMPS_HOST_NSTEPS=<n> and LAMMPS_HOST_DT=<dt> in the environment it then runs n velocity-Verlet
// steps with FixNVESH (fix nve/sh) from rest, the way Verlet::run orders them, and writes
// x v quat angmom of the owned atoms to <out>.traj (ghost-free beds only: nghost = 0).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "fix_nve_sh.h"
#include "pair_sh.h"

using namespace LAMMPS_NS;

int main(int argc, char **argv)
{
  if (argc < 7) {
    fprintf(stderr, "usage: %s bed out nq kn exponent shapes...\n", argv[0]);
    return 2;
  }
  FILE *fp = fopen(argv[1], "r");
  if (!fp) return 3;
  int nlocal, nghost, ntypes, newton, eflag;
  if (fscanf(fp, "%d %d %d %d %d", &nlocal, &nghost, &ntypes, &newton, &eflag) != 5) return 3;
  const int nall = nlocal + nghost;
  LAMMPS lmp;
  Memory mem;
  double **x, **quat, **f, **tq;
  mem.create(x, nall, 3, "x");
  mem.create(quat, nall, 4, "quat");
  mem.create(f, nall, 3, "f");
  mem.create(tq, nall, 3, "torque");
  std::vector<int> type(nall), shtype(nall);
  for (int i = 0; i < nall; i++)
    if (fscanf(fp, "%lf %lf %lf %lf %lf %lf %lf %d %d", &x[i][0], &x[i][1], &x[i][2], &quat[i][0], &quat[i][1],
               &quat[i][2], &quat[i][3], &type[i], &shtype[i]) != 9)
      return 3;
  int inum;
  if (fscanf(fp, "%d", &inum) != 1) return 3;
  std::vector<int> ilist(inum), numneigh(nall, 0);
  std::vector<std::vector<int>> rows(nall);
  std::vector<int *> firstneigh(nall, nullptr);
  for (int ii = 0; ii < inum; ii++) {
    int i, n;
    if (fscanf(fp, "%d %d", &i, &n) != 2) return 3;
    ilist[ii] = i;
    numneigh[i] = n;
    rows[i].resize(n);
    for (int k = 0; k < n; k++)
      if (fscanf(fp, "%d", &rows[i][k]) != 1) return 3;
    firstneigh[i] = rows[i].data();
  }
  fclose(fp);

  Atom *atom = lmp.atom;
  atom->ntypes = ntypes;
  atom->nlocal = nlocal;
  atom->nmax = nall;
  atom->nghost = nghost;
  atom->x = x;
  atom->f = f;
  atom->torque = tq;
  atom->type = type.data();
  int *ivec[1] = {shtype.data()};
#if SHPAIR_STUB_GEN >= 3
  double **darr[1] = {quat};
#endif
  if (const char *gof = getenv("LAMMPS_HOST_GHOST_OWNERS")) {
#if SHPAIR_STUB_GEN < 3
    (void) gof;
    fprintf(stderr, "this LAMMPS generation has no 2-d custom per-atom arrays: quaternions cannot be a custom property\n");
    return 6;
#else
    FILE *gp = fopen(gof, "r");
    if (!gp) return 3;
    lmp.comm->ghost_owner.resize(nghost);
    for (int g = 0; g < nghost; g++)
      if (fscanf(gp, "%d", &lmp.comm->ghost_owner[g]) != 1) return 3;
    fclose(gp);
    atom->custom_names.push_back("quat");
    atom->custom_flag.push_back(1);
    atom->custom_cols.push_back(4);
    atom->darray = darr;
#endif
  } else {
    atom->extractable["quat"] = (void *) quat;        // as atom_style spherharm would expose them
  }
  if (getenv("LAMMPS_HOST_SHTYPE_CUSTOM")) {          // fix property/atom i_shtype: a custom integer vector
    atom->custom_names.push_back("shtype");
    atom->custom_flag.push_back(0);
    atom->custom_cols.push_back(0);
    atom->ivector = ivec;
    if (atom->custom_names.size() != 1) return 6;     // (index 0 of ivector: not together with the custom quaternions)
  } else {
    atom->extractable["shtype"] = (void *) shtype.data();
  }
  lmp.force->newton_pair = newton;

  NeighList list;
  list.inum = inum;
  list.ilist = ilist.data();
  list.numneigh = numneigh.data();
  list.firstneigh = firstneigh.data();
  lmp.neighbor->lastcall = 1;

  PairSH pair(&lmp);
  pair.list = &list;
  std::vector<char *> sargs;
  std::string kw = "shapes";
  sargs.push_back(argv[3]);
  sargs.push_back(kw.data());
  for (int a = 6; a < argc; a++) sargs.push_back(argv[a]);
  pair.settings((int) sargs.size(), sargs.data());
  std::string star = "*";
  char *cargs[4] = {star.data(), star.data(), argv[4], argv[5]};
  pair.coeff(4, cargs);
  pair.init_style();
  double cut = 0.0;
  for (int i = 1; i <= ntypes; i++)
    for (int j = i; j <= ntypes; j++) cut = pair.init_one(i, j);

  fprintf(stderr, "lammps_host: stub generation %d\n", SHPAIR_STUB_GEN);
  pair.compute(eflag ? 3 : 0, eflag ? 3 : 0);    // step 1: global and per-atom tallies
  const double e1 = pair.eng_vdwl;
  std::vector<double> ea(nall, 0.0), va(6 * (size_t) nall, 0.0);
  if (eflag) {
    for (int i = 0; i < nall; i++) {
      ea[i] = pair.eatom[i];
      for (int a = 0; a < 6; a++) va[6 * (size_t) i + a] = pair.vatom[i][a];
    }
  }
  // step 2 reuses the uploaded neighbour list (lastcall unchanged); LAMMPS clears forces in between
  for (int i = 0; i < nall; i++)
    for (int a = 0; a < 3; a++) f[i][a] = tq[i][a] = 0.0;
  pair.compute(eflag, eflag);

  fp = fopen(argv[2], "w");
  if (!fp) return 4;
  fprintf(fp, "%.17g %.17g %.17g %d\n", cut, e1, pair.eng_vdwl, lmp.comm->forward_calls);
  fprintf(fp, "%.17g %.17g %.17g %.17g %.17g %.17g\n", pair.virial[0], pair.virial[1], pair.virial[2],
          pair.virial[3], pair.virial[4], pair.virial[5]);
  for (int i = 0; i < nall; i++)
    fprintf(fp, "%.17g %.17g %.17g %.17g %.17g %.17g\n", f[i][0], f[i][1], f[i][2], tq[i][0], tq[i][1], tq[i][2]);
  fclose(fp);
  if (eflag) {    // per-atom tallies of step 1
    std::string pa = std::string(argv[2]) + ".peratom";
    fp = fopen(pa.c_str(), "w");
    if (!fp) return 4;
    for (int i = 0; i < nall; i++)
      fprintf(fp, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", ea[i], va[6 * (size_t) i], va[6 * (size_t) i + 1],
              va[6 * (size_t) i + 2], va[6 * (size_t) i + 3], va[6 * (size_t) i + 4], va[6 * (size_t) i + 5]);
    fclose(fp);
  }

  const char *ns = getenv("LAMMPS_HOST_NSTEPS");
  if (ns && atoi(ns) > 0 && nghost == 0) {
    const int nsteps = atoi(ns);
    const char *dts = getenv("LAMMPS_HOST_DT");
    lmp.update->dt = dts ? atof(dts) : 1e-3;
    double **v, **angmom;
    mem.create(v, nall, 3, "v");
    mem.create(angmom, nall, 3, "angmom");
    std::vector<int> mask(nall, 1);
    atom->v = v;
    atom->angmom = angmom;
    atom->mask = mask.data();
    lmp.force->pair = &pair;
    std::string a0 = "1", a1 = "all", a2 = "nve/sh", a3 = "density", a4 = "1.5";
    char *fargs[5] = {a0.data(), a1.data(), a2.data(), a3.data(), a4.data()};
    FixNVESH fix(&lmp, 5, fargs);
    if (fix.setmask() != (FixConst::INITIAL_INTEGRATE | FixConst::FINAL_INTEGRATE)) return 5;
    fix.init();
    // f, tq hold the forces of the current positions (second compute above)
    for (int step = 0; step < nsteps; step++) {
      fix.initial_integrate(0);
      for (int i = 0; i < nall; i++)
        for (int a = 0; a < 3; a++) f[i][a] = tq[i][a] = 0.0;
      pair.compute(0, 0);
      fix.final_integrate();
    }
    std::string tr = std::string(argv[2]) + ".traj";
    fp = fopen(tr.c_str(), "w");
    if (!fp) return 4;
    for (int i = 0; i < nlocal; i++)
      fprintf(fp, "%.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", x[i][0], x[i][1],
              x[i][2], v[i][0], v[i][1], v[i][2], quat[i][0], quat[i][1], quat[i][2], quat[i][3], angmom[i][0],
              angmom[i][1], angmom[i][2]);
    fclose(fp);
  }
  return 0;
}
