// shstep_api.hip — the C ABI of include/shstep.h (docs/SPEC.md Part II) on top of step_kernels.hpp.
// Host side: per-shape rigid-body table, box / bin geometry, buffer ownership, the blocking read-backs
// (ghost count, pair count, rebuild flag).  No CPU fallback: every entry point launches gfx950 kernels.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/shstep.h"
#include "sh_tables.hpp"
#include "shpair_ctx.hpp"
#include "step_kernels.hpp"

using namespace shp;

struct shstep_state {
  BoxParams box{};
  bool have_box = false;
  double skin = 0.0;

  DevBuf<double> d_mass;  // kMassStride doubles per shape
  std::vector<double> h_mass;

  DevBuf<int> d_flags;     // [0] error bits, [1] moved flag
  int* h_flags = nullptr;  // pinned, 4 ints

  // borders
  DevBuf<int> d_cnt, d_goff, d_sums, d_gowner, d_gcode;
  int b_nlocal = 0, nghost = 0;
  // bins + list
  DevBuf<int> d_cell, d_cellcount, d_cellstart, d_atoms, d_nn, d_offs;
  DevBuf<int> d_part_i, d_part_j, d_part_scan;   // "halo_overlap": the row-major list (kept for shstep_copy_neighbors) / scan scratch
  bool partitioned = false;
  DevBuf<double> d_xhold;
  int l_nlocal = -1;

  // staging of the host-pointer integrator
  DevBuf<double> s_x, s_v, s_q, s_L, s_f, s_t;
  DevBuf<int> s_sh, s_mask;

  void release()
  {
    d_mass.release(); d_flags.release(); d_cnt.release(); d_goff.release(); d_sums.release(); d_gowner.release();
    d_gcode.release(); d_cell.release(); d_cellcount.release(); d_cellstart.release(); d_atoms.release();
    d_nn.release(); d_offs.release(); d_part_i.release(); d_part_j.release(); d_part_scan.release(); d_xhold.release(); s_x.release(); s_v.release(); s_q.release();
    s_L.release(); s_f.release(); s_t.release(); s_sh.release(); s_mask.release();
    if (h_flags) (void)hipHostFree(h_flags);
    h_flags = nullptr;
  }
};

// called by shpair_destroy (shpair_api.hip)
void shstep_release_state(shpair_ctx* c)
{
  if (!c || !c->step) return;
  c->step->release();
  delete c->step;
  c->step = nullptr;
}

// a host-supplied list replaced the device-built one (shpair_api.hip)
void shstep_invalidate_list(shpair_ctx* c)
{
  if (c && c->step) c->step->l_nlocal = -1;
}

static inline unsigned nblk(long long n, int b) { return (unsigned)((n + b - 1) / b > 0 ? (n + b - 1) / b : 1); }

static int get_state(shpair_ctx* c, shstep_state** out)
{
  if (!c->step) {
    shstep_state* s = new (std::nothrow) shstep_state();
    if (!s) CTX_FAIL(c, SHPAIR_ENOMEM, "out of host memory");
    c->step = s;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, s->d_flags.ensure(4));
    HIPCHK(c, hipMemset(s->d_flags.p, 0, 4 * sizeof(int)));
    HIPCHK(c, hipHostMalloc((void**)&s->h_flags, 4 * sizeof(int)));
  }
  *out = c->step;
  return SHPAIR_OK;
}

// Per-shape rows: m, 1/m, c[3], Iinv[6], rmax.  Rebuilt when shapes or densities changed.
static int refresh_mass(shpair_ctx* c, shstep_state* s)
{
  if (c->nshapes <= 0) CTX_FAIL(c, SHPAIR_ESTATE, "shapes are not set");
  if (!c->mass_dirty && s->d_mass.p) return SHPAIR_OK;
  s->h_mass.assign((size_t)kMassStride * c->nshapes, 0.0);
  for (int k = 0; k < c->nshapes; ++k) {
    const Shape& sh = c->shapes[k];
    if (sh.lmax < 0) CTX_FAIL(c, SHPAIR_ESTATE, "shape %d is not set", k);
    double mp[10], inv[6];
    mass_props(sh.lmax, sh.anm.data(), mp);
    if (!(mp[0] > 0.0) || !inertia_inverse(mp, sh.density, inv))
      CTX_FAIL(c, SHPAIR_EINVAL, "shape %d: volume %g / inertia tensor is not positive (is r > 0 everywhere?)", k, mp[0]);
    double* r = &s->h_mass[(size_t)kMassStride * k];
    r[0] = sh.density * mp[0];
    r[1] = 1.0 / r[0];
    r[2] = mp[1]; r[3] = mp[2]; r[4] = mp[3];
    for (int q = 0; q < 6; ++q) r[5 + q] = inv[q];
    r[11] = sh.rmax;
  }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());  // an enqueued kernel may still read the old table
  HIPCHK(c, s->d_mass.ensure(s->h_mass.size()));
  HIPCHK(c, hipMemcpy(s->d_mass.p, s->h_mass.data(), s->h_mass.size() * sizeof(double), hipMemcpyHostToDevice));
  c->mass_dirty = false;
  return SHPAIR_OK;
}

// cmax, bin grid. Needs shapes (bounding radii).
static int refresh_box(shpair_ctx* c, shstep_state* s)
{
  if (!s->have_box) CTX_FAIL(c, SHPAIR_ESTATE, "shstep_set_box() must come first");
  double rm = 0.0;
  for (int k = 0; k < c->nshapes; ++k) {
    if (c->shapes[k].lmax < 0) CTX_FAIL(c, SHPAIR_ESTATE, "shape %d is not set", k);
    rm = std::fmax(rm, c->shapes[k].rmax);
  }
  if (!(rm > 0.0)) CTX_FAIL(c, SHPAIR_ESTATE, "shapes are not set");
  BoxParams& b = s->box;
  b.cmax = 2.0 * rm + s->skin;
  double ncell = 1.0;
  for (int d = 0; d < 3; ++d) {
    if (b.periodic[d] && b.len[d] < 2.0 * b.cmax)
      CTX_FAIL(c, SHPAIR_EINVAL, "periodic box edge %d (%g) is shorter than twice the ghost cutoff (%g)", d, b.len[d], b.cmax);
    b.glo[d] = b.periodic[d] ? b.lo[d] - b.cmax : b.lo[d];
    const double ext = b.periodic[d] ? b.len[d] + 2.0 * b.cmax : b.len[d];
    double n = std::floor(ext / b.cmax);
    if (!(n >= 1.0)) n = 1.0;
    if (n > 1024.0) n = 1024.0;  // <= 2^30 cells in all; further capped below
    b.nc[d] = (int)n;
    ncell *= n;
  }
  while (ncell > 67108864.0) {  // 2^26 cells: coarsen the longest direction
    int d = 0;
    for (int k = 1; k < 3; ++k)
      if (b.nc[k] > b.nc[d]) d = k;
    ncell /= b.nc[d];
    b.nc[d] = (b.nc[d] + 1) / 2;
    ncell *= b.nc[d];
  }
  for (int d = 0; d < 3; ++d) {
    const double ext = b.periodic[d] ? b.len[d] + 2.0 * b.cmax : b.len[d];
    b.binv[d] = b.nc[d] / ext;
  }
  return SHPAIR_OK;
}

static int exclusive_scan(shpair_ctx* c, shstep_state* s, const int* in, int* out, int n, hipStream_t st)
{
  const unsigned nb = nblk(n, kScanBlock);
  HIPCHK(c, s->d_sums.ensure(nb + 1));
  hipLaunchKernelGGL(scan_local_kernel, dim3(nb), dim3(kScanBlock), 0, st, in, n, out, s->d_sums.p);
  hipLaunchKernelGGL(scan_sums_kernel, dim3(1), dim3(kScanBlock), 0, st, s->d_sums.p, (int)nb);
  hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(kScanBlock), 0, st, out, n, (const int*)s->d_sums.p, (int)nb);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

// reads and clears the device error bits; stream must be idle
static int check_device_flags(shpair_ctx* c, shstep_state* s, hipStream_t st)
{
  HIPCHK(c, hipMemcpyAsync(s->h_flags, s->d_flags.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (s->h_flags[0]) {
    const int bits = s->h_flags[0];
    HIPCHK(c, hipMemsetAsync(s->d_flags.p, 0, sizeof(int), st));
    if (bits & kErrShape) CTX_FAIL(c, SHPAIR_EINVAL, "a shape index (shtype) outside [0,%d) reached a kernel; those particles were skipped", c->nshapes);
  }
  return SHPAIR_OK;
}

#define STEP_PROLOGUE(c)                     \
  if (!(c)) return SHPAIR_EINVAL;            \
  shstep_state* s = nullptr;                 \
  {                                          \
    const int _rc = get_state((c), &s);      \
    if (_rc) return _rc;                     \
  }                                          \
  HIPCHK((c), hipSetDevice((c)->device))

#define RC(call)            \
  do {                      \
    const int _rc = (call); \
    if (_rc) return _rc;    \
  } while (0)

extern "C" {

int shstep_shape_mass_props(int lmax, const double* anm, double* out)
{
  if (lmax < 0 || lmax > SHPAIR_MAX_LMAX || !anm || !out) return SHPAIR_EINVAL;
  mass_props(lmax, anm, out);
  return SHPAIR_OK;
}

int shstep_set_density(shpair_ctx* c, int ishape, double rho)
{
  if (!c) return SHPAIR_EINVAL;
  if (ishape < 0 || ishape >= c->nshapes || c->shapes[ishape].lmax < 0)
    CTX_FAIL(c, SHPAIR_EINVAL, "density: shape %d is not set", ishape);
  if (!(rho > 0.0) || !std::isfinite(rho)) CTX_FAIL(c, SHPAIR_EINVAL, "density %g must be finite and > 0", rho);
  c->shapes[ishape].density = rho;
  c->mass_dirty = true;
  return SHPAIR_OK;
}

int shstep_get_body(const shpair_ctx* c, int ishape, double* mass, double* com, double* inertia)
{
  if (!c) return SHPAIR_EINVAL;
  if (ishape < 0 || ishape >= c->nshapes || c->shapes[ishape].lmax < 0) return SHPAIR_EINVAL;
  const Shape& sh = c->shapes[ishape];
  double mp[10];
  mass_props(sh.lmax, sh.anm.data(), mp);
  if (mass) *mass = sh.density * mp[0];
  if (com)
    for (int k = 0; k < 3; ++k) com[k] = mp[1 + k];
  if (inertia)
    for (int k = 0; k < 6; ++k) inertia[k] = sh.density * mp[4 + k];
  return SHPAIR_OK;
}

int shstep_nve_device(shpair_ctx* c, int phase, int nlocal, double dt, double* x, double* v, double* quat,
                      double* angmom, const double* f, const double* torque, const int* shtype, const int* mask,
                      int groupbit, void* stream)
{
  STEP_PROLOGUE(c);
  if (phase != 0 && phase != 1) CTX_FAIL(c, SHPAIR_EINVAL, "phase %d is neither 0 (initial) nor 1 (final)", phase);
  if (nlocal < 0 || !std::isfinite(dt)) CTX_FAIL(c, SHPAIR_EINVAL, "bad nlocal (%d) or dt (%g)", nlocal, dt);
  if (nlocal == 0) return SHPAIR_OK;
  if (!x || !v || !quat || !angmom || !f || !torque || !shtype || !mask) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  RC(refresh_mass(c, s));
  hipStream_t st = (hipStream_t)stream;
  if (phase == 0)
    hipLaunchKernelGGL(nve_kernel<0>, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, dt,
                       (const double*)s->d_mass.p, c->nshapes, x, v, quat, angmom, f, torque, shtype, mask, groupbit,
                       s->d_flags.p);
  else
    hipLaunchKernelGGL(nve_kernel<1>, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, dt,
                       (const double*)s->d_mass.p, c->nshapes, x, v, quat, angmom, f, torque, shtype, mask, groupbit,
                       s->d_flags.p);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

int shstep_nve(shpair_ctx* c, int phase, int nlocal, double dt, double* x, double* v, double* quat, double* angmom,
               const double* f, const double* torque, const int* shtype, const int* mask, int groupbit)
{
  STEP_PROLOGUE(c);
  if (nlocal < 0) CTX_FAIL(c, SHPAIR_EINVAL, "nlocal %d < 0", nlocal);
  if (nlocal == 0) return SHPAIR_OK;
  if (!x || !v || !quat || !angmom || !f || !torque || !shtype || !mask) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  for (int i = 0; i < nlocal; ++i)
    if ((mask[i] & groupbit) && (shtype[i] < 0 || shtype[i] >= c->nshapes))
      CTX_FAIL(c, SHPAIR_EINVAL, "shtype[%d] = %d outside [0,%d)", i, shtype[i], c->nshapes);
  const size_t n = (size_t)nlocal;
  HIPCHK(c, s->s_x.ensure(3 * n)); HIPCHK(c, s->s_v.ensure(3 * n)); HIPCHK(c, s->s_q.ensure(4 * n));
  HIPCHK(c, s->s_L.ensure(3 * n)); HIPCHK(c, s->s_f.ensure(3 * n)); HIPCHK(c, s->s_t.ensure(3 * n));
  HIPCHK(c, s->s_sh.ensure(n)); HIPCHK(c, s->s_mask.ensure(n));
  hipStream_t st = c->stream;
  HIPCHK(c, hipMemcpyAsync(s->s_v.p, v, 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->s_q.p, quat, 4 * n * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->s_L.p, angmom, 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->s_f.p, f, 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->s_t.p, torque, 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->s_sh.p, shtype, n * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(s->s_mask.p, mask, n * sizeof(int), hipMemcpyHostToDevice, st));
  if (phase == 0) HIPCHK(c, hipMemcpyAsync(s->s_x.p, x, 3 * n * sizeof(double), hipMemcpyHostToDevice, st));
  RC(shstep_nve_device(c, phase, nlocal, dt, s->s_x.p, s->s_v.p, s->s_q.p, s->s_L.p, s->s_f.p, s->s_t.p, s->s_sh.p,
                       s->s_mask.p, groupbit, st));
  HIPCHK(c, hipMemcpyAsync(v, s->s_v.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(angmom, s->s_L.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, st));
  if (phase == 0) {
    HIPCHK(c, hipMemcpyAsync(x, s->s_x.p, 3 * n * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipMemcpyAsync(quat, s->s_q.p, 4 * n * sizeof(double), hipMemcpyDeviceToHost, st));
  }
  HIPCHK(c, hipStreamSynchronize(st));
  return SHPAIR_OK;
}

int shstep_force_clear_device(shpair_ctx* c, int nall, double* f, double* torque, void* stream)
{
  if (!c) return SHPAIR_EINVAL;
  if (nall < 0) CTX_FAIL(c, SHPAIR_EINVAL, "negative atom count");
  if (nall == 0) return SHPAIR_OK;
  if (!f || !torque) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  HIPCHK(c, hipSetDevice(c->device));
  const long long n3 = 3LL * nall;
  long long blocks = (n3 + kStepBlock - 1) / kStepBlock;
  if (blocks > 4096) blocks = 4096;   // grid-stride: 16 workgroups per CU are plenty for a store-only kernel
  hipLaunchKernelGGL(force_clear_kernel, dim3((unsigned)blocks), dim3(kStepBlock), 0, (hipStream_t)stream, n3, f, torque);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

int shstep_post_force_device(shpair_ctx* c, int nlocal, const double* g, double gamma_t, double gamma_r, const double* v,
                             const double* quat, const double* angmom, const int* shtype, const int* mask, int groupbit,
                             double* f, double* torque, void* stream)
{
  STEP_PROLOGUE(c);
  if (nlocal < 0 || !g) CTX_FAIL(c, SHPAIR_EINVAL, "bad nlocal (%d) or null gravity", nlocal);
  if (!std::isfinite(g[0] + g[1] + g[2] + gamma_t + gamma_r)) CTX_FAIL(c, SHPAIR_EINVAL, "gravity / damping is not finite");
  if (nlocal == 0) return SHPAIR_OK;
  if (!v || !quat || !angmom || !f || !torque || !shtype || !mask) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  RC(refresh_mass(c, s));
  hipLaunchKernelGGL(post_force_kernel, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, (hipStream_t)stream, nlocal,
                     (const double*)s->d_mass.p, c->nshapes, g[0], g[1], g[2], gamma_t, gamma_r, v, quat, angmom, shtype,
                     mask, groupbit, f, torque, s->d_flags.p);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

int shstep_energies_device(shpair_ctx* c, int nlocal, const double* g, const double* x, const double* v, const double* quat,
                           const double* angmom, const int* shtype, const int* mask, int groupbit, double* out3,
                           void* stream)
{
  STEP_PROLOGUE(c);
  if (nlocal < 0 || !g || !out3) CTX_FAIL(c, SHPAIR_EINVAL, "bad nlocal (%d), null gravity or null output", nlocal);
  if (nlocal == 0) return SHPAIR_OK;
  if (!x || !v || !quat || !angmom || !shtype || !mask) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  RC(refresh_mass(c, s));
  hipLaunchKernelGGL(energies_kernel, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, (hipStream_t)stream, nlocal,
                     (const double*)s->d_mass.p, c->nshapes, g[0], g[1], g[2], x, v, quat, angmom, shtype, mask, groupbit,
                     out3, s->d_flags.p);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

int shstep_set_box(shpair_ctx* c, const double* lo, const double* hi, const int* periodic, double skin)
{
  STEP_PROLOGUE(c);
  if (!lo || !hi || !periodic) CTX_FAIL(c, SHPAIR_EINVAL, "null box pointer");
  if (!(skin >= 0.0) || !std::isfinite(skin)) CTX_FAIL(c, SHPAIR_EINVAL, "skin %g must be finite and >= 0", skin);
  for (int d = 0; d < 3; ++d)
    if (!std::isfinite(lo[d]) || !std::isfinite(hi[d]) || !(hi[d] > lo[d]))
      CTX_FAIL(c, SHPAIR_EINVAL, "box dimension %d: [%g, %g) is empty or not finite", d, lo[d], hi[d]);
  for (int d = 0; d < 3; ++d) {
    s->box.lo[d] = lo[d];
    s->box.hi[d] = hi[d];
    s->box.len[d] = hi[d] - lo[d];
    s->box.periodic[d] = periodic[d] ? 1 : 0;
  }
  s->skin = skin;
  s->have_box = true;
  s->l_nlocal = -1;
  s->nghost = 0;
  s->b_nlocal = 0;
  return SHPAIR_OK;
}

int shstep_borders_device(shpair_ctx* c, int nlocal, int nmax, double* x, double* quat, int* type, int* shtype, int* tag,
                          int* nghost, void* stream)
{
  STEP_PROLOGUE(c);
  if (nghost) *nghost = 0;
  if (nlocal < 0 || nmax < nlocal || !nghost) CTX_FAIL(c, SHPAIR_EINVAL, "bad nlocal (%d) / nmax (%d) / null nghost", nlocal, nmax);
  RC(refresh_box(c, s));
  s->nghost = 0;
  s->b_nlocal = nlocal;
  if (nlocal == 0) return SHPAIR_OK;
  if (!x || !quat || !type || !shtype) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, s->d_cnt.ensure((size_t)nlocal));
  HIPCHK(c, s->d_goff.ensure((size_t)nlocal + 1));
  hipLaunchKernelGGL(wrap_count_kernel, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, s->box, x, s->d_cnt.p);
  RC(exclusive_scan(c, s, s->d_cnt.p, s->d_goff.p, nlocal, st));
  HIPCHK(c, hipMemcpyAsync(s->h_flags + 2, s->d_goff.p + nlocal, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const int ng = s->h_flags[2];
  *nghost = ng;
  if ((long long)nlocal + ng > nmax)
    CTX_FAIL(c, SHPAIR_ENOMEM, "%d owned + %d ghost particles exceed the caller's capacity nmax = %d", nlocal, ng, nmax);
  if (ng > 0) {
    HIPCHK(c, s->d_gowner.ensure((size_t)ng));
    HIPCHK(c, s->d_gcode.ensure((size_t)ng));
    hipLaunchKernelGGL(fill_ghosts_kernel, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, nmax, s->box,
                       (const int*)s->d_goff.p, x, quat, type, shtype, tag, s->d_gowner.p, s->d_gcode.p);
    HIPCHK(c, hipGetLastError());
  }
  s->nghost = ng;
  return SHPAIR_OK;
}

int shstep_forward_device(shpair_ctx* c, double* x, double* quat, void* stream)
{
  STEP_PROLOGUE(c);
  if (s->nghost == 0) return SHPAIR_OK;
  if (!x || !quat) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  hipLaunchKernelGGL(forward_kernel, dim3(nblk(s->nghost, kStepBlock)), dim3(kStepBlock), 0, (hipStream_t)stream, s->b_nlocal,
                     s->nghost, s->box, (const int*)s->d_gowner.p, (const int*)s->d_gcode.p, x, quat);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

int shstep_reverse_device(shpair_ctx* c, double* f, double* torque, void* stream)
{
  STEP_PROLOGUE(c);
  if (s->nghost == 0) return SHPAIR_OK;
  if (!f || !torque) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  hipLaunchKernelGGL(reverse_kernel, dim3(nblk(s->nghost, kStepBlock)), dim3(kStepBlock), 0, (hipStream_t)stream, s->b_nlocal,
                     s->nghost, (const int*)s->d_gowner.p, f, torque);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

int shstep_neighbor_build_device(shpair_ctx* c, int nlocal, int nghost, const double* x, const int* shtype, const int* tag,
                                 int* npairs, void* stream)
{
  STEP_PROLOGUE(c);
  if (npairs) *npairs = 0;
  if (nlocal < 0 || nghost < 0 || !npairs) CTX_FAIL(c, SHPAIR_EINVAL, "bad nlocal (%d) / nghost (%d) / null npairs", nlocal, nghost);
  RC(refresh_box(c, s));
  RC(refresh_mass(c, s));
  if (!tag && nghost > 0 && (nghost != s->nghost || nlocal != s->b_nlocal))
    CTX_FAIL(c, SHPAIR_ESTATE, "without tags the ghosts must be those of the last shstep_borders_device() (%d owned, %d ghosts)",
             s->b_nlocal, s->nghost);
  hipStream_t st = (hipStream_t)stream;
  const int nall = nlocal + nghost;
  // the previous list may still be in use by an enqueued compute on another stream
  HIPCHK(c, hipDeviceSynchronize());
  c->have_neighbors = false;
  s->l_nlocal = -1;
  if (nlocal == 0) {
    // a rank that lost all its atoms by migration: an empty list, and nothing of the previous list's partition survives
    // (shhalo_run_device cuts its slot ranges at n_interior)
    c->npairs = 0;
    c->n_interior = 0;
    s->partitioned = false;
    c->max_atom_index = nall - 1;
    HIPCHK(c, c->d_pair_i.ensure(1));
    HIPCHK(c, c->d_pair_j.ensure(1));
    HIPCHK(c, s->d_offs.ensure(1));
    HIPCHK(c, hipMemsetAsync(s->d_offs.p, 0, sizeof(int), st));
    c->have_neighbors = true;
    s->l_nlocal = 0;
    return SHPAIR_OK;
  }
  if (!x || !shtype) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  const BoxParams& b = s->box;
  const size_t ncell = (size_t)b.nc[0] * b.nc[1] * b.nc[2];
  HIPCHK(c, s->d_cell.ensure((size_t)nall));
  HIPCHK(c, s->d_atoms.ensure((size_t)nall));
  HIPCHK(c, s->d_cellcount.ensure(ncell));
  HIPCHK(c, s->d_cellstart.ensure(ncell + 1));
  HIPCHK(c, s->d_nn.ensure((size_t)nlocal));
  HIPCHK(c, s->d_offs.ensure((size_t)nlocal + 1));
  HIPCHK(c, s->d_xhold.ensure(3 * (size_t)nlocal));
  HIPCHK(c, hipMemsetAsync(s->d_cellcount.p, 0, ncell * sizeof(int), st));
  hipLaunchKernelGGL(bin_count_kernel, dim3(nblk(nall, kStepBlock)), dim3(kStepBlock), 0, st, nall, b, x, s->d_cell.p,
                     s->d_cellcount.p);
  RC(exclusive_scan(c, s, s->d_cellcount.p, s->d_cellstart.p, (int)ncell, st));
  HIPCHK(c, hipMemsetAsync(s->d_cellcount.p, 0, ncell * sizeof(int), st));  // reused as the fill cursor
  hipLaunchKernelGGL(bin_fill_kernel, dim3(nblk(nall, kStepBlock)), dim3(kStepBlock), 0, st, nall, (const int*)s->d_cell.p,
                     (const int*)s->d_cellstart.p, s->d_cellcount.p, s->d_atoms.p);
  hipLaunchKernelGGL(half_list_kernel<false>, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, nall, b, s->skin,
                     x, shtype, tag, (const int*)s->d_gowner.p, (const double*)s->d_mass.p, c->nshapes,
                     (const int*)s->d_cell.p, (const int*)s->d_cellstart.p, (const int*)s->d_atoms.p, s->d_nn.p,
                     (const int*)nullptr, (int*)nullptr, (int*)nullptr, s->d_flags.p);
  RC(exclusive_scan(c, s, s->d_nn.p, s->d_offs.p, nlocal, st));
  HIPCHK(c, hipMemcpyAsync(s->h_flags + 2, s->d_offs.p + nlocal, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  const int np = s->h_flags[2];
  if (np < 0) CTX_FAIL(c, SHPAIR_EINVAL, "half list too long (pair count overflowed)");
  HIPCHK(c, c->d_pair_i.ensure(np ? (size_t)np : 1));
  HIPCHK(c, c->d_pair_j.ensure(np ? (size_t)np : 1));
  HIPCHK(c, shp_size_pair_buffers(c, (size_t)np));   // per-slot buffers of the pair kernels (shpair_api.hip)
  if (np > 0)
    hipLaunchKernelGGL(half_list_kernel<true>, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, nall, b, s->skin,
                       x, shtype, tag, (const int*)s->d_gowner.p, (const double*)s->d_mass.p, c->nshapes,
                       (const int*)s->d_cell.p, (const int*)s->d_cellstart.p, (const int*)s->d_atoms.p, (int*)nullptr,
                       (const int*)s->d_offs.p, c->d_pair_i.p, c->d_pair_j.p, s->d_flags.p);
  hipLaunchKernelGGL(copy_x_kernel, dim3(nblk(3LL * nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, x, s->d_xhold.p);
  HIPCHK(c, hipGetLastError());
  RC(check_device_flags(c, s, st));
  c->n_interior = np;   // no ghost j, or no partition: every slot may run before the ghosts arrive only if there are none
  s->partitioned = false;
  if (c->opt_overlap && nghost > 0 && np > 0) {
    // interior slots first (stable), ghost-j slots behind them: the context's list becomes the partitioned one, the
    // row-major j list stays in d_part_j for shstep_copy_neighbors
    HIPCHK(c, s->d_part_i.ensure((size_t)np));
    HIPCHK(c, s->d_part_j.ensure((size_t)np));
    HIPCHK(c, s->d_part_scan.ensure(2 * (size_t)np + 2));
    int* flag = s->d_part_scan.p + np + 1;
    hipLaunchKernelGGL(part_flag_kernel, dim3(nblk(np, kStepBlock)), dim3(kStepBlock), 0, st, np, nlocal, (const int*)c->d_pair_j.p, flag);
    RC(exclusive_scan(c, s, flag, s->d_part_scan.p, np, st));
    hipLaunchKernelGGL(part_scatter_kernel, dim3(nblk(np, kStepBlock)), dim3(kStepBlock), 0, st, np, nlocal, (const int*)c->d_pair_i.p,
                       (const int*)c->d_pair_j.p, (const int*)s->d_part_scan.p, s->d_part_i.p, s->d_part_j.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(s->h_flags + 2, s->d_part_scan.p + np, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipStreamSynchronize(st));
    c->n_interior = s->h_flags[2];
    std::swap(c->d_pair_i, s->d_part_i);
    std::swap(c->d_pair_j, s->d_part_j);
    s->partitioned = true;
  } else if (nghost > 0) {
    c->n_interior = 0;   // ghosts but no partition: nothing may run ahead of the forward exchange
  }
  c->npairs = np;
  c->max_atom_index = nall - 1;
  c->have_neighbors = true;
  s->l_nlocal = nlocal;
  *npairs = np;
  return SHPAIR_OK;
}

int shstep_neighbor_check_device(shpair_ctx* c, int nlocal, const double* x, int* rebuild, void* stream)
{
  STEP_PROLOGUE(c);
  if (!rebuild) CTX_FAIL(c, SHPAIR_EINVAL, "null rebuild pointer");
  *rebuild = 1;
  if (s->l_nlocal < 0 || nlocal != s->l_nlocal) return SHPAIR_OK;  // no list, or the particle count changed
  if (nlocal == 0) {
    *rebuild = 0;
    return SHPAIR_OK;
  }
  if (!x) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  hipStream_t st = (hipStream_t)stream;
  const double trig = 0.5 * s->skin;
  HIPCHK(c, hipMemsetAsync(s->d_flags.p + 1, 0, sizeof(int), st));
  hipLaunchKernelGGL(check_distance_kernel, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, x,
                     (const double*)s->d_xhold.p, trig * trig, s->d_flags.p + 1);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipMemcpyAsync(s->h_flags, s->d_flags.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  *rebuild = s->h_flags[1] ? 1 : 0;
  if (s->h_flags[0]) {
    HIPCHK(c, hipMemsetAsync(s->d_flags.p, 0, sizeof(int), st));
    CTX_FAIL(c, SHPAIR_EINVAL, "a shape index (shtype) outside [0,%d) reached a kernel; those particles were skipped", c->nshapes);
  }
  return SHPAIR_OK;
}

}  // extern "C"

// Internal (shpair_ctx.hpp): the three-pass exclusive scan, for the plan builder of shhalo_api.hip.  out[n] = total.
int shstep_exclusive_scan(shpair_ctx* c, const int* in, int* out, int n, void* stream)
{
  STEP_PROLOGUE(c);
  return exclusive_scan(c, s, in, out, n, (hipStream_t)stream);
}

// Internal (shpair_ctx.hpp), for the multi-rank loop of shhalo_api.hip: enqueues the displacement test of
// Neighbor::check_distance and hands back the device flag (1 = an owned row moved more than skin/2) instead of reading
// it, so that the caller can all-reduce it first.  *forced = 1 (nothing enqueued): there is no list for these rows.
int shstep_enqueue_check(shpair_ctx* c, int nlocal, const double* x, int** flag_dev, int* forced, void* stream)
{
  STEP_PROLOGUE(c);
  if (!flag_dev || !forced) CTX_FAIL(c, SHPAIR_EINVAL, "null output pointer");
  HIPCHK(c, s->d_flags.ensure(4));
  *flag_dev = s->d_flags.p + 1;
  *forced = 0;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipMemsetAsync(s->d_flags.p + 1, 0, sizeof(int), st));
  if (s->l_nlocal < 0 || nlocal != s->l_nlocal) {
    *forced = 1;
    return SHPAIR_OK;
  }
  if (nlocal == 0) return SHPAIR_OK;
  if (!x) CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  const double trig = 0.5 * s->skin;
  hipLaunchKernelGGL(check_distance_kernel, dim3(nblk(nlocal, kStepBlock)), dim3(kStepBlock), 0, st, nlocal, x,
                     (const double*)s->d_xhold.p, trig * trig, s->d_flags.p + 1);
  HIPCHK(c, hipGetLastError());
  return SHPAIR_OK;
}

extern "C" {

int shstep_copy_neighbors(shpair_ctx* c, int* offsets, int* jlist)
{
  STEP_PROLOGUE(c);
  if (s->l_nlocal < 0 || !c->have_neighbors) CTX_FAIL(c, SHPAIR_ESTATE, "no device-built neighbour list");
  HIPCHK(c, hipDeviceSynchronize());
  if (offsets) HIPCHK(c, hipMemcpy(offsets, s->d_offs.p, ((size_t)s->l_nlocal + 1) * sizeof(int), hipMemcpyDeviceToHost));
  // (with "halo_overlap" the context's list is partitioned interior / boundary; the row-major copy is kept beside it)
  if (jlist && c->npairs > 0)
    HIPCHK(c, hipMemcpy(jlist, s->partitioned ? s->d_part_j.p : c->d_pair_j.p, (size_t)c->npairs * sizeof(int), hipMemcpyDeviceToHost));
  return SHPAIR_OK;
}

}  // extern "C"

namespace {
struct StepGraphs {
  hipGraphExec_t a = nullptr, b = nullptr;
  void reset()
  {
    if (a) (void)hipGraphExecDestroy(a);
    if (b) (void)hipGraphExecDestroy(b);
    a = b = nullptr;
  }
};
}  // namespace

// segment A of a step: half kick + drift, and (when asked) the displacement test with its flag read-back
static int enqueue_a(shpair_ctx* c, shstep_state* s, const shstep_arrays* a, bool with_check, hipStream_t st)
{
  RC(shstep_nve_device(c, 0, a->nlocal, a->dt, a->x, a->v, a->quat, a->angmom, a->f, a->torque, a->shtype, a->mask,
                       a->groupbit, st));
  if (with_check) {
    const double trig = 0.5 * s->skin;
    HIPCHK(c, hipMemsetAsync(s->d_flags.p + 1, 0, sizeof(int), st));
    hipLaunchKernelGGL(check_distance_kernel, dim3(nblk(a->nlocal, kStepBlock)), dim3(kStepBlock), 0, st, a->nlocal,
                       (const double*)a->x, (const double*)s->d_xhold.p, trig * trig, s->d_flags.p + 1);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(s->h_flags, s->d_flags.p, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
  }
  return SHPAIR_OK;
}

// segment B: ghosts, forces, second half kick
static int enqueue_b(shpair_ctx* c, shstep_state* s, const shstep_arrays* a, int nghost, bool body, hipStream_t st)
{
  const size_t nall = (size_t)a->nlocal + nghost;
  RC(shstep_forward_device(c, a->x, a->quat, st));
  RC(shstep_force_clear_device(c, (int)nall, a->f, a->torque, st));
  RC(shpair_compute_device(c, a->nlocal, nghost, a->x, a->quat, a->type, a->shtype, 1, 0, 0, a->f, a->torque, nullptr, st));
  RC(shstep_reverse_device(c, a->f, a->torque, st));
  if (body)
    RC(shstep_post_force_device(c, a->nlocal, a->gravity, a->gamma_t, a->gamma_r, a->v, a->quat, a->angmom, a->shtype,
                                a->mask, a->groupbit, a->f, a->torque, st));
  RC(shstep_nve_device(c, 1, a->nlocal, a->dt, a->x, a->v, a->quat, a->angmom, a->f, a->torque, a->shtype, a->mask,
                       a->groupbit, st));
  (void)s;
  return SHPAIR_OK;
}

template <typename F>
static int capture(shpair_ctx* c, hipStream_t st, hipGraphExec_t* out, F&& body)
{
  hipGraph_t g = nullptr;
  HIPCHK(c, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  const int rc = body();
  const hipError_t e = hipStreamEndCapture(st, &g);
  if (rc) {
    if (g) (void)hipGraphDestroy(g);
    return rc;
  }
  if (e != hipSuccess) CTX_FAIL(c, SHPAIR_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  const hipError_t e2 = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e2 != hipSuccess) CTX_FAIL(c, SHPAIR_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e2));
  return SHPAIR_OK;
}

extern "C" {

int shstep_run_device(shpair_ctx* c, const shstep_arrays* a, int nsteps, int use_graph, int* nghost_io, int* rebuilds,
                      void* stream)
{
  STEP_PROLOGUE(c);
  if (rebuilds) *rebuilds = 0;
  if (!a || !nghost_io || nsteps < 0) CTX_FAIL(c, SHPAIR_EINVAL, "null arguments or nsteps < 0");
  if (a->nlocal < 0 || a->nmax < a->nlocal || a->check_every < 1 || !std::isfinite(a->dt))
    CTX_FAIL(c, SHPAIR_EINVAL, "bad nlocal (%d) / nmax (%d) / check_every (%d) / dt", a->nlocal, a->nmax, a->check_every);
  if (nsteps == 0 || a->nlocal == 0) return SHPAIR_OK;
  if (!a->x || !a->v || !a->quat || !a->angmom || !a->f || !a->torque || !a->type || !a->shtype || !a->mask)
    CTX_FAIL(c, SHPAIR_EINVAL, "null array pointer");
  if (s->l_nlocal != a->nlocal || !c->have_neighbors || s->b_nlocal != a->nlocal || *nghost_io != s->nghost)
    CTX_FAIL(c, SHPAIR_ESTATE, "run: ghosts and neighbour list of the current particles must be built first "
             "(shstep_borders_device + shstep_neighbor_build_device)");
  hipStream_t st = (hipStream_t)stream;
  if (use_graph && !st) CTX_FAIL(c, SHPAIR_EINVAL, "run: graph replay needs an explicit stream (the null stream cannot be captured)");
  if (use_graph && (c->opt_timing || c->opt_count)) CTX_FAIL(c, SHPAIR_ESTATE, "run: switch the timing / count options off for graph replay");
  RC(refresh_mass(c, s));
  RC(refresh_box(c, s));
  RC(shpair_prepare_tables(c));
  const bool body = a->gravity[0] != 0.0 || a->gravity[1] != 0.0 || a->gravity[2] != 0.0 || a->gamma_t != 0.0 || a->gamma_r != 0.0;
  int nghost = *nghost_io, nreb = 0;
  StepGraphs G;
  hipGraphExec_t g_nocheck = nullptr;
  int rc = SHPAIR_OK;
  // the graphs hold kernel arguments (ghost and pair counts, the context's list and x-hold buffers): they
  // are captured once and again after every rebuild
  auto recapture = [&]() -> int {
    G.reset();
    if (g_nocheck) (void)hipGraphExecDestroy(g_nocheck);
    g_nocheck = nullptr;
    int r = capture(c, st, &G.a, [&] { return enqueue_a(c, s, a, true, st); });
    if (!r && a->check_every > 1) r = capture(c, st, &g_nocheck, [&] { return enqueue_a(c, s, a, false, st); });
    if (!r) r = capture(c, st, &G.b, [&] { return enqueue_b(c, s, a, nghost, body, st); });
    return r;
  };
  auto launch = [&](hipGraphExec_t g) -> int {
    const hipError_t e = hipGraphLaunch(g, st);
    if (e != hipSuccess) {
      c->err = std::string("hipGraphLaunch failed: ") + hipGetErrorString(e);
      return SHPAIR_EHIP;
    }
    return SHPAIR_OK;
  };
  if (use_graph) rc = recapture();
  for (int step = 0; step < nsteps && rc == SHPAIR_OK; ++step) {
    const bool check = ((step + 1) % a->check_every) == 0;
    rc = use_graph ? launch(check ? G.a : g_nocheck) : enqueue_a(c, s, a, check, st);
    if (rc) break;
    if (check) {
      const hipError_t e = hipStreamSynchronize(st);
      if (e != hipSuccess) {
        c->err = std::string("hipStreamSynchronize failed: ") + hipGetErrorString(e);
        rc = SHPAIR_EHIP;
        break;
      }
      if (s->h_flags[0]) {
        (void)hipMemsetAsync(s->d_flags.p, 0, sizeof(int), st);
        c->err = "a shape index (shtype) outside the table reached a kernel; those particles were skipped";
        rc = SHPAIR_EINVAL;
        break;
      }
      if (s->h_flags[1]) {
        int np = 0;
        rc = shstep_borders_device(c, a->nlocal, a->nmax, a->x, a->quat, a->type, a->shtype, nullptr, &nghost, st);
        if (!rc) rc = shstep_neighbor_build_device(c, a->nlocal, nghost, a->x, a->shtype, nullptr, &np, st);
        if (!rc) ++nreb;
        if (!rc && use_graph) rc = recapture();
        if (rc) break;
      }
    }
    rc = use_graph ? launch(G.b) : enqueue_b(c, s, a, nghost, body, st);
  }
  const hipError_t es = hipStreamSynchronize(st);
  G.reset();
  if (g_nocheck) (void)hipGraphExecDestroy(g_nocheck);
  *nghost_io = nghost;
  if (rebuilds) *rebuilds = nreb;
  if (rc) return rc;
  if (es != hipSuccess) CTX_FAIL(c, SHPAIR_EHIP, "hipStreamSynchronize failed: %s", hipGetErrorString(es));
  return SHPAIR_OK;
}

}  // extern "C"
