// shpair_api.hip — the C ABI of include/shpair.h on top of the HIP kernels.
//
// Host side of the drop-in boundary: owns the per-shape tables, the expanded
// half list and (for the host-pointer entry point) the staging buffers.
// There is no CPU fallback in this library: every compute path launches the
// gfx950 kernels of pair_kernel.hpp.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/shpair.h"
#include "fp64_peak.hpp"
#include "pair_kernel.hpp"
#include "det_kernels.hpp"
#include "pair_setup.hpp"
#include "shpair_ctx.hpp"
#include "sh_const.hpp"
#include "sh_tables.hpp"

namespace shp {
#define SHP_DECL(L) void shp_launch_L##L(const PairParams&, bool, hipStream_t, hipEvent_t); \
  hipError_t shp_attr_L##L(bool, bool, hipFuncAttributes*, bool, bool, bool);
SHP_DECL(0) SHP_DECL(1) SHP_DECL(2) SHP_DECL(3) SHP_DECL(4) SHP_DECL(5) SHP_DECL(6)
SHP_DECL(7) SHP_DECL(8) SHP_DECL(9) SHP_DECL(10) SHP_DECL(11) SHP_DECL(12)
#undef SHP_DECL
void shp_launch_Lrt(const PairParams&, bool, hipStream_t, hipEvent_t);
hipError_t shp_attr_Lrt(bool, bool, hipFuncAttributes*, bool, bool, bool);

constexpr int kMaxUnrolledL = 12;
static const pair_launch_fn kLaunch[kMaxUnrolledL + 1] = {
    shp_launch_L0, shp_launch_L1, shp_launch_L2, shp_launch_L3, shp_launch_L4, shp_launch_L5, shp_launch_L6,
    shp_launch_L7, shp_launch_L8, shp_launch_L9, shp_launch_L10, shp_launch_L11, shp_launch_L12};
static const pair_attr_fn kAttr[kMaxUnrolledL + 1] = {
    shp_attr_L0, shp_attr_L1, shp_attr_L2, shp_attr_L3, shp_attr_L4, shp_attr_L5, shp_attr_L6,
    shp_attr_L7, shp_attr_L8, shp_attr_L9, shp_attr_L10, shp_attr_L11, shp_attr_L12};

// Sums the per-slot flags the pair kernel wrote: out[0] = contact pairs
// (flag >= 1), out[1] = touching pairs (flag == 2). One atomic per wave.
__global__ void count_flags_kernel(const unsigned char* __restrict__ flags, int npairs, unsigned long long* out)
{
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned char fl = (p < npairs) ? flags[p] : 0;
  const unsigned long long m1 = __ballot(fl >= 1), m2 = __ballot(fl == 2);
  if ((threadIdx.x & 63) == 0) {
    if (m1) atomicAdd(&out[0], (unsigned long long)__popcll(m1));
    if (m2) atomicAdd(&out[1], (unsigned long long)__popcll(m2));
  }
}

// Expands a device-resident CSR half list into one (i, j) per slot; one thread per row segment entry.
__global__ void expand_csr_kernel(const int* __restrict__ ilist, const int* __restrict__ offsets,
                                  const int* __restrict__ jlist, int inum, int* __restrict__ pair_i,
                                  int* __restrict__ pair_j)
{
  // one wave per row: rows are short (~6 entries), lanes stride the row
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (row >= inum) return;
  const int b = offsets[row], e = offsets[row + 1], i = ilist[row];
  for (int p = b + lane; p < e; p += 64) {
    pair_i[p] = i;
    pair_j[p] = jlist[p] & SHPAIR_NEIGHMASK;
  }
}

}  // namespace shp

using namespace shp;

extern "C" {

const char* shpair_version(void) { return "shpair 0.1 gfx950"; }

const char* shpair_strerror(int code)
{
  switch (code) {
    case SHPAIR_OK: return "ok";
    case SHPAIR_EINVAL: return "invalid argument";
    case SHPAIR_ENODEV: return "no usable HIP device (this library has no CPU fallback)";
    case SHPAIR_EHIP: return "HIP runtime error";
    case SHPAIR_ESTATE: return "call order error: shapes, coefficients or neighbour list not set";
    case SHPAIR_ENOMEM: return "out of memory";
    case SHPAIR_ELMAX: return "lmax or nq above the compiled limit";
    default: return "unknown shpair error";
  }
}

const char* shpair_last_error(const shpair_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int shpair_create(shpair_ctx** out, int device_id)
{
  if (!out) return SHPAIR_EINVAL;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SHPAIR_ENODEV;
  if (device_id < 0 || device_id >= ndev) return SHPAIR_ENODEV;
  if (hipSetDevice(device_id) != hipSuccess) return SHPAIR_ENODEV;
  shpair_ctx* c = new (std::nothrow) shpair_ctx();
  if (!c) return SHPAIR_ENOMEM;
  c->device = device_id;
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream_up, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_up, hipEventDisableTiming) != hipSuccess ||
      hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess ||
      hipEventCreate(&c->evA) != hipSuccess || hipEventCreate(&c->evB) != hipSuccess ||
      hipHostMalloc((void**)&c->h_ev, 7 * sizeof(double)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_counters, 2 * sizeof(unsigned long long)) != hipSuccess ||
      hipHostMalloc((void**)&c->h_err, sizeof(int)) != hipSuccess || c->d_err.ensure(1) != hipSuccess ||
      hipMemset(c->d_err.p, 0, sizeof(int)) != hipSuccess ||
      c->d_counters.ensure(2) != hipSuccess || c->d_ev.ensure(7) != hipSuccess) {
    shpair_destroy(c);
    return SHPAIR_EHIP;
  }
  *out = c;
  return SHPAIR_OK;
}

void shpair_destroy(shpair_ctx* c)
{
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  shstep_release_state(c);
  c->d_rc.release(); c->d_coef.release(); c->d_coefm.release(); c->d_rmax.release(); c->d_kn.release(); c->d_expo.release();
  c->d_quad.release(); c->d_pair_i.release(); c->d_pair_j.release();
  c->d_creal.release(); c->d_xval.release(); c->d_gscale.release(); c->d_xcol.release(); c->d_xinfo.release(); c->d_jval.release(); c->d_jcol.release();
  c->d_x.release(); c->d_quat.release(); c->d_f.release(); c->d_torque.release(); c->d_ev.release();
  c->d_type.release(); c->d_shtype.release(); c->d_counters.release(); c->d_flags.release();
  c->d_eatom.release(); c->d_vatom.release(); c->d_list.release(); c->d_err.release(); c->d_rec.release(); c->d_rec_i.release(); c->d_rot.release();
  c->d_pair_ft.release(); c->d_pair_ev.release(); c->d_rev_start.release(); c->d_rev_cur.release(); c->d_rev_ent.release();
  if (c->h_list) (void)hipHostFree(c->h_list);
  if (c->h_err) (void)hipHostFree(c->h_err);
  if (c->h_ev) (void)hipHostFree(c->h_ev);
  if (c->h_counters) (void)hipHostFree(c->h_counters);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->evA) (void)hipEventDestroy(c->evA);
  if (c->evB) (void)hipEventDestroy(c->evB);
  for (auto& pr : c->pinned) (void)hipHostUnregister(pr.first);
  (void)hipGetLastError();
  if (c->ev_up) (void)hipEventDestroy(c->ev_up);
  if (c->stream_up) (void)hipStreamDestroy(c->stream_up);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

int shpair_settings(shpair_ctx* c, int nq)
{
  if (!c) return SHPAIR_EINVAL;
  if (nq < 1) CTX_FAIL(c, SHPAIR_EINVAL, "pair_style sh: nq must be >= 1 (got %d)", nq);
  if (nq > SHPAIR_MAX_NQ) CTX_FAIL(c, SHPAIR_ELMAX, "pair_style sh: nq %d > %d", nq, SHPAIR_MAX_NQ);
  c->nq = nq;
  c->quad_dirty = true;
  return SHPAIR_OK;
}

int shpair_set_ntypes(shpair_ctx* c, int ntypes, int nshapes)
{
  if (!c) return SHPAIR_EINVAL;
  if (ntypes < 1 || nshapes < 1) CTX_FAIL(c, SHPAIR_EINVAL, "ntypes (%d) and nshapes (%d) must be >= 1", ntypes, nshapes);
  c->ntypes = ntypes;
  c->nshapes = nshapes;
  c->shapes.assign(nshapes, Shape());
  c->mass_dirty = true;
  c->kn.assign((size_t)(ntypes + 1) * (ntypes + 1), std::nan(""));
  c->expo.assign((size_t)(ntypes + 1) * (ntypes + 1), std::nan(""));
  c->tables_dirty = true;
  return SHPAIR_OK;
}

int shpair_set_shape(shpair_ctx* c, int ishape, int lmax, const double* anm, double rmax)
{
  if (!c) return SHPAIR_EINVAL;
  if (c->nshapes <= 0) CTX_FAIL(c, SHPAIR_ESTATE, "shpair_set_ntypes() must come first");
  if (ishape < 0 || ishape >= c->nshapes) CTX_FAIL(c, SHPAIR_EINVAL, "shape index %d outside [0,%d)", ishape, c->nshapes);
  if (!anm) CTX_FAIL(c, SHPAIR_EINVAL, "null coefficient pointer");
  if (lmax < 0) CTX_FAIL(c, SHPAIR_EINVAL, "lmax %d < 0", lmax);
  if (lmax > SHPAIR_MAX_LMAX) CTX_FAIL(c, SHPAIR_ELMAX, "lmax %d > %d", lmax, SHPAIR_MAX_LMAX);
  const int n = (lmax + 1) * (lmax + 2);
  for (int k = 0; k < n; ++k)
    if (!std::isfinite(anm[k])) CTX_FAIL(c, SHPAIR_EINVAL, "shape %d: coefficient %d is not finite", ishape, k);
  Shape& s = c->shapes[ishape];
  s.lmax = lmax;
  s.anm.assign(anm, anm + n);
  // The bounding radius is a HARD bound in the algorithm (bounding-sphere reject, cap angle, LAMMPS' cutoff): an
  // underestimate silently drops contacts.  The default is 1.01 x the maximum over a sample grid (docs/SPEC.md §1);
  // the maximum between the samples is found by a local search from the best nodes, and a radius below it is refused.
  const double rtrue = refined_max_radius(lmax, anm);
  const double rdef = default_rmax(lmax, anm);
  // rtrue is itself a rounded host evaluation: an exactly tight user bound (a sphere's a00 / sqrt(4 pi), say) may land an
  // ulp below it, and a shortfall of 1e-12 relative cannot drop a contact
  if (rmax > 0.0 && rmax < rtrue * (1.0 - 1e-12))
    CTX_FAIL(c, SHPAIR_EINVAL, "shape %d: the bounding radius %.17g is below the shape's largest radius %.17g", ishape, rmax, rtrue);
  if (!(rmax > 0.0) && rdef < rtrue)
    CTX_FAIL(c, SHPAIR_EINVAL, "shape %d: the default bounding radius %.17g (1.01 x the sampled maximum) is below the largest "
             "radius %.17g found between the samples; pass an explicit rmax", ishape, rdef, rtrue);
  s.rmax = (rmax > 0.0) ? rmax : rdef;
  if (!(s.rmax > 0.0) || !std::isfinite(s.rmax)) CTX_FAIL(c, SHPAIR_EINVAL, "shape %d: bounding radius %g is not positive", ishape, s.rmax);
  c->tables_dirty = true;
  c->mass_dirty = true;
  return SHPAIR_OK;
}

int shpair_set_coeff(shpair_ctx* c, int itype, int jtype, double kn, double exponent)
{
  if (!c) return SHPAIR_EINVAL;
  if (c->ntypes <= 0) CTX_FAIL(c, SHPAIR_ESTATE, "shpair_set_ntypes() must come first");
  if (itype < 1 || itype > c->ntypes || jtype < 1 || jtype > c->ntypes)
    CTX_FAIL(c, SHPAIR_EINVAL, "pair_coeff types %d %d outside [1,%d]", itype, jtype, c->ntypes);
  if (!(kn >= 0.0) || !std::isfinite(kn)) CTX_FAIL(c, SHPAIR_EINVAL, "pair_coeff: kn %g must be finite and >= 0", kn);
  if (!(exponent >= 1.0) || !std::isfinite(exponent)) CTX_FAIL(c, SHPAIR_EINVAL, "pair_coeff: exponent %g must be finite and >= 1", exponent);
  c->kn[(size_t)itype * (c->ntypes + 1) + jtype] = kn;
  c->expo[(size_t)itype * (c->ntypes + 1) + jtype] = exponent;
  c->tables_dirty = true;
  return SHPAIR_OK;
}

int shpair_get_rmax(const shpair_ctx* c, int ishape, double* rmax)
{
  if (!c || !rmax) return SHPAIR_EINVAL;
  if (ishape < 0 || ishape >= c->nshapes || c->shapes[ishape].lmax < 0) return SHPAIR_EINVAL;
  *rmax = c->shapes[ishape].rmax;
  return SHPAIR_OK;
}

int shpair_shape_radius(int lmax, const double* anm, const double* u, double* r)
{
  if (lmax < 0 || lmax > SHPAIR_MAX_LMAX || !anm || !u || !r) return SHPAIR_EINVAL;
  *r = host_radius(lmax, anm, u);
  return SHPAIR_OK;
}

int shpair_shape_default_rmax(int lmax, const double* anm, double* rmax)
{
  if (lmax < 0 || lmax > SHPAIR_MAX_LMAX || !anm || !rmax) return SHPAIR_EINVAL;
  *rmax = default_rmax(lmax, anm);
  return SHPAIR_OK;
}

// Host list -> device: the rows are flattened into ONE pinned buffer [ilist | offsets | jlist] (a LAMMPS list is
// paged, so one pass over it is unavoidable), uploaded with one copy and expanded to one (i, j) per slot on the
// device (expand_csr_kernel, which also strips the NEIGHMASK bits).
static int stage_list(shpair_ctx* c, int inum, size_t tot)
{
  const size_t need = 2 * (size_t)inum + 1 + tot;
  if (c->h_list_cap < need) {
    if (c->h_list) (void)hipHostFree(c->h_list);
    c->h_list = nullptr;
    c->h_list_cap = 0;
    const size_t want = need + need / 4 + 64;
    HIPCHK(c, hipHostMalloc((void**)&c->h_list, want * sizeof(int)));
    c->h_list_cap = want;
  }
  return SHPAIR_OK;
}

static int upload_staged_list(shpair_ctx* c, int inum, size_t tot, int max_index)
{
  HIPCHK(c, hipSetDevice(c->device));
  const size_t need = 2 * (size_t)inum + 1 + tot;
  HIPCHK(c, c->d_list.ensure(need ? need : 1));
  HIPCHK(c, c->d_pair_i.ensure(tot ? tot : 1));
  HIPCHK(c, c->d_pair_j.ensure(tot ? tot : 1));
  // the previous list may still be in use by an enqueued compute
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (tot > 0) {
    HIPCHK(c, hipMemcpyAsync(c->d_list.p, c->h_list, need * sizeof(int), hipMemcpyHostToDevice, c->stream));
    const int threads = 256, rows_per_block = threads / 64;
    hipLaunchKernelGGL(expand_csr_kernel, dim3((inum + rows_per_block - 1) / rows_per_block), dim3(threads), 0, c->stream,
                       (const int*)c->d_list.p, (const int*)c->d_list.p + inum, (const int*)c->d_list.p + 2 * (size_t)inum + 1, inum,
                       c->d_pair_i.p, c->d_pair_j.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  HIPCHK(c, shp_size_pair_buffers(c, tot));
  c->npairs = (int)tot;
  c->n_interior = 0;   // a host-installed list: which slots touch ghosts is not known here
  c->max_atom_index = max_index;
  c->have_neighbors = true;
  ++c->list_gen;
  shstep_invalidate_list(c);
  return SHPAIR_OK;
}

int shpair_set_neighbors(shpair_ctx* c, int inum, const int* ilist, const int* numneigh, const int* const* firstneigh)
{
  if (!c) return SHPAIR_EINVAL;
  if (inum < 0 || (inum > 0 && (!ilist || !numneigh || !firstneigh))) CTX_FAIL(c, SHPAIR_EINVAL, "bad neighbour list arguments");
  size_t tot = 0;
  for (int ii = 0; ii < inum; ++ii) {
    const int i = ilist[ii];
    if (i < 0) CTX_FAIL(c, SHPAIR_EINVAL, "negative atom index in ilist (row %d)", ii);
    const int n = numneigh[i];
    if (n < 0) CTX_FAIL(c, SHPAIR_EINVAL, "numneigh[%d] = %d", i, n);
    if (n > 0 && !firstneigh[i]) CTX_FAIL(c, SHPAIR_EINVAL, "firstneigh[%d] is null", i);
    tot += (size_t)n;
  }
  if (tot > 0x7fffffffULL) CTX_FAIL(c, SHPAIR_EINVAL, "half list too long (%zu pairs)", tot);
  {
    const int rc = stage_list(c, inum, tot);
    if (rc) return rc;
  }
  int* il = c->h_list;
  int* of = c->h_list + inum;
  int* jl = c->h_list + 2 * (size_t)inum + 1;
  int mx = -1;
  size_t p = 0;
  for (int ii = 0; ii < inum; ++ii) {
    const int i = ilist[ii];
    const int n = numneigh[i];
    const int* row = firstneigh[i];
    il[ii] = i;
    of[ii] = (int)p;
    if (i > mx) mx = i;
    for (int jj = 0; jj < n; ++jj) {
      const int j = row[jj] & SHPAIR_NEIGHMASK;
      jl[p + jj] = j;
      if (j > mx) mx = j;
    }
    p += (size_t)n;
  }
  if (inum >= 0) of[inum] = (int)p;
  return upload_staged_list(c, inum, tot, mx);
}

int shpair_set_neighbors_csr(shpair_ctx* c, int inum, const int* ilist, const int* offsets, const int* jlist)
{
  if (!c) return SHPAIR_EINVAL;
  if (inum < 0 || (inum > 0 && (!ilist || !offsets))) CTX_FAIL(c, SHPAIR_EINVAL, "bad neighbour list arguments");
  size_t tot = 0;
  if (inum > 0) {
    if (offsets[0] != 0) CTX_FAIL(c, SHPAIR_EINVAL, "offsets[0] must be 0");
    if (offsets[inum] < 0 || (offsets[inum] > 0 && !jlist)) CTX_FAIL(c, SHPAIR_EINVAL, "bad CSR neighbour list");
    tot = (size_t)offsets[inum];
    for (int ii = 0; ii < inum; ++ii) {
      if (offsets[ii + 1] < offsets[ii]) CTX_FAIL(c, SHPAIR_EINVAL, "offsets not monotone at %d", ii);
      if (ilist[ii] < 0) CTX_FAIL(c, SHPAIR_EINVAL, "negative atom index in ilist (row %d)", ii);
    }
  }
  {
    const int rc = stage_list(c, inum, tot);
    if (rc) return rc;
  }
  int mx = -1;
  if (inum > 0) {
    std::memcpy(c->h_list, ilist, (size_t)inum * sizeof(int));
    std::memcpy(c->h_list + inum, offsets, ((size_t)inum + 1) * sizeof(int));
    int* jl = c->h_list + 2 * (size_t)inum + 1;
    for (int ii = 0; ii < inum; ++ii)
      if (ilist[ii] > mx) mx = ilist[ii];
    for (size_t k = 0; k < tot; ++k) {
      const int j = jlist[k] & SHPAIR_NEIGHMASK;
      jl[k] = j;
      if (j > mx) mx = j;
    }
  } else {
    c->h_list[0] = 0;
  }
  return upload_staged_list(c, inum, tot, mx);
}

int shpair_set_neighbors_device(shpair_ctx* c, int inum, const int* ilist, const int* offsets, const int* jlist,
                                int npairs, int max_atom_index, void* stream)
{
  if (!c) return SHPAIR_EINVAL;
  if (inum < 0 || npairs < 0 || (inum > 0 && (!ilist || !offsets)) || (npairs > 0 && !jlist))
    CTX_FAIL(c, SHPAIR_EINVAL, "bad device neighbour list arguments");
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, c->d_pair_i.ensure(npairs ? npairs : 1));
  HIPCHK(c, c->d_pair_j.ensure(npairs ? npairs : 1));
  hipStream_t st = (hipStream_t)stream;
  if (inum > 0 && npairs > 0) {
    const int threads = 256, rows_per_block = threads / 64;
    hipLaunchKernelGGL(expand_csr_kernel, dim3((inum + rows_per_block - 1) / rows_per_block), dim3(threads), 0, st, ilist,
                       offsets, jlist, inum, c->d_pair_i.p, c->d_pair_j.p);
    HIPCHK(c, hipGetLastError());
  }
  HIPCHK(c, shp_size_pair_buffers(c, (size_t)npairs));
  c->npairs = npairs;
  c->n_interior = 0;
  c->max_atom_index = max_atom_index;
  c->have_neighbors = true;
  ++c->list_gen;
  shstep_invalidate_list(c);
  return SHPAIR_OK;
}

// Which kernel family evaluates particle j (pair_kernel.hpp): per-azimuth polynomials in the pair's common frame
// (JPT kernels + rotation kernel) or the body-frame Horner evaluation.  The first trades ~170 instructions and a
// table build per pair for 60 fewer per radius evaluation: it wins unless a pair has very few cap nodes.  Option
// "jpoly": 1 / 0 force, -1 (default) the measured rule (interleaved A/B over L = 0..12 x n_q = 4..32,
// profiles/r02_y_jpoly_matrix.txt: the body-frame family was faster only at n_q = 4 from L = 6 and at n_q <= 8 from L = 9;
// re-measured in round 4, below).
static bool use_jpoly_at(const shpair_ctx* c, const int L)
{
  if (L > kMaxUnrolledL || c->opt_variant == 1 || c->opt_rule) return false;
  if (c->opt_jpoly >= 0) return c->opt_jpoly == 1;
  // Round 4 (end-of-round kernels, profiles/r04_q6_jpoly_small_nq.txt, r04_q6_jpoly_tiny_nq.txt): the per-azimuth family
  // has caught up everywhere but at L >= 10 with n_q <= 5 (L = 12 / 4 +3 %, L = 11 / 5 +3 %, L = 12 / 1 +16 %) — round 2's
  // rule kept the body-frame kernels at n_q < 6 from L = 6 and at n_q < 12 from L = 9, where they now lose by 5...28 %
  // (L = 9 / 10 3.10 -> 2.24 ms, L = 12 / 10 4.41 -> 3.29, L = 8 / 4 1.90 -> 1.55, L = 6 / 4 1.25 -> 1.09)
  if (L >= 10) return c->nq >= 6;   // (L = 10 / 3, 4, 5: the body-frame kernels 3.5...4.5 % faster, r04_q7_sweep4.txt)
  return true;
}
static bool use_jpoly(const shpair_ctx* c) { return use_jpoly_at(c, c->lmax); }

// Two waves per pair (pair_kernel.hpp pair_lds_layout2): the JPT kernels of the orders it is compiled for, even n_q.
// Pays where one wave's private copy of the tables leaves a CU too few waves for its dependent FP64 chains — large L
// with large n_q (L = 12, n_q = 32: 14.6 KB per one-wave pair = 11 waves per CU, 17.3 KB per two-wave pair = the
// 16 the registers allow).  Option "split": 1 / 0 force, -1 (default) the measured rule (profiles/r03_*_split_matrix.txt).
static bool use_split(const shpair_ctx* c, const bool jpoly)
{
  if (!jpoly || !split_compiled(c->lmax) || c->lmax > kMaxUnrolledL || (c->nq & 1) || c->nq < 8) return false;
  if (c->opt_split >= 0) return c->opt_split == 1;
  // measured (interleaved A/B over L = 7..12 x n_q = 8..32, profiles/r03_g/h_split_matrix.txt and, on the end-of-round
  // kernels with their ring groups re-sized, r03_fin_split_matrix.txt; the boxes' noise is +-3 %): two waves win at
  // n_q = 32 from L = 8 on (0...-7 %), at n_q = 24 from L = 11 on (-3 %; L = 10: +2.5 %) and at L = 12 from n_q = 16
  // on (-11 %); they lose below (at n_q = 8 half of each wave's lanes have no node pair: +40 %)
  // Round 4, end-of-round kernels (Horner ring tables, larger node queue, direct batches; profiles/r04_q5_sweep2.txt):
  // with all 16 rings resident one wave beats two at L = 12 / n_q = 16 (-4.9 %), and one wave with 8-ring groups at
  // L = 11 / n_q = 24 (-4.9 %); two waves keep n_q >= 32 from L = 8 (L = 9 / 32 -4.8 %) and L = 12 from n_q = 18
  // (L = 12 / 20 -5.7 % against the best one-wave form)
  // ... and, from L = 9, n_q >= 22 where one wave's ring groups cannot end on slab boundaries (n_q = 22, 26, 28, 30; not
  // 24): L = 11 / 22 -13 %, L = 9 / 26 -7.7 %, L = 9 / 28 -6.1 %, L = 11 / 26 -5.0 %, L = 10 / 22 -4.2 %
  // (profiles/r04_q7_sweep4.txt)
  int g = 64, a = c->nq;
  while (a) { const int t = g % a; g = a; a = t; }   // gcd(64, n_q): a one-wave slab spans 64 / g rings
  const bool aligned1 = 2 * (64 / g) <= c->nq;
  return (c->lmax >= 8 && c->nq >= 32) || (c->lmax >= 12 && c->nq >= 18) || (c->lmax >= 9 && c->nq >= 22 && !aligned1);
}

}  // extern "C"

// Sizes the per-slot buffers the pair kernels write (records; rotated coefficient vectors of the JPT family) for a
// list of `np` slots: called wherever a list is installed, so that a compute — possibly inside a stream capture —
// allocates nothing.
hipError_t shp_size_pair_buffers(shpair_ctx* c, size_t np)
{
  if (np == 0) np = 1;
  c->rev_dirty = true;   // a list is being installed: the reverse index of the deterministic mode is stale
  hipError_t e = c->d_rec.ensure(np * kRecStride);
  if (e == hipSuccess && c->opt_deterministic) {
    e = c->d_pair_ft.ensure(np * 12);
    if (e == hipSuccess) e = c->d_rev_ent.ensure(np * 2);
  }
  if (e == hipSuccess) e = c->d_rec_i.ensure(np * 4);
  if (e == hipSuccess) e = c->d_pair_ev.ensure(8 * (np + (np + kTallyChunk - 1) / kTallyChunk));   // 64 B per slot: thermo steps
  int L = c->lmax;
  for (int s = 0; s < c->nshapes; ++s)
    if (c->shapes[s].lmax > L) L = c->shapes[s].lmax;
  if (e == hipSuccess && L >= 0 && c->nq > 0) {
    if (use_jpoly_at(c, L)) e = c->d_rot.ensure(rot_buffer_doubles(L, 2 * np));
  }
  return e;
}

extern "C" {

static int upload_tables(shpair_ctx* c)
{
  if (c->nshapes <= 0 || c->ntypes <= 0) CTX_FAIL(c, SHPAIR_ESTATE, "shpair_set_ntypes() not called");
  int L = -1;
  for (int s = 0; s < c->nshapes; ++s) {
    if (c->shapes[s].lmax < 0) CTX_FAIL(c, SHPAIR_ESTATE, "shape %d was never set", s);
    if (c->shapes[s].lmax > L) L = c->shapes[s].lmax;
  }
  c->any_nonunit_exponent = false;
  for (int a = 1; a <= c->ntypes; ++a)
    for (int b = 1; b <= c->ntypes; ++b) {
      const double k = c->kn[(size_t)a * (c->ntypes + 1) + b], m = c->expo[(size_t)a * (c->ntypes + 1) + b];
      if (std::isnan(k) || std::isnan(m)) CTX_FAIL(c, SHPAIR_ESTATE, "pair_coeff for types %d %d was never set", a, b);
      if (m != 1.0) c->any_nonunit_exponent = true;
    }
  std::vector<double> rc_n, rc, scale, cw_n, cw, all, allm, wm, rmax;
  build_recurrence(L, rc_n, scale);
  to_m_major(L, 1, rc_n, rc);
  const int T = (L + 1) * (L + 2) / 2;
  all.reserve((size_t)c->nshapes * 2 * T);
  for (int s = 0; s < c->nshapes; ++s) {
    build_coefficients(L, c->shapes[s].lmax, c->shapes[s].anm.data(), rc_n, scale, cw_n);
    to_m_major(L, 2, cw_n, cw);
    cw.resize(sh_chunk_stride(L), 0.0);
    all.insert(all.end(), cw.begin(), cw.end());
    build_monomial(L, c->shapes[s].lmax, c->shapes[s].anm.data(), wm);
    wm.resize(sh_chunk_stride(L), 0.0);
    allm.insert(allm.end(), wm.begin(), wm.end());
    rmax.push_back(c->shapes[s].rmax);
  }
  // cap-frame evaluation of particle i: real-basis coefficients, X matrices, ring scale
  std::vector<double> creal_all, cr, xval, gs;
  std::vector<int> xcol, xinfo;
  for (int s = 0; s < c->nshapes; ++s) {
    real_coefficients(L, c->shapes[s].lmax, c->shapes[s].anm.data(), cr);
    creal_all.insert(creal_all.end(), cr.begin(), cr.end());
  }
  build_xmats_ell(L, xval, xcol, xinfo);
  if (xval.empty()) CTX_FAIL(c, SHPAIR_EINVAL, "internal: X matrix row wider than lmax/2+1");
  build_ring_scale(L, gs);
  std::vector<double> jval;
  std::vector<int> jcol;
  build_jpoly_ell(L, jval, jcol);
  if (jval.empty()) CTX_FAIL(c, SHPAIR_EINVAL, "internal: per-azimuth polynomial row wider than lmax/2+1");
  // a kernel still in flight on ANY stream (the caller's, not only the context's) may be reading the old tables
  HIPCHK(c, hipDeviceSynchronize());
  HIPCHK(c, c->d_creal.ensure(creal_all.size()));
  HIPCHK(c, c->d_xval.ensure(xval.size()));
  HIPCHK(c, c->d_xcol.ensure(xcol.size()));
  HIPCHK(c, c->d_xinfo.ensure(xinfo.size()));
  HIPCHK(c, c->d_gscale.ensure(gs.size()));
  HIPCHK(c, c->d_jval.ensure(jval.size()));
  HIPCHK(c, c->d_jcol.ensure(jcol.size()));
  HIPCHK(c, hipMemcpy(c->d_jval.p, jval.data(), jval.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_jcol.p, jcol.data(), jcol.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_creal.p, creal_all.data(), creal_all.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_xval.p, xval.data(), xval.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_xcol.p, xcol.data(), xcol.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_xinfo.p, xinfo.data(), xinfo.size() * sizeof(int), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_gscale.p, gs.data(), gs.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, c->d_rc.ensure(rc.size()));
  HIPCHK(c, c->d_coef.ensure(all.size()));
  HIPCHK(c, c->d_coefm.ensure(allm.size()));
  HIPCHK(c, hipMemcpy(c->d_coefm.p, allm.data(), allm.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, c->d_rmax.ensure(rmax.size()));
  HIPCHK(c, c->d_kn.ensure(c->kn.size()));
  HIPCHK(c, c->d_expo.ensure(c->expo.size()));
  HIPCHK(c, hipMemcpy(c->d_rc.p, rc.data(), rc.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_coef.p, all.data(), all.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_rmax.p, rmax.data(), rmax.size() * sizeof(double), hipMemcpyHostToDevice));
  // unset entries were rejected above; upload as is
  HIPCHK(c, hipMemcpy(c->d_kn.p, c->kn.data(), c->kn.size() * sizeof(double), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(c->d_expo.p, c->expo.data(), c->expo.size() * sizeof(double), hipMemcpyHostToDevice));
  c->lmax = L;
  c->cstride = sh_chunk_stride(L);
  (void)T;
  c->tables_dirty = false;
  c->quad_dirty = true;  // the cos/sin(m psi) table of upload_quadrature is sized by lmax
  return SHPAIR_OK;
}

static int upload_quadrature(shpair_ctx* c)
{
  const int nq = c->nq, npsi = 2 * nq;
  const int nm = c->lmax >= 2 ? c->lmax - 1 : 0;  // orders m = 2..lmax of the cos/sin(m psi) table, m-major
  // ... followed by (cos, sin)(m psi_l), m = 0..lmax + 1, of the first n_q azimuths, l-major: the azimuth stage of particle
  // j's polynomials (pair_kernel.hpp jpoly_build; psi_(l + n_q) = psi_l + pi only flips the sign of the odd orders)
  const size_t trigj_off = 2 * nq + 2 * npsi + (size_t)nm * 2 * npsi;
  std::vector<double> t, w, q(trigj_off + (size_t)nq * (c->lmax + 2) * 2);
  gauss_legendre(nq, t, w);
  for (int k = 0; k < nq; ++k) {
    q[k] = t[k];
    q[nq + k] = w[k];
  }
  for (int l = 0; l < npsi; ++l) {
    const double psi = 2.0 * 3.14159265358979323846264338327950288 * (l + 0.5) / npsi;
    q[2 * nq + l] = std::cos(psi);
    q[2 * nq + npsi + l] = std::sin(psi);
    for (int m = 2; m <= c->lmax; ++m) {
      // layout: pair_kernel.hpp trig_lmajor()
      const size_t e = trig_lmajor(c->lmax) ? ((size_t)l * nm + (m - 2)) : ((size_t)(m - 2) * npsi + l);
      q[2 * nq + 2 * npsi + 2 * e] = std::cos(m * psi);
      q[2 * nq + 2 * npsi + 2 * e + 1] = std::sin(m * psi);
    }
    if (l < nq)
      for (int m = 0; m <= c->lmax + 1; ++m) {   // one order more than exists: jpoly_build reads it against zeros
        q[trigj_off + ((size_t)l * (c->lmax + 2) + m) * 2] = std::cos(m * psi);
        q[trigj_off + ((size_t)l * (c->lmax + 2) + m) * 2 + 1] = std::sin(m * psi);
      }
  }
  HIPCHK(c, hipDeviceSynchronize());  // as in upload_tables
  HIPCHK(c, c->d_quad.ensure(q.size()));
  HIPCHK(c, hipMemcpy(c->d_quad.p, q.data(), q.size() * sizeof(double), hipMemcpyHostToDevice));
  c->quad_dirty = false;
  return SHPAIR_OK;
}

}  // extern "C"

// Uploads whatever table is stale (blocking copies): what shpair_compute_device() does on demand, callable
// ahead of a stream capture in which such copies are not allowed (shstep_run_device).
int shpair_prepare_tables(shpair_ctx* c)
{
  HIPCHK(c, hipSetDevice(c->device));
  if (c->tables_dirty) {
    const int rc = upload_tables(c);
    if (rc) return rc;
  }
  if (c->quad_dirty) {
    const int rc = upload_quadrature(c);
    if (rc) return rc;
  }
  HIPCHK(c, shp_size_pair_buffers(c, (size_t)c->npairs));
  return SHPAIR_OK;
}

// Reads and clears the error bits the pair kernel raises instead of reading outside a table.  Blocks on `stream`.
int shpair_check_device_errors(shpair_ctx* c, void* stream)
{
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(c, hipMemcpyAsync(c->h_err, c->d_err.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  if (*c->h_err) {
    const int bits = *c->h_err;
    HIPCHK(c, hipMemsetAsync(c->d_err.p, 0, sizeof(int), st));
    if (bits & (kPairErrShape | kPairErrType))
      CTX_FAIL(c, SHPAIR_EINVAL, "an atom %s outside its table reached the pair kernel; the pairs of those atoms were skipped",
               (bits & kPairErrShape) ? "shape index (shtype)" : "type");
    CTX_FAIL(c, SHPAIR_EINVAL, "coincident centres: a listed pair has separation 0 (or a position that is not a number); it was "
             "skipped (docs/SPEC.md 2, step 1)");
  }
  return SHPAIR_OK;
}

extern "C" {

// The pair path over a RANGE of list slots.  part & kPartPre: everything that has to happen once before the first slot
// of a step (buffer memsets of the deterministic mode and the tallies, the reverse index, the start-of-timing event);
// part & kPartPost: what follows the last slot (ordered gather, tally reduce, end-of-timing event, contact counts).
// shpair_compute_device = both parts over the whole list; the halo loop (shhalo_api.hip) runs the slots whose atoms
// are all owned — [0, split) — with kPartPre while the forward exchange is in flight, then [split, npairs) with
// kPartPost.  `split` must be a multiple of 32 (rotation tiles).
int shpair_compute_device(shpair_ctx* c, int nlocal, int nghost, const double* x, const double* quat, const int* type,
                          const int* shtype, int newton_pair, int eflag, int vflag, double* f, double* torque,
                          double* ev, void* stream)
{
  if (!c) return SHPAIR_EINVAL;
  return shp_compute_range(c, nlocal, nghost, x, quat, type, shtype, newton_pair, eflag, vflag, f, torque, ev, stream, 0,
                           c->npairs, kPartPre | kPartPost);
}

int shp_compute_range(shpair_ctx* c, int nlocal, int nghost, const double* x, const double* quat, const int* type,
                      const int* shtype, int newton_pair, int eflag, int vflag, double* f, double* torque, double* ev,
                      void* stream, const int slot0, const int slot_end, const int part)
{
  if (!c) return SHPAIR_EINVAL;
  if (slot0 < 0 || slot_end < slot0 || slot_end > c->npairs || (slot0 & 31) != 0)
    CTX_FAIL(c, SHPAIR_EINVAL, "compute range [%d, %d) of a list of %d slots (the first slot must be a multiple of 32)", slot0,
             slot_end, c->npairs);
  const bool pre = (part & kPartPre) != 0, post = (part & kPartPost) != 0;
  if (nlocal < 0 || nghost < 0) CTX_FAIL(c, SHPAIR_EINVAL, "negative atom counts");
  if (!c->have_neighbors) CTX_FAIL(c, SHPAIR_ESTATE, "no neighbour list: call shpair_set_neighbors() first");
  if ((eflag || vflag) && !ev) CTX_FAIL(c, SHPAIR_EINVAL, "eflag/vflag set but ev_dev is null");
  HIPCHK(c, hipSetDevice(c->device));
  if (c->tables_dirty) {
    const int rc = upload_tables(c);
    if (rc) return rc;
  }
  if (c->quad_dirty) {
    const int rc = upload_quadrature(c);
    if (rc) return rc;
  }
  if (pre) {
    c->timed_last = false;
    c->counted_last = false;
    c->stats.n_candidates = c->npairs;
  }
  if (c->npairs == 0) return SHPAIR_OK;
  if (!x || !quat || !type || !shtype || !f || !torque) CTX_FAIL(c, SHPAIR_EINVAL, "null atom array");
  if ((long long)c->max_atom_index >= (long long)nlocal + nghost)
    CTX_FAIL(c, SHPAIR_EINVAL, "the neighbour list refers to atom %d but nlocal + nghost = %lld (stale list?)",
             c->max_atom_index, (long long)nlocal + nghost);
  hipStream_t st = (hipStream_t)stream;  // NULL = HIP null stream
  {
    // The accumulation uses hardware FP64 atomics (-munsafe-fp-atomics), which are only reliable on ordinary
    // (coarse-grained) device memory: on host-coherent / managed allocations the adds can be dropped silently.
    const void* outp[3] = {f, torque, ev};
    for (int k = 0; k < 3; ++k) {
      if (!outp[k] || outp[k] == c->ok_ptr[k]) continue;
      hipPointerAttribute_t at;
      const hipError_t e = hipPointerGetAttributes(&at, outp[k]);
      if (e != hipSuccess) {
        (void)hipGetLastError();
        CTX_FAIL(c, SHPAIR_EINVAL, "%s is not a device pointer known to HIP (%s): the output arrays must be hipMalloc memory",
                 k == 0 ? "f" : (k == 1 ? "torque" : "ev"), hipGetErrorString(e));
      }
      if (at.type != hipMemoryTypeDevice)
        CTX_FAIL(c, SHPAIR_EINVAL, "%s is %s memory: the FP64 atomic accumulation needs ordinary device memory (hipMalloc)",
                 k == 0 ? "f" : (k == 1 ? "torque" : "ev"), at.type == hipMemoryTypeManaged ? "managed" : "host");
      c->ok_ptr[k] = outp[k];
    }
  }

  PairParams P;
  P.x = x; P.quat = quat; P.type = type; P.shtype = shtype; P.f = f; P.torque = torque;
  P.pair_i = c->d_pair_i.p; P.pair_j = c->d_pair_j.p; P.npairs = slot_end; P.slot0 = slot0;
  P.nlocal = nlocal; P.newton_pair = newton_pair ? 1 : 0;
  P.rc = c->d_rc.p; P.coef = c->d_coef.p; P.rmax = c->d_rmax.p; P.cstride = c->cstride; P.lmax = c->lmax;
  P.nshapes = c->nshapes; P.err = c->d_err.p;
  P.kn = c->d_kn.p; P.expo = c->d_expo.p; P.ntypes = c->ntypes;
  const int nq = c->nq;
  P.glt = c->d_quad.p; P.glw = c->d_quad.p + nq; P.cpsi = c->d_quad.p + 2 * nq; P.spsi = c->d_quad.p + 4 * nq;
  P.rule = c->opt_rule;
  P.eatom = c->eatom_dev;
  P.vatom = c->vatom_dev;
  P.nq = nq;
  P.trig = c->d_quad.p + 6 * nq;
  P.trig_stride = trig_lmajor(c->lmax) ? 2 * (c->lmax - 1) : 4 * nq;
  P.creal = c->d_creal.p; P.xval = c->d_xval.p; P.xcol = c->d_xcol.p; P.xinfo = c->d_xinfo.p; P.gscale = c->d_gscale.p;
  P.jval = c->d_jval.p; P.jcol = c->d_jcol.p;
  P.trigj = c->d_quad.p + 6 * nq + (size_t)(c->lmax >= 2 ? c->lmax - 1 : 0) * 4 * nq;
  // compiled orders evaluate particle j from per-azimuth polynomials in the pair's common frame (pair_kernel.hpp)
  const bool jpoly = use_jpoly(c);
  c->last_jpoly = jpoly;
  P.jpoly = jpoly ? 1 : 0;
  const bool split = use_split(c, jpoly);
  c->last_split = split;
  P.split = split ? 1 : 0;
  const int nqj = jpoly ? nq : 0;   // rows of the per-azimuth table in a wave's LDS
  {
    // Resident ring rows: all nq if a wave then needs <= 8 KB of LDS (five 4-wave workgroups per CU,
    // the VGPR-limited 5 waves/SIMD), else as many as fit 8 KB, never fewer than one slab of 64 nodes
    // spans.  Measured at lmax 12, nq 32 (tools/ab_libs.py --ring-rows): 32 or 18 rows 55 ms
    // (2 workgroups per CU), 9 rows 38.7 ms, 4 rows 37.4 ms.
    const int npsi = 2 * nq;
    // lanes per ring in phase 1: 2 n_q nodes, or n_q node pairs in the per-azimuth-polynomial kernels, whose table of
    // particle j comes on top of the 8 KB
    const int per_ring = jpoly ? (split ? nq / 2 : nq) : npsi;
    const int rows_min = 1 + (63 + per_ring - 1) / per_ring;
    int rows = nq;
    if (c->opt_ring_rows > 0) rows = c->opt_ring_rows;
    else if (jpoly && !c->opt_rule) {
      // Per-azimuth kernels, one wave per pair (sweeps of --ring-rows on the end-of-round kernels,
      // profiles/r03_fin_ring_rows.txt): one ring group while the wave's LDS — particle j's table included — stays
      // within 11.5 KB (13-14 waves per CU; L = 6, n_q = 24: one group of 24 rows beats two of 12 by 4 %); beyond, groups
      // of about 10 KB (L = 8, 9 / n_q = 24 -7...-9 % against 12-13 KB groups, L = 7 / 24 -4 %), see below (the rule
      // before sized the groups without j's table and left L = 9, n_q = 16 with groups of 14 + 2 rings: +6 %).  Known
      // exception: L = 8, n_q = 20, where 17 + 3 rings measured 4 % faster than the 10 + 10 this rule picks.
      const auto total = [&](const int r) { return wave_lds_layout(c->lmax, r, false, nqj).bytes; };
      int step = 64, a = per_ring;
      while (a) { const int t = step % a; step = a; a = t; }   // gcd(64, per_ring)
      step = 64 / step;   // rings per whole number of slabs
      // Round 4 (profiles/r04_q5_sweep2.txt, r04_q5_ring_rows.txt; the table builds got cheaper, the node loops did not):
      // one group up to 13 KB where the cap is four slabs (n_q <= 16: L = 11 / 16 -6.8 %, L = 12 / 16 -4.9 % with the one
      // wave that goes with it), and up to 14.5 KB from L = 9 on where groups cannot end on slab boundaries (n_q = 20:
      // L = 9 -3.6 %, L = 10 -3.2 %, L = 11 -2.7 %); n_q = 24 and 32 keep their aligned groups at those sizes
      const int one_group_max = (2 * step <= nq) ? (nq <= 16 ? 13312 : 11776) : (c->lmax >= 9 ? 14848 : 11776);
      if (total(nq) > one_group_max) {
        if (2 * step <= nq) {
          // groups that end on a slab boundary (no slab straddles a hand-over: at n_q = 24 every order measured,
          // L = 7...11, wants 8 rings = 3 slabs, not the 12 a budget alone gives): the largest such group within 10 KB
          // (16 waves per CU), the smallest if none fits; then as few groups as that takes, of equal aligned size
          int rfit = step;
          while (rfit + step <= nq && total(rfit + step) <= 10 * 1024) rfit += step;
          const int groups = (nq + rfit - 1) / rfit;
          rows = (((nq + groups - 1) / groups + step - 1) / step) * step;
        } else {
          int rmax = rows_min;
          while (rmax < nq && total(rmax + 1) <= 10752) ++rmax;   // 15 waves per CU
          const int groups = (nq + rmax - 1) / rmax;
          rows = (nq + groups - 1) / groups;
        }
        if (rows < (nq + 3) / 4) rows = (nq + 3) / 4;   // never more than four groups (large L x n_q: j's table alone
      }                                                  // fills the budget; those run two waves per pair anyway)
    } else if (wave_lds_layout(c->lmax, nq, false, 0).bytes > 8 * 1024) {
      const int fixed = wave_lds_layout(c->lmax, 0, false, 0).bytes;
      rows = (8 * 1024 - fixed) / (32 * (c->lmax + 1));
    }
    if (rows < rows_min) rows = rows_min;
    if (rows > nq) rows = nq;
    if (c->opt_rule) {
      // SPEC §2.8: the weights of a slab need its neighbours' residuals, which the kernel keeps in a window of
      // three slabs; a ring group must hold the rings of the slab being weighed and of the next one
      if (c->lmax > kMaxUnrolledL || c->opt_variant == 1 || nq > 32)
        CTX_FAIL(c, SHPAIR_ELMAX, "the weighted rule needs lmax <= %d and nq <= 32 (have lmax %d, nq %d)", kMaxUnrolledL,
                 c->lmax, nq);
      // the queue carries the weights too (+1 KB): all rings resident up to 8.75 KB per wave (18 waves per CU;
      // L = 6, n_q = 16 needs 8.5 KB and runs 6 % faster that way than in two groups), 8 KB groups beyond
      rows = (c->opt_ring_rows > 0) ? c->opt_ring_rows : nq;
      if (c->opt_ring_rows <= 0 && wave_lds_layout(c->lmax, nq, true, nqj).bytes > 8960) {
        const int fixed = wave_lds_layout(c->lmax, 0, true, nqj).bytes;
        rows = (8 * 1024 - fixed) / (32 * (c->lmax + 1));
      }
      const int rows_min_w = 2 + (127 + npsi - 1) / npsi;
      if (rows < rows_min_w) rows = rows_min_w;
      if (rows > nq) rows = nq;
    }
    if (split && c->opt_ring_rows <= 0) {
      // two waves per pair: a slab of one wave spans 64 / per_ring rings; two slabs' worth of rings per group, all of
      // them if the pair then stays within 20 KB (8 pairs = 16 waves per CU)
      // ... as many slabs' worth of rings per group as keep the pair within the LDS share of the waves its registers
      // allow (all rings if they fit), never fewer than two slabs' worth
      const int per_slab = (64 + per_ring - 1) / per_ring;
      int wsimd = 4;
      {
        hipFuncAttributes fa;
        if (kAttr[c->lmax](true, false, &fa, true, true, false) == hipSuccess && fa.numRegs > 0) {
          wsimd = 512 / (((fa.numRegs + 7) / 8) * 8);
          if (wsimd > 8) wsimd = 8;
          if (wsimd < 1) wsimd = 1;
        }
      }
      const int budget = (160 * 1024) / (2 * wsimd);   // bytes per pair: 4 wsimd waves per CU, two per pair
      rows = 2 * per_slab;
      if (pair_lds_layout2(c->lmax, nq, nq).bytes <= budget) rows = nq;
      else
        while (rows + per_slab <= nq && pair_lds_layout2(c->lmax, rows + per_slab, nq).bytes <= budget) rows += per_slab;
      if (rows < rows_min) rows = rows_min;
      if (rows > nq) rows = nq;
    }
    WaveLdsLayout wl = split ? pair_lds_layout2(c->lmax, rows, nq) : wave_lds_layout(c->lmax, rows, c->opt_rule != 0, nqj);
    // per-azimuth kernels: the node queue grows into what is left of the last LDS granule (queue_capacity, pair_kernel.hpp)
    int qcap = kQueue;
    if (jpoly && !c->opt_rule && c->opt_wpb <= 1 && c->opt_queue_slack) {
      qcap = queue_capacity(wl.bytes, split ? 2 : 1);
      if (qcap > kQueue) {
        const WaveLdsLayout wg = split ? pair_lds_layout2(c->lmax, rows, nq, qcap) : wave_lds_layout(c->lmax, rows, false, nqj, qcap);
        if ((wg.bytes + kLdsGranule - 1) / kLdsGranule == (wl.bytes + kLdsGranule - 1) / kLdsGranule) wl = wg;
        else qcap = kQueue;
      }
    }
    P.qcap = qcap;
    if (wl.bytes > 160 * 1024)
      CTX_FAIL(c, SHPAIR_ELMAX, "lmax %d with nq %d needs %d bytes of LDS per pair, more than a CU has", c->lmax, nq,
               wl.bytes);
    // One wave (= one pair) per workgroup: pairs differ in cost (a grazing pair leaves after phase 1),
    // and a multi-wave workgroup holds its LDS and wave slots until its slowest pair is done.
    // A/B (tools/ab_libs.py --wpb): 1 wave 4.13 ms, 2 waves 4.22, 4 waves 4.34 at L = 6.
    int wpb = 1;
    if (c->opt_wpb > 1) {
      wpb = c->opt_wpb < kMaxWavesPerBlock ? c->opt_wpb : kMaxWavesPerBlock;
      if (wpb * wl.bytes > 160 * 1024) wpb = (160 * 1024) / wl.bytes;
    }
    if (split) wpb = 1;   // the workgroup is the pair; wave_lds_bytes its whole LDS
    P.wave_lds_bytes = wl.bytes;
    P.waves_per_block = wpb;
    P.ring_rows = rows;
    P.spec = c->opt_spec ? 1 : 0;
    c->last_lds_bytes = P.wave_lds_bytes;
    c->last_ring_rows = rows;
    c->last_qcap = qcap;
    c->last_spec = c->lmax <= kMaxUnrolledL && c->opt_variant != 1 && pair_spec_matches_rt(c->lmax, P);
  }
  P.pair_ft = nullptr;
  if (c->opt_deterministic) {
    // deterministic accumulation: reverse index (once per list), a clean per-slot buffer, stores instead of atomics
    const int nall_idx = c->max_atom_index + 1;
    HIPCHK(c, c->d_pair_ft.ensure((size_t)c->npairs * 12));
    HIPCHK(c, c->d_rev_ent.ensure((size_t)c->npairs * 2));
    HIPCHK(c, c->d_rev_start.ensure((size_t)nall_idx + 1));
    HIPCHK(c, c->d_rev_cur.ensure((size_t)nall_idx + 1));
    if (c->rev_dirty || c->rev_nall != nall_idx) {
      HIPCHK(c, hipMemsetAsync(c->d_rev_cur.p, 0, ((size_t)nall_idx + 1) * sizeof(int), st));
      hipLaunchKernelGGL(det_count_kernel, dim3((c->npairs + kDetBlock - 1) / kDetBlock), dim3(kDetBlock), 0, st, c->npairs,
                         (const int*)c->d_pair_i.p, (const int*)c->d_pair_j.p, nall_idx, c->d_rev_cur.p);
      HIPCHK(c, hipGetLastError());
      {
        const int rc = shstep_exclusive_scan(c, c->d_rev_cur.p, c->d_rev_start.p, nall_idx, st);
        if (rc) return rc;
      }
      HIPCHK(c, hipMemsetAsync(c->d_rev_cur.p, 0, ((size_t)nall_idx + 1) * sizeof(int), st));
      hipLaunchKernelGGL(det_fill_kernel, dim3((c->npairs + kDetBlock - 1) / kDetBlock), dim3(kDetBlock), 0, st, c->npairs,
                         (const int*)c->d_pair_i.p, (const int*)c->d_pair_j.p, nall_idx, (const int*)c->d_rev_start.p,
                         c->d_rev_cur.p, c->d_rev_ent.p);
      hipLaunchKernelGGL(det_sort_kernel, dim3((nall_idx + kDetBlock - 1) / kDetBlock), dim3(kDetBlock), 0, st, nall_idx,
                         (const int*)c->d_rev_start.p, c->d_rev_ent.p);
      HIPCHK(c, hipGetLastError());
      c->rev_dirty = false;
      c->rev_nall = nall_idx;
    }
    if (pre) HIPCHK(c, hipMemsetAsync(c->d_pair_ft.p, 0, (size_t)c->npairs * 12 * sizeof(double), st));
    P.pair_ft = c->d_pair_ft.p;
  }
  P.ev = ev; P.pair_out = c->pair_out;
  P.pair_ev = nullptr;
  const int tally_blocks = (c->npairs + kTallyChunk - 1) / kTallyChunk;
  if (eflag || vflag) {
    // per-slot rows + the block sums behind them (sized with the list, shp_size_pair_buffers: nothing is allocated in a capture)
    HIPCHK(c, c->d_pair_ev.ensure(8 * ((size_t)c->npairs + (size_t)tally_blocks)));
    if (pre) HIPCHK(c, hipMemsetAsync(c->d_pair_ev.p, 0, 8 * (size_t)c->npairs * sizeof(double), st));
    P.pair_ev = c->d_pair_ev.p;
  }
  P.flags = nullptr;
  P.dbg = c->dbg;
  P.eflag = eflag ? 1 : 0; P.vflag = vflag ? 1 : 0;
  if (c->opt_count) {
    HIPCHK(c, c->d_flags.ensure(c->npairs));
    if (pre) {
      HIPCHK(c, hipMemsetAsync(c->d_counters.p, 0, 2 * sizeof(unsigned long long), st));
      HIPCHK(c, hipMemsetAsync(c->d_flags.p, 0, c->npairs, st));
    }
    P.flags = c->d_flags.p;
  }
  const bool needv = c->opt_force_volume || eflag || c->any_nonunit_exponent || c->eatom_dev != nullptr;
  c->last_needv = needv || c->opt_rule != 0;   // the template argument launched: the weighted rule has one instance, with the volume path
  // per-pair records (pair_setup.hpp); the buffers are sized when a list is installed, so nothing is allocated here
  // unless a caller swapped the list behind the context's back
  HIPCHK(c, c->d_rec.ensure((size_t)c->npairs * kRecStride));
  HIPCHK(c, c->d_rec_i.ensure((size_t)c->npairs * 4));
  P.rec = c->d_rec.p;
  P.rec_i = c->d_rec_i.p;
  if (jpoly) {   // grows only when the list or the order grew: sized by shpair_prepare_tables() ahead of a stream capture
    HIPCHK(c, c->d_rot.ensure(rot_buffer_doubles(c->lmax, 2 * (size_t)c->npairs)));
    P.rot = c->d_rot.p;
  } else {
    P.rot = nullptr;
  }
  if (c->opt_timing && pre) HIPCHK(c, hipEventRecord(c->ev0, st));
  launch_pair_setup(P, c->d_rec.p, c->d_rec_i.p, st);
  if (c->lmax <= kMaxUnrolledL && c->opt_variant != 1) {
    P.coef = c->d_coefm.p;  // compiled orders read the monomial (Horner) table
    kLaunch[c->lmax](P, needv, st, c->pre_contact_wait);
  } else {
    shp_launch_Lrt(P, needv, st, c->pre_contact_wait);
  }
  HIPCHK(c, hipGetLastError());
  if (!post) return SHPAIR_OK;
  if (c->opt_deterministic) {
    const int nall_idx = c->max_atom_index + 1;
    hipLaunchKernelGGL(det_gather_kernel, dim3((6 * nall_idx + kDetBlock - 1) / kDetBlock), dim3(kDetBlock), 0, st, nall_idx,
                       (const int*)c->d_rev_start.p, (const int*)c->d_rev_ent.p, (const double*)c->d_pair_ft.p, f, torque);
    HIPCHK(c, hipGetLastError());
  }
  if (eflag || vflag) {
    double* part = c->d_pair_ev.p + 8 * (size_t)c->npairs;
    hipLaunchKernelGGL(tally_partial_kernel, dim3(tally_blocks), dim3(kTallyBlock), 0, st, c->npairs, (const double*)c->d_pair_ev.p, part);
    hipLaunchKernelGGL(tally_final_kernel, dim3(1), dim3(kTallyBlock), 0, st, tally_blocks, (const double*)part, ev, eflag ? 1 : 0,
                       vflag ? 1 : 0);
    HIPCHK(c, hipGetLastError());
  }
  if (c->opt_timing) {
    HIPCHK(c, hipEventRecord(c->ev1, st));
    c->timed_last = true;
  }
  if (c->opt_count) {
    hipLaunchKernelGGL(count_flags_kernel, dim3((c->npairs + 255) / 256), dim3(256), 0, st, c->d_flags.p, c->npairs,
                       c->d_counters.p);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipMemcpyAsync(c->h_counters, c->d_counters.p, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, st));
    HIPCHK(c, hipEventRecord(c->evB, st));
    c->counted_last = true;
  }
  return SHPAIR_OK;
}

int shpair_compute(shpair_ctx* c, int nlocal, int nghost, const double* x, const double* quat, const int* type,
                   const int* shtype, int newton_pair, int eflag, int vflag, double* f, double* torque,
                   double* eng_vdwl, double* virial)
{
  if (!c) return SHPAIR_EINVAL;
  if (nlocal < 0 || nghost < 0) CTX_FAIL(c, SHPAIR_EINVAL, "negative atom counts");
  if (!c->have_neighbors) CTX_FAIL(c, SHPAIR_ESTATE, "no neighbour list: call shpair_set_neighbors() first");
  const size_t nall = (size_t)nlocal + (size_t)nghost;
  if (nall == 0 || c->npairs == 0) {
    c->stats.n_candidates = c->npairs;
    c->timed_last = c->counted_last = false;
    return SHPAIR_OK;
  }
  if (!x || !quat || !type || !shtype || !f || !torque) CTX_FAIL(c, SHPAIR_EINVAL, "null atom array");
  if (eflag && !eng_vdwl) CTX_FAIL(c, SHPAIR_EINVAL, "eflag set but eng_vdwl is null");
  if (vflag && !virial) CTX_FAIL(c, SHPAIR_EINVAL, "vflag set but virial is null");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = c->stream;
  HIPCHK(c, c->d_x.ensure(3 * nall));
  HIPCHK(c, c->d_quat.ensure(4 * nall));
  HIPCHK(c, c->d_type.ensure(nall));
  HIPCHK(c, c->d_shtype.ensure(nall));
  HIPCHK(c, c->d_f.ensure(3 * nall));
  HIPCHK(c, c->d_torque.ensure(3 * nall));
  // Types and shape indices are not range-checked on the host any more (an O(nall) scan per step): the kernel guards
  // its table reads and raises an error bit, which this call reads back below and reports.  They are uploaded every
  // call: LAMMPS may change a type without reneighbouring (fix atom/swap).
  HIPCHK(c, hipMemcpyAsync(c->d_type.p, type, nall * sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_shtype.p, shtype, nall * sizeof(int), hipMemcpyHostToDevice, st));
  // per-atom tallies of the host form: staged like the forces (shpair_set_peratom_host)
  // the staged arrays stand in for the device-form pointers for the duration of this call, whatever way it ends
  struct Restore {
    shpair_ctx* c;
    double *e, *v;
    ~Restore() { c->eatom_dev = e; c->vatom_dev = v; }
  } restore{c, c->eatom_dev, c->vatom_dev};
  const bool pe = c->eatom_host != nullptr, pv = c->vatom_host != nullptr;
  if (pe || pv) {
    if (pe) HIPCHK(c, c->d_eatom.ensure(nall));
    if (pv) HIPCHK(c, c->d_vatom.ensure(6 * nall));
    // ADD semantics without a host pass: the caller's values go up, the kernel adds to them, the sums come back
    if (pe) HIPCHK(c, hipMemcpyAsync(c->d_eatom.p, c->eatom_host, nall * sizeof(double), hipMemcpyHostToDevice, st));
    if (pv) HIPCHK(c, hipMemcpyAsync(c->d_vatom.p, c->vatom_host, 6 * nall * sizeof(double), hipMemcpyHostToDevice, st));
    c->eatom_dev = pe ? c->d_eatom.p : nullptr;
    c->vatom_dev = pv ? c->d_vatom.p : nullptr;
  }
  HIPCHK(c, hipEventRecord(c->evA, st));
  HIPCHK(c, hipMemcpyAsync(c->d_x.p, x, 3 * nall * sizeof(double), hipMemcpyHostToDevice, st));
  HIPCHK(c, hipMemcpyAsync(c->d_quat.p, quat, 4 * nall * sizeof(double), hipMemcpyHostToDevice, st));
  // f and torque are ADDED to (another pair style of a hybrid run, or a pre_force fix, may have been there first): they
  // travel up, the kernel accumulates into them on the device, and the sums overwrite the host arrays.  Measured at
  // 100k atoms against staging zeros and adding on the host (interleaved runs, tools/gpu_check.py): call wall time
  // minus kernel time 0.45 ms instead of 0.52 ms; 16 MB cross PCIe per call either way.
  // They go up on a second stream, beside the set-up and rotation kernels, which do not touch them: the contact kernel
  // (its epilogue's atomics; the gather of the deterministic mode) waits for the event.  With the caller's arrays
  // registered (shpair_pin_host) the copies are true asynchronous DMA and the overlap is real; pageable memory is
  // staged by the runtime and mostly serialises.
  HIPCHK(c, hipEventRecord(c->ev_up, st));                 // the previous call's read-back of d_f / d_torque is long done;
  HIPCHK(c, hipStreamWaitEvent(c->stream_up, c->ev_up, 0));   // orders the second stream behind this one all the same
  HIPCHK(c, hipMemcpyAsync(c->d_f.p, f, 3 * nall * sizeof(double), hipMemcpyHostToDevice, c->stream_up));
  HIPCHK(c, hipMemcpyAsync(c->d_torque.p, torque, 3 * nall * sizeof(double), hipMemcpyHostToDevice, c->stream_up));
  HIPCHK(c, hipEventRecord(c->ev_up, c->stream_up));
  HIPCHK(c, hipMemsetAsync(c->d_ev.p, 0, 7 * sizeof(double), st));
  c->pre_contact_wait = c->ev_up;
  const int rc = shpair_compute_device(c, nlocal, nghost, c->d_x.p, c->d_quat.p, c->d_type.p, c->d_shtype.p,
                                       newton_pair, eflag, vflag, c->d_f.p, c->d_torque.p, c->d_ev.p, st);
  c->pre_contact_wait = nullptr;
  HIPCHK(c, hipStreamWaitEvent(st, c->ev_up, 0));   // whatever path the launch took (no pairs, an early error): st is behind the uploads
  if (rc) {
    (void)hipStreamSynchronize(st);   // stream_up may still be reading the caller's f / torque: not after this call has returned
    return rc;
  }
  if (pe) HIPCHK(c, hipMemcpyAsync(c->eatom_host, c->d_eatom.p, nall * sizeof(double), hipMemcpyDeviceToHost, st));
  if (pv) HIPCHK(c, hipMemcpyAsync(c->vatom_host, c->d_vatom.p, 6 * nall * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(f, c->d_f.p, 3 * nall * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(torque, c->d_torque.p, 3 * nall * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(c->h_ev, c->d_ev.p, 7 * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipMemcpyAsync(c->h_err, c->d_err.p, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipEventRecord(c->evB, st));
  HIPCHK(c, hipStreamSynchronize(st));
  c->total_timed_last = true;
  if (*c->h_err) {
    const int bits = *c->h_err;
    HIPCHK(c, hipMemsetAsync(c->d_err.p, 0, sizeof(int), st));
    if (bits & (kPairErrShape | kPairErrType))
      CTX_FAIL(c, SHPAIR_EINVAL, "an atom %s outside its table reached the pair kernel (those pairs were skipped): types or shape "
               "indices changed without a new neighbour list?", (bits & kPairErrShape) ? "shape index" : "type");
    CTX_FAIL(c, SHPAIR_EINVAL, "coincident centres: a listed pair has separation 0 (or a position that is not a number); it was "
             "skipped (docs/SPEC.md 2, step 1)");
  }
  if (eflag) *eng_vdwl += c->h_ev[0];
  if (vflag)
    for (int a = 0; a < 6; ++a) virial[a] += c->h_ev[1 + a];
  return SHPAIR_OK;
}

int shpair_get_kernel_info(shpair_ctx* c, shpair_kernel_info* out)
{
  if (!c || !out) return SHPAIR_EINVAL;
  if (c->lmax < 0 || c->last_lds_bytes <= 0) CTX_FAIL(c, SHPAIR_ESTATE, "kernel info: no compute has run yet");
  HIPCHK(c, hipSetDevice(c->device));
  hipFuncAttributes a;
  const bool compiled = c->lmax <= kMaxUnrolledL && c->opt_variant != 1;
  HIPCHK(c, compiled ? kAttr[c->lmax](c->last_needv, c->opt_rule != 0, &a, c->last_jpoly, c->last_split, c->last_spec)
                     : shp_attr_Lrt(c->last_needv, false, &a, false, false, false));
  out->lmax = c->lmax;
  out->compiled_order = compiled ? 1 : 0;
  out->vgprs = a.numRegs;
  out->scratch_bytes = (int)a.localSizeBytes;
  const int wpp = (compiled && c->last_split) ? 2 : 1;   // waves per pair
  out->lds_bytes_per_wave = c->last_lds_bytes / wpp;
  out->waves_per_pair = wpp;
  out->ring_rows = c->last_ring_rows;
  out->queue_entries = c->last_qcap;
  // gfx950: 512 VGPRs per SIMD lane in blocks of 8, at most 8 waves per SIMD, 160 KiB LDS per CU of 4 SIMDs
  const int vg = ((a.numRegs + 7) / 8) * 8;
  int w = vg > 0 ? 512 / vg : 8;
  if (w > 8) w = 8;
  // LDS is allocated in granules of 1 280 B (160 KB / 128; measured: +448 B on 8 512 B is free, +512 B costs two waves,
  // profiles/r04_ac_lds_granule.txt)
  const int lds_alloc = ((c->last_lds_bytes + 1279) / 1280) * 1280;
  const int by_lds = wpp * ((160 * 1024) / lds_alloc);  // workgroups (= pairs) per CU x waves per pair
  out->waves_per_simd_vgpr = w;
  out->waves_per_cu_lds = by_lds;
  const int cu = (4 * w < by_lds) ? 4 * w : by_lds;
  out->waves_per_cu = cu;
  out->family = (compiled && c->last_jpoly) ? 1 : 0;
  out->needv = c->last_needv ? 1 : 0;
  out->weighted = (compiled && c->opt_rule != 0) ? 1 : 0;
  out->specialised = (compiled && c->last_spec) ? 1 : 0;
  return SHPAIR_OK;
}

// Page-locks a caller-owned host array for the host-pointer entry point (hipHostRegister): hipMemcpyAsync of a
// registered range is a direct DMA at PCIe rate instead of the runtime's staged copy of pageable memory.
int shpair_pin_host(shpair_ctx* c, void* ptr, size_t bytes)
{
  if (!c) return SHPAIR_EINVAL;
  if (!ptr || bytes == 0) CTX_FAIL(c, SHPAIR_EINVAL, "pin_host: null pointer or zero size");
  HIPCHK(c, hipSetDevice(c->device));
  for (auto& pr : c->pinned)
    if (pr.first == ptr) {
      if (pr.second == bytes) return SHPAIR_OK;
      (void)hipHostUnregister(ptr);   // same start, another length: the array was reallocated in place
      (void)hipGetLastError();
      pr = c->pinned.back();
      c->pinned.pop_back();
      break;
    }
  const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    CTX_FAIL(c, SHPAIR_EHIP, "hipHostRegister(%p, %zu) failed: %s (the copies fall back to the runtime's staging)", ptr, bytes,
             hipGetErrorString(e));
  }
  c->pinned.emplace_back(ptr, bytes);
  return SHPAIR_OK;
}

int shpair_unpin_host(shpair_ctx* c, void* ptr)
{
  if (!c) return SHPAIR_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
  for (size_t k = 0; k < c->pinned.size(); ++k)
    if (c->pinned[k].first == ptr) {
      (void)hipHostUnregister(ptr);
      (void)hipGetLastError();
      c->pinned[k] = c->pinned.back();
      c->pinned.pop_back();
      return SHPAIR_OK;
    }
  CTX_FAIL(c, SHPAIR_EINVAL, "unpin_host: %p was not pinned through this context", ptr);
}

int shpair_set_peratom_output(shpair_ctx* c, double* eatom_dev, double* vatom_dev)
{
  if (!c) return SHPAIR_EINVAL;
  c->eatom_dev = eatom_dev;
  c->vatom_dev = vatom_dev;
  return SHPAIR_OK;
}

int shpair_set_peratom_host(shpair_ctx* c, double* eatom, double* vatom)
{
  if (!c) return SHPAIR_EINVAL;
  c->eatom_host = eatom;
  c->vatom_host = vatom;
  return SHPAIR_OK;
}

int shpair_set_option(shpair_ctx* c, const char* key, int value)
{
  if (!c || !key) return SHPAIR_EINVAL;
  if (!strcmp(key, "force_volume")) c->opt_force_volume = value ? 1 : 0;
  else if (!strcmp(key, "timing")) c->opt_timing = value ? 1 : 0;
  else if (!strcmp(key, "count")) c->opt_count = value ? 1 : 0;
  else if (!strcmp(key, "variant")) c->opt_variant = value;
  else if (!strcmp(key, "rule")) {
    if (value != 0 && value != 1) CTX_FAIL(c, SHPAIR_EINVAL, "rule %d is neither 0 (sharp) nor 1 (weighted)", value);
    c->opt_rule = value;
  }
  else if (!strcmp(key, "ring_rows")) c->opt_ring_rows = value;
  else if (!strcmp(key, "jpoly")) c->opt_jpoly = value;
  else if (!strcmp(key, "split")) c->opt_split = value;
  else if (!strcmp(key, "deterministic")) {
    c->opt_deterministic = value ? 1 : 0;
    c->rev_dirty = true;
  }
  else if (!strcmp(key, "waves_per_block")) c->opt_wpb = value;
  else if (!strcmp(key, "queue_slack")) c->opt_queue_slack = value ? 1 : 0;
  else if (!strcmp(key, "spec")) c->opt_spec = value != 0;
  else if (!strcmp(key, "halo_overlap")) c->opt_overlap = value <= 0 ? 0 : (value >= 2 ? 2 : 1);
  else if (!strcmp(key, "halo_stream_priority")) c->opt_halo_prio = value != 0;   // takes effect at the next shhalo_run_device (both kinds of stream are kept)
  else CTX_FAIL(c, SHPAIR_EINVAL, "unknown option '%s'", key);
  return SHPAIR_OK;
}

int shpair_get_stats(shpair_ctx* c, shpair_stats* out)
{
  if (!c || !out) return SHPAIR_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  c->stats.kernel_ms = 0.0;
  c->stats.total_ms = 0.0;
  c->stats.n_contact = -1;
  c->stats.n_touching = -1;
  if (c->timed_last) {
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0.f;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    c->stats.kernel_ms = ms;
  }
  if (c->counted_last) {
    HIPCHK(c, hipEventSynchronize(c->evB));
    c->stats.n_contact = (long long)c->h_counters[0];
    c->stats.n_touching = (long long)c->h_counters[1];
  }
  if (c->total_timed_last) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->evA, c->evB) == hipSuccess) c->stats.total_ms = ms;
  }
  *out = c->stats;
  if (c->timed_last || c->counted_last) {
    // the compute these numbers belong to has finished: report what its kernel could not index
    HIPCHK(c, hipDeviceSynchronize());
    return shpair_check_device_errors(c, c->stream);
  }
  return SHPAIR_OK;
}

int shpair_set_pair_output(shpair_ctx* c, double* pair_out_dev)
{
  if (!c) return SHPAIR_EINVAL;
  c->pair_out = pair_out_dev;
  return SHPAIR_OK;
}

// Not part of include/shpair.h: work counters of SHP_STATS diagnostic builds.
int shpair_debug_set_counters(shpair_ctx* c, unsigned long long* dbg_dev)
{
  if (!c) return SHPAIR_EINVAL;
  c->dbg = dbg_dev;
  return SHPAIR_OK;
}

int shpair_fp64_peak(shpair_ctx* c, int mode, double target_ms, double* valu_tflops, double* mfma_tflops)
{
  if (!c) return SHPAIR_EINVAL;
  if (mode < 0 || mode > 2 || !(target_ms > 0.0) || target_ms > 2000.0)
    CTX_FAIL(c, SHPAIR_EINVAL, "fp64_peak: mode %d not in 0..2 or target_ms %g not in (0, 2000]", mode, target_ms);
  HIPCHK(c, hipSetDevice(c->device));
  Fp64PeakResult r;
  HIPCHK(c, fp64_peak_run(mode, target_ms, 5, &r, c->stream));
  if (valu_tflops) *valu_tflops = r.valu_tflops;
  if (mfma_tflops) *mfma_tflops = r.mfma_tflops;
  return SHPAIR_OK;
}

int shpair_get_stream(shpair_ctx* c, void** stream)
{
  if (!c || !stream) return SHPAIR_EINVAL;
  *stream = (void*)c->stream;
  return SHPAIR_OK;
}

int shpair_synchronize(shpair_ctx* c)
{
  if (!c) return SHPAIR_EINVAL;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return shpair_check_device_errors(c, c->stream);
}

}  // extern "C"
