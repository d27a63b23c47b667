// shpair_ctx.hpp — the context behind the C ABI (include/shpair.h, include/shstep.h), shared by the
// translation units that implement it.  Internal: nothing here is part of the boundary.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <string>
#include <vector>

#include "../../include/shpair.h"

namespace shp {
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t n)
  {
    if (n <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = n + n / 8 + 16;
    hipError_t e = hipMalloc((void**)&p, want * sizeof(T));
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release()
  {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

struct Shape {
  int lmax = -1;
  std::vector<double> anm;
  double rmax = 0.0;
  double density = 1.0;  // shstep_set_density
};

}  // namespace shp

struct shstep_state;  // shstep_api.hip

struct shpair_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream_up = nullptr;   // host-pointer form: f / torque go up here beside the set-up and rotation kernels
  hipEvent_t ev_up = nullptr;        // ... and the contact kernel waits for this
  hipEvent_t pre_contact_wait = nullptr;   // set by shpair_compute for the duration of its call (launch_pair_contact)
  std::vector<std::pair<void*, size_t>> pinned;   // shpair_pin_host registrations
  std::string err;

  int nq = 16;
  int ntypes = 0, nshapes = 0;
  std::vector<shp::Shape> shapes;
  std::vector<double> kn, expo;
  bool tables_dirty = true, quad_dirty = true;
  bool mass_dirty = true;  // rigid-body table of shstep_api.hip
  bool any_nonunit_exponent = false;
  int lmax = -1, cstride = 0;

  shp::DevBuf<double> d_rc, d_coef, d_coefm, d_rmax, d_kn, d_expo, d_quad, d_creal, d_xval, d_gscale;
  shp::DevBuf<int> d_xcol, d_xinfo;
  shp::DevBuf<double> d_jval;   // first stage of particle j's per-azimuth polynomials (sh_tables.cpp build_jpoly_ell)
  shp::DevBuf<int> d_jcol;
  shp::DevBuf<int> d_pair_i, d_pair_j;
  shp::DevBuf<double> d_rot;  // rotated coefficient vectors of both particles of every list slot (pair_rotate_kernel)
  shp::DevBuf<double> d_rec;  // per-pair records of pair_setup.hpp, kRecStride doubles per list slot
  shp::DevBuf<int> d_rec_i;   // 4 ints per list slot
  int npairs = 0;
  int max_atom_index = -1;  // largest i or j in the uploaded list
  bool have_neighbors = false;

  // staging for the host-pointer entry point
  shp::DevBuf<double> d_x, d_quat, d_f, d_torque, d_ev;
  shp::DevBuf<int> d_type, d_shtype;
  double *h_ev = nullptr;  // pinned 7

  // neighbour lists installed so far
  unsigned long long list_gen = 0;
  // flattened LAMMPS list on its way to the device (pinned), and its device copy (expanded by expand_csr_kernel)
  int* h_list = nullptr;
  size_t h_list_cap = 0;
  shp::DevBuf<int> d_list;
  // device error bits raised by the pair kernel (pair_kernel.hpp kPairErr*), read at the blocking calls
  shp::DevBuf<int> d_err;
  int* h_err = nullptr;  // pinned
  // last output pointers that passed the device-memory check of shpair_compute_device
  const void* ok_ptr[3] = {nullptr, nullptr, nullptr};

  shp::DevBuf<unsigned long long> d_counters;
  shp::DevBuf<unsigned char> d_flags;
  unsigned long long* h_counters = nullptr;  // pinned 2

  int opt_force_volume = 0, opt_timing = 0, opt_count = 0, opt_variant = 0, opt_ring_rows = 0, opt_wpb = 0, opt_rule = 0;
  int opt_queue_slack = 1;   // "queue_slack" (diagnostic): the node queue of the per-azimuth kernels takes the rest of its last LDS granule
  int opt_overlap = 0;   // "halo_overlap" (default 0 since round 5: the exchanges and the pair kernels follow each other on the caller's
                         // stream; 1 / 2 are opt-in until a run between GPUs has measured them — bench.py --gpus N tries 2, checks it
                         // against 0 in the run itself and reports both): device-built lists are partitioned interior / boundary and
                         // shhalo_run_device runs the interior slots while the forward (2: and the reverse) exchange is in flight
  int opt_halo_prio = 0; // "halo_stream_priority": 1 = the exchange stream of "halo_overlap" is one at the highest stream priority (a hardware queue of its own; shhalo_api.hip)
  int n_interior = 0;    // slots [0, n_interior) of the installed list touch owned atoms only (device-built lists)
  int opt_jpoly = -1;      // 1 / 0: compiled orders evaluate particle j from per-azimuth polynomials or not; -1: by the
                           // measured rule (shpair_api.hip use_jpoly)
  bool last_jpoly = false;
  // deterministic accumulation (det_kernels.hpp): per-slot results + reverse index (atom -> its list slots)
  int opt_deterministic = 0;
  shp::DevBuf<double> d_pair_ft;
  shp::DevBuf<double> d_pair_ev;    // eflag / vflag: 8 doubles per slot (E, 6 virial terms, pad) + the block sums of the ordered reduce
  shp::DevBuf<int> d_rev_start, d_rev_cur, d_rev_ent;
  bool rev_dirty = true;
  int rev_nall = 0;
  int opt_split = -1;      // 1 / 0: two waves per pair (pair_kernel.hpp WPP = 2) or one; -1: by the rule use_split
  bool last_split = false;
  int last_lds_bytes = 0, last_ring_rows = 0, last_qcap = 0;  // of the last launch (shpair_get_kernel_info)
  bool last_needv = false;
  int opt_spec = 1;        // "spec": launches whose (n_q, ring rows, queue) are PairSpec<L>'s take the specialised instance (pair_kernel.hpp)
  bool last_spec = false;
  double* pair_out = nullptr;
  double *eatom_dev = nullptr, *vatom_dev = nullptr;    // shpair_set_peratom_output
  double *eatom_host = nullptr, *vatom_host = nullptr;  // shpair_set_peratom_host
  shp::DevBuf<double> d_eatom, d_vatom;                 // staging of the host form
  unsigned long long* dbg = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, evA = nullptr, evB = nullptr;
  bool timed_last = false, counted_last = false, total_timed_last = false;
  shpair_stats stats{};

  shstep_state* step = nullptr;  // integrator / borders / neighbour-build state, created on first use
};

void shstep_release_state(shpair_ctx* c);   // shstep_api.hip
void shstep_invalidate_list(shpair_ctx* c);
int shpair_prepare_tables(shpair_ctx* c);    // shpair_api.hip
int shpair_check_device_errors(shpair_ctx* c, void* stream);  // shpair_api.hip: reads + clears the kernel's error bits (blocks)
int shstep_exclusive_scan(shpair_ctx* c, const int* in, int* out, int n, void* stream);                           // shstep_api.hip
int shstep_enqueue_check(shpair_ctx* c, int nlocal, const double* x, int** flag_dev, int* forced, void* stream);  // shstep_api.hip

#define CTX_FAIL(ctx, code, ...)                         \
  do {                                                   \
    char _b[512];                                        \
    snprintf(_b, sizeof(_b), __VA_ARGS__);               \
    (ctx)->err = _b;                                     \
    return (code);                                       \
  } while (0)

#define HIPCHK(ctx, call)                                                                          \
  do {                                                                                             \
    hipError_t _e = (call);                                                                        \
    if (_e != hipSuccess)                                                                          \
      CTX_FAIL(ctx, SHPAIR_EHIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)

// shpair_api.hip: sizes the per-slot buffers of the pair kernels for a list of np slots (used by every list install)
hipError_t shp_size_pair_buffers(shpair_ctx* c, size_t np);
// the pair path over the slots [slot0, slot_end) of the installed list (shpair_api.hip; part: kPartPre | kPartPost)
enum { kPartPre = 1, kPartPost = 2 };
extern "C" int shp_compute_range(shpair_ctx* c, int nlocal, int nghost, const double* x, const double* quat, const int* type,
                                 const int* shtype, int newton_pair, int eflag, int vflag, double* f, double* torque, double* ev,
                                 void* stream, int slot0, int slot_end, int part);
