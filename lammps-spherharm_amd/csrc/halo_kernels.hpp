// halo_kernels.hpp — gfx950 kernels of the N > 1 path (include/shhalo.h): plan building (stable multi-list
// partition of the owned rows), packing and unpacking of the forward / reverse / migration messages.
// All of them stream a few dozen bytes per row: one lane per row, 256-lane workgroups, HBM / latency bound.
// The per-row decisions are the inline functions of halo_plan.hpp, shared with the host planner.
#pragma once
#include <hip/hip_runtime.h>

#include "halo_plan.hpp"

namespace shp {

constexpr int kHaloBlock = 256;
constexpr int kHaloWaves = kHaloBlock / 64;
constexpr int kHaloMaxSlots = 28;   // 26 directions, or 1 + 26 destinations of a migration
constexpr int kFwdWidth = 7;        // x[3] quat[4]
constexpr int kBorderWidth = 9;     // + (tag, type), (shtype, 0) as two 64-bit words
constexpr int kRevWidth = 6;        // f[3] torque[3]
constexpr int kMigWidth = 15;       // x[3] quat[4] v[3] angmom[3] + (tag, type), (shtype, mask)
constexpr int kHaloErrLost = 1;     // device flag: an owned atom left the brick and its 26 neighbours

// What the kernels need of the plan, by value.
struct HaloSlots {
  int nslots;
  int code_of_slot[kHaloMaxSlots];   // ghost lists: direction code of send slot s, slots ordered by (peer, code)
  int cat_of_code[27];               // migration: category of a destination direction (0 = stays)
};

struct HaloMsgTables {
  double shift[27][3];
  int self[27];        // the direction leads back to this rank (undecomposed periodic dimension)
  int send_off[27];    // first row of the direction's block in the send buffer
  int recv_off[27];    // first ghost row (relative to nlocal) of what arrives through the direction
  int recv_cnt[27];
};

__device__ inline double pack2i(int a, int b)
{
  const unsigned long long u = ((unsigned long long)(unsigned)b << 32) | (unsigned)a;
  return __longlong_as_double((long long)u);
}
__device__ inline void unpack2i(double w, int& a, int& b)
{
  const unsigned long long u = (unsigned long long)__double_as_longlong(w);
  a = (int)(unsigned)(u & 0xffffffffULL);
  b = (int)(unsigned)(u >> 32);
}

// ---- Comm::exchange, decision: wrap into the box, destination category of every owned row -------------------
__global__ __launch_bounds__(kHaloBlock) void halo_wrap_dest_kernel(int n, HaloGeom g, HaloSlots t, double* __restrict__ x,
                                                                     unsigned char* __restrict__ cat, int* __restrict__ flags)
{
  const int i = blockIdx.x * kHaloBlock + threadIdx.x;
  if (i >= n) return;
  double xi[3];
  for (int d = 0; d < 3; ++d) {
    const double p = x[3 * i + d], w = halo_wrap(g, d, p);
    if (w != p) x[3 * i + d] = w;
    xi[d] = w;
  }
  const int code = halo_dest_code(g, xi);
  int c = 0;
  if (code < 0 || (code != 13 && g.peer[code] < 0)) {
    // through an open boundary the outermost brick keeps the atom (halo_brick_coord clamps), so this is a row
    // that moved further than one brick since the last exchange
    atomicOr(flags, kHaloErrLost);
  } else if (code != 13) {
    c = t.cat_of_code[code];
  }
  cat[i] = (unsigned char)c;
}

// ---- stable partition of the owned rows into `nslots` lists ----------------------------------------------------
// MODE 0: membership from the ghost mask of x (a row may be in several lists); MODE 1: list cat[i] only.
// Pass 1 counts per (slot, workgroup); an exclusive scan over the slot-major array gives every (slot, workgroup)
// its first position in the concatenation of the lists; pass 2 writes the row indices, ascending within a list.
template <int MODE>
__device__ inline unsigned halo_membership(const HaloGeom& g, const HaloSlots& t, int i, int n, const double* __restrict__ x,
                                           const unsigned char* __restrict__ cat)
{
  if (i >= n) return 0u;
  if (MODE == 1) return 1u << cat[i];
  const double xi[3] = {x[3 * i], x[3 * i + 1], x[3 * i + 2]};
  const unsigned m = halo_ghost_mask(g, xi);
  unsigned sm = 0u;
  for (int s = 0; s < t.nslots; ++s) sm |= ((m >> t.code_of_slot[s]) & 1u) << s;
  return sm;
}

template <int MODE>
__global__ __launch_bounds__(kHaloBlock) void halo_count_kernel(int n, HaloGeom g, HaloSlots t, const double* __restrict__ x,
                                                                 const unsigned char* __restrict__ cat,
                                                                 int* __restrict__ blockcnt, int nb)
{
  __shared__ int wcnt[kHaloWaves][kHaloMaxSlots];
  const int i = blockIdx.x * kHaloBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned sm = halo_membership<MODE>(g, t, i, n, x, cat);
  for (int s = 0; s < t.nslots; ++s) {
    const unsigned long long b = __ballot((sm >> s) & 1u);
    if (lane == 0) wcnt[wv][s] = __popcll(b);
  }
  __syncthreads();
  if ((int)threadIdx.x < t.nslots) {
    int c = 0;
    for (int w = 0; w < kHaloWaves; ++w) c += wcnt[w][threadIdx.x];
    blockcnt[(size_t)threadIdx.x * nb + blockIdx.x] = c;
  }
}

template <int MODE>
__global__ __launch_bounds__(kHaloBlock) void halo_fill_kernel(int n, HaloGeom g, HaloSlots t, const double* __restrict__ x,
                                                                const unsigned char* __restrict__ cat,
                                                                const int* __restrict__ start, int nb, int* __restrict__ out_idx,
                                                                unsigned char* __restrict__ out_code)
{
  __shared__ int wcnt[kHaloWaves][kHaloMaxSlots];
  const int i = blockIdx.x * kHaloBlock + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const unsigned sm = halo_membership<MODE>(g, t, i, n, x, cat);
  for (int s = 0; s < t.nslots; ++s) {
    const unsigned long long b = __ballot((sm >> s) & 1u);
    if (lane == 0) wcnt[wv][s] = __popcll(b);
  }
  __syncthreads();
  for (int s = 0; s < t.nslots; ++s) {
    const bool in = (sm >> s) & 1u;
    const unsigned long long b = __ballot(in);
    if (!in) continue;
    int pos = start[(size_t)s * nb + blockIdx.x] + (int)__popcll(b & ((1ULL << lane) - 1ULL));
    for (int w = 0; w < wv; ++w) pos += wcnt[w][s];
    out_idx[pos] = i;
    if (out_code) out_code[pos] = (unsigned char)(MODE == 0 ? t.code_of_slot[s] : s);
  }
}

// Per-list totals from the scanned (slot, workgroup) array: totals[s]; and, for every remote peer k, the 27-entry
// count vector it is sent: msg[k][code] = my count of the direction `code` if that direction leads to the peer.
__global__ void halo_totals_kernel(HaloSlots t, const int* __restrict__ start, int nb, int mode, int npeers,
                                   const int* __restrict__ peer_of_slot /* index into the remote peers, -1: self */,
                                   int* __restrict__ totals, int* __restrict__ msg)
{
  const int s = threadIdx.x;
  if (s >= t.nslots) return;
  const int c = start[(size_t)(s + 1) * nb] - start[(size_t)s * nb];
  totals[s] = c;
  if (mode == 0) {
    const int k = peer_of_slot[s];
    if (k >= 0) msg[k * 27 + t.code_of_slot[s]] = c;
  } else if (s >= 1) {
    msg[(s - 1) * 27] = c;   // migration: one count per remote peer
  }
  (void)npeers;
}

// ---- forward: owners' rows -> send buffer (remote peers) or straight into this rank's receive buffer ---------
template <int W>
__global__ __launch_bounds__(kHaloBlock) void halo_pack_kernel(int nsend, HaloMsgTables T, const int* __restrict__ send_idx,
                                                                const unsigned char* __restrict__ send_code,
                                                                const double* __restrict__ x, const double* __restrict__ quat,
                                                                const int* __restrict__ tag, const int* __restrict__ type,
                                                                const int* __restrict__ shtype, double* __restrict__ sendbuf,
                                                                double* __restrict__ recvbuf)
{
  const int e = blockIdx.x * kHaloBlock + threadIdx.x;
  if (e >= nsend) return;
  const int i = send_idx[e], c = send_code[e];
  double* o = T.self[c] ? recvbuf + (size_t)W * (T.recv_off[26 - c] + (e - T.send_off[c])) : sendbuf + (size_t)W * e;
  for (int d = 0; d < 3; ++d) o[d] = x[3 * i + d] + T.shift[c][d];
  for (int k = 0; k < 4; ++k) o[3 + k] = quat[4 * i + k];
  if (W == kBorderWidth) {
    o[7] = pack2i(tag[i], type[i]);
    o[8] = pack2i(shtype[i], 0);
  }
}

template <int W>
__global__ __launch_bounds__(kHaloBlock) void halo_unpack_kernel(int nghost, int nlocal, const double* __restrict__ recvbuf,
                                                                  double* __restrict__ x, double* __restrict__ quat,
                                                                  int* __restrict__ tag, int* __restrict__ type,
                                                                  int* __restrict__ shtype)
{
  const int g = blockIdx.x * kHaloBlock + threadIdx.x;
  if (g >= nghost) return;
  const double* r = recvbuf + (size_t)W * g;
  const int row = nlocal + g;
  for (int d = 0; d < 3; ++d) x[3 * row + d] = r[d];
  for (int k = 0; k < 4; ++k) quat[4 * row + k] = r[3 + k];
  if (W == kBorderWidth) {
    int a, b;
    unpack2i(r[7], a, b);
    tag[row] = a;
    type[row] = b;
    unpack2i(r[8], a, b);
    shtype[row] = a;
  }
}

// ---- reverse: ghost rows of f / torque -> the buffer that travels back (or straight into this rank's own
// receive buffer, at the send-list position of the row's owner) ------------------------------------------------
__global__ __launch_bounds__(kHaloBlock) void halo_rpack_kernel(int nghost, int nlocal, HaloMsgTables T,
                                                                 const double* __restrict__ f, const double* __restrict__ tq,
                                                                 double* __restrict__ rsend, double* __restrict__ rrecv)
{
  const int g = blockIdx.x * kHaloBlock + threadIdx.x;
  if (g >= nghost) return;
  // the direction block this ghost row arrived through
  int k = -1;
  for (int c = 0; c < 27; ++c)
    if (g >= T.recv_off[c] && g < T.recv_off[c] + T.recv_cnt[c]) k = c;
  const int row = nlocal + g;
  double* o = rsend + (size_t)kRevWidth * g;
  if (k >= 0 && T.self[k]) o = rrecv + (size_t)kRevWidth * (T.send_off[26 - k] + (g - T.recv_off[k]));
  for (int d = 0; d < 3; ++d) {
    o[d] = f[3 * row + d];
    o[3 + d] = tq[3 * row + d];
  }
}

__global__ __launch_bounds__(kHaloBlock) void halo_runpack_kernel(int nsend, const int* __restrict__ send_idx,
                                                                   const double* __restrict__ rrecv, double* __restrict__ f,
                                                                   double* __restrict__ tq)
{
  const int e = blockIdx.x * kHaloBlock + threadIdx.x;
  if (e >= nsend) return;
  const int i = send_idx[e];
  const double* r = rrecv + (size_t)kRevWidth * e;
  for (int d = 0; d < 3; ++d) {
    if (r[d] != 0.0) atomicAdd(&f[3 * i + d], r[d]);
    if (r[3 + d] != 0.0) atomicAdd(&tq[3 * i + d], r[3 + d]);
  }
}

// The same for ONE direction's block of send rows [off, off + n): an owner appears at most once per direction, so plain
// adds have a single writer per row — no atomics; the 26 blocks are launched one after the other in code order, which
// fixes the order in which an owner's contributions from its images are added (the deterministic mode, det_kernels.hpp).
__global__ __launch_bounds__(kHaloBlock) void halo_runpack_block_kernel(int n, int off, const int* __restrict__ send_idx,
                                                                         const double* __restrict__ rrecv, double* __restrict__ f,
                                                                         double* __restrict__ tq)
{
  const int t = blockIdx.x * kHaloBlock + threadIdx.x;
  if (t >= n) return;
  const int e = off + t;
  const int i = send_idx[e];
  const double* r = rrecv + (size_t)kRevWidth * e;
  for (int d = 0; d < 3; ++d) {
    f[3 * i + d] += r[d];
    tq[3 * i + d] += r[3 + d];
  }
}

// ---- migration rows ----------------------------------------------------------------------------------------------
struct HaloArrays {
  double *x, *v, *quat, *angmom;
  int *type, *shtype, *mask, *tag;
};

// rows[r] = the owned row order[r] (order = the partition: stayers first, then the leavers peer by peer)
__global__ __launch_bounds__(kHaloBlock) void halo_mig_gather_kernel(int n, const int* __restrict__ order, HaloArrays a,
                                                                      double* __restrict__ rows)
{
  const int r = blockIdx.x * kHaloBlock + threadIdx.x;
  if (r >= n) return;
  const int i = order[r];
  double* o = rows + (size_t)kMigWidth * r;
  for (int d = 0; d < 3; ++d) {
    o[d] = a.x[3 * i + d];
    o[7 + d] = a.v[3 * i + d];
    o[10 + d] = a.angmom[3 * i + d];
  }
  for (int k = 0; k < 4; ++k) o[3 + k] = a.quat[4 * i + k];
  o[13] = pack2i(a.tag[i], a.type[i]);
  o[14] = pack2i(a.shtype[i], a.mask[i]);
}

// owned rows first .. first + count - 1 = rows[0 .. count - 1]
__global__ __launch_bounds__(kHaloBlock) void halo_mig_scatter_kernel(int count, int first, const double* __restrict__ rows,
                                                                       HaloArrays a)
{
  const int r = blockIdx.x * kHaloBlock + threadIdx.x;
  if (r >= count) return;
  const int i = first + r;
  const double* o = rows + (size_t)kMigWidth * r;
  for (int d = 0; d < 3; ++d) {
    a.x[3 * i + d] = o[d];
    a.v[3 * i + d] = o[7 + d];
    a.angmom[3 * i + d] = o[10 + d];
  }
  for (int k = 0; k < 4; ++k) a.quat[4 * i + k] = o[3 + k];
  int p, q;
  unpack2i(o[13], p, q);
  a.tag[i] = p;
  a.type[i] = q;
  unpack2i(o[14], p, q);
  a.shtype[i] = p;
  a.mask[i] = q;
}

// rebuild flag of this rank (shstep check) | error bits -> one int the all-reduce takes the maximum of
__global__ void halo_flag_merge_kernel(const int* __restrict__ moved, int* __restrict__ out)
{
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = moved[0] ? 1 : 0;
}

}  // namespace shp
