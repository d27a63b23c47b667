// det_kernels.hpp — the deterministic accumulation mode (shpair_set_option "deterministic", include/shpair.h).
//
// The default scatter of a pair's force and torque is one 6-lane hardware FP64 atomic per atom (pair_kernel.hpp
// epilogue): the order in which an atom's ~12 contributions are added varies from run to run, and with it the last
// bits of f and torque (1e-16 relative per add; trajectories of a chaotic granular bed then diverge between two runs
// of the same input).  A serial CPU pair style adds in list order and is reproducible bit for bit.  In this mode the
// contact kernel WRITES each pair's twelve numbers once into a per-slot buffer and a second kernel adds, for every
// atom, its contributions in a fixed order — ascending (list slot, side) — through a reverse index that is built on
// the device whenever a list is installed:
//   count (atomic increments: counts do not depend on their order) -> exclusive scan -> fill (atomic cursor) ->
//   sort of each atom's short segment, which makes the order independent of the fill's.
// Cost per step: a 96-byte store per pair instead of two atomics, a memset of the buffer, one gather pass.
#pragma once
#include <hip/hip_runtime.h>

namespace shp {

constexpr int kDetBlock = 256;

__global__ __launch_bounds__(kDetBlock) void det_count_kernel(const int np, const int* __restrict__ pi, const int* __restrict__ pj,
                                                             const int nall, int* __restrict__ cnt)
{
  const int w = blockIdx.x * kDetBlock + threadIdx.x;
  if (w >= np) return;
  const int i = pi[w], j = pj[w];
  if ((unsigned)i < (unsigned)nall) atomicAdd(&cnt[i], 1);
  if ((unsigned)j < (unsigned)nall) atomicAdd(&cnt[j], 1);
}

__global__ __launch_bounds__(kDetBlock) void det_fill_kernel(const int np, const int* __restrict__ pi, const int* __restrict__ pj,
                                                            const int nall, const int* __restrict__ start, int* __restrict__ cur,
                                                            int* __restrict__ ent)
{
  const int w = blockIdx.x * kDetBlock + threadIdx.x;
  if (w >= np) return;
  const int i = pi[w], j = pj[w];
  if ((unsigned)i < (unsigned)nall) ent[start[i] + atomicAdd(&cur[i], 1)] = 2 * w;
  if ((unsigned)j < (unsigned)nall) ent[start[j] + atomicAdd(&cur[j], 1)] = 2 * w + 1;
}

// one lane per atom: insertion sort of its segment (a dozen entries in a packed bed)
__global__ __launch_bounds__(kDetBlock) void det_sort_kernel(const int nall, const int* __restrict__ start, int* __restrict__ ent)
{
  const int a = blockIdx.x * kDetBlock + threadIdx.x;
  if (a >= nall) return;
  const int b = start[a], e = start[a + 1];
  for (int k = b + 1; k < e; ++k) {
    const int v = ent[k];
    int q = k - 1;
    while (q >= b && ent[q] > v) {
      ent[q + 1] = ent[q];
      --q;
    }
    ent[q + 1] = v;
  }
}

// one lane per (atom, component): components 0-2 the force, 3-5 the torque; adds into f / torque like the atomics do
__global__ __launch_bounds__(kDetBlock) void det_gather_kernel(const int nall, const int* __restrict__ start, const int* __restrict__ ent,
                                                              const double* __restrict__ pair_ft, double* __restrict__ f,
                                                              double* __restrict__ torque)
{
  const int t = blockIdx.x * kDetBlock + threadIdx.x;
  if (t >= 6 * nall) return;
  const int a = t / 6, c = t - 6 * a;
  const int b = start[a], e = start[a + 1];
  if (b == e) return;
  double s = 0.0;
  for (int k = b; k < e; ++k) {
    const int v = ent[k];
    s += pair_ft[(size_t)12 * (v >> 1) + 6 * (v & 1) + c];
  }
  double* o = (c < 3) ? f + 3 * (size_t)a + c : torque + 3 * (size_t)a + (c - 3);
  *o += s;
}

// ---- global energy / virial tally: ordered reduction of the per-slot rows (pair_kernel.hpp epilogue) ----
// Stage 1: block b adds the rows of slots [b * kTallyChunk, (b + 1) * kTallyChunk) — each thread a fixed strided subset
// in ascending order, then a fixed LDS tree — and writes its 7 sums to part[b][0..6].  Stage 2 (one block) adds the block
// sums in ascending order with the same tree and ADDS the result into ev[0..6].  The partition depends on the number of
// slots only: the sums are bitwise reproducible, in the default (atomic) mode as well as in the deterministic one.
constexpr int kTallyBlock = 256;
constexpr int kTallyChunk = 8192;   // slots per block of stage 1

__device__ __forceinline__ void tally_block_tree(double (*sh)[kTallyBlock], const int t)
{
  for (int half = kTallyBlock / 2; half > 0; half >>= 1) {
    __syncthreads();
    if (t < half)
      for (int c = 0; c < 7; ++c) sh[c][t] += sh[c][t + half];
  }
  __syncthreads();
}

__global__ __launch_bounds__(kTallyBlock) void tally_partial_kernel(const int np, const double* __restrict__ pair_ev,
                                                                   double* __restrict__ part)
{
  __shared__ double sh[7][kTallyBlock];
  const int t = threadIdx.x;
  const int lo = blockIdx.x * kTallyChunk;
  const int hi = (lo + kTallyChunk < np) ? lo + kTallyChunk : np;
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int w = lo + t; w < hi; w += kTallyBlock) {
    const double4* r = (const double4*)(pair_ev + 8 * (size_t)w);
    const double4 a = r[0], b = r[1];
    s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
    s[4] += b.x; s[5] += b.y; s[6] += b.z;
  }
  for (int c = 0; c < 7; ++c) sh[c][t] = s[c];
  tally_block_tree(sh, t);
  if (t < 7) part[8 * (size_t)blockIdx.x + t] = sh[t][0];
}

__global__ __launch_bounds__(kTallyBlock) void tally_final_kernel(const int nblocks, const double* __restrict__ part,
                                                                 double* __restrict__ ev, const int eflag, const int vflag)
{
  __shared__ double sh[7][kTallyBlock];
  const int t = threadIdx.x;
  double s[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int b = t; b < nblocks; b += kTallyBlock)
    for (int c = 0; c < 7; ++c) s[c] += part[8 * (size_t)b + c];
  for (int c = 0; c < 7; ++c) sh[c][t] = s[c];
  tally_block_tree(sh, t);
  if (t == 0 && eflag) ev[0] += sh[0][0];
  if (t >= 1 && t < 7 && vflag) ev[t] += sh[t][0];
}

}  // namespace shp
