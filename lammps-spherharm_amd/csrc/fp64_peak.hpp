// fp64_peak.hpp — the FP64 ceilings of the box, measured instead of trusted ("measure with a
// v_fma_f64 microbenchmark on the box rather than trust this").
//
// Three loops, HIP-event timed, every CU busy with 8 waves per SIMD (2048 threads per CU):
//   mode 0  VALU   independent v_fma_f64 chains (8 per lane)                      2 FLOP per lane and instruction
//   mode 1  MFMA   v_mfma_f64_16x16x4_f64 on 4 independent accumulators           2048 FLOP per wave instruction
//   mode 2  both   waves 0-3 of a 512-thread workgroup run the VALU loop, waves 4-7 (their SIMD partners,
//                  MI355X_MICROARCH.md "Two waves per SIMD") the MFMA loop: what the two pipes deliver side by side
// The pair kernel is priced against mode 0 (it issues no MFMA); modes 1 and 2 say what an MFMA reformulation
// of part of the work could add at best.
#pragma once
#include <hip/hip_runtime.h>

namespace shp {

typedef double peak_d4 __attribute__((ext_vector_type(4)));

constexpr int kPeakChains = 8;
constexpr int kPeakInner = 64;  // unrolled FMAs per chain and outer iteration

__device__ __forceinline__ void peak_valu_loop(int iters, double b, double c, double* sink, int tid)
{
  double a[kPeakChains];
#pragma unroll
  for (int k = 0; k < kPeakChains; ++k) a[k] = 1.0 + 1e-3 * (double)(k + tid);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < kPeakInner; ++u) {
#pragma unroll
      for (int k = 0; k < kPeakChains; ++k) a[k] = __builtin_fma(a[k], b, c);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < kPeakChains; ++k) s += a[k];
  if (s == 123.456) sink[tid] = s;  // never true for the inputs used; keeps the chains alive
}

__device__ __forceinline__ void peak_mfma_loop(int iters, double b, double c, double* sink, int tid)
{
  peak_d4 acc[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) acc[k] = peak_d4{c, c + 1.0, c + 2.0, c + (double)k};
  const double av = 1.0 + 1e-6 * (double)(tid & 63), bv = b;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < kPeakInner / 4; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[k], 0, 0, 0);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 123.456) sink[tid] = s;
}

// MODE 0 VALU, 1 MFMA, 2 waves 0-3 VALU + waves 4-7 MFMA
template <int MODE>
__global__ void __launch_bounds__(512) fp64_peak_kernel(int iters, double b, double c, double* sink)
{
  const int tid = blockIdx.x * 512 + threadIdx.x;
  if (MODE == 0) peak_valu_loop(iters, b, c, sink, tid);
  else if (MODE == 1) peak_mfma_loop(iters, b, c, sink, tid);
  else {
    if ((threadIdx.x >> 6) < 4) peak_valu_loop(iters, b, c, sink, tid);
    else peak_mfma_loop(iters, b, c, sink, tid);
  }
}

struct Fp64PeakResult {
  double valu_tflops = 0, mfma_tflops = 0;  // of the waves running that loop
  double ms = 0;
  int cus = 0;
  double clock_mhz = 0;  // hipDeviceProp_t::clockRate (the nominal maximum, not the clock held under load)
};

// Runs `mode` for about `target_ms` per timed launch (best of `reps`). Returns a hipError_t.
inline hipError_t fp64_peak_run(int mode, double target_ms, int reps, Fp64PeakResult* out, hipStream_t st = nullptr)
{
  hipDeviceProp_t prop;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return e;
  const int cus = prop.multiProcessorCount;
  const int blocks = cus * 4 * 4;  // 4 resident 512-thread workgroups per CU (8 waves per SIMD), 4 rounds
  double* sink = nullptr;
  e = hipMalloc((void**)&sink, (size_t)blocks * 512 * sizeof(double));
  if (e != hipSuccess) return e;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  auto launch = [&](int iters) {
    const double b = 0.999999, c = 1e-6;
    if (mode == 0) hipLaunchKernelGGL(fp64_peak_kernel<0>, dim3(blocks), dim3(512), 0, st, iters, b, c, sink);
    else if (mode == 1) hipLaunchKernelGGL(fp64_peak_kernel<1>, dim3(blocks), dim3(512), 0, st, iters, b, c, sink);
    else hipLaunchKernelGGL(fp64_peak_kernel<2>, dim3(blocks), dim3(512), 0, st, iters, b, c, sink);
  };
  auto timed = [&](int iters, float* ms) -> hipError_t {
    (void)hipEventRecord(e0, st);
    launch(iters);
    (void)hipEventRecord(e1, st);
    hipError_t r = hipEventSynchronize(e1);
    if (r != hipSuccess) return r;
    return hipEventElapsedTime(ms, e0, e1);
  };
  float ms = 0.f;
  int iters = 64;
  e = timed(iters, &ms);                    // warm-up / clock ramp
  for (int k = 0; k < 6 && e == hipSuccess; ++k) e = timed(iters, &ms);
  if (e == hipSuccess && ms > 0.f) {
    double want = target_ms / (double)ms * iters;
    if (want < 16) want = 16;
    if (want > 1e6) want = 1e6;
    iters = (int)want;
  }
  double best = 1e30;
  for (int r = 0; r < reps && e == hipSuccess; ++r) {
    e = timed(iters, &ms);
    if (e == hipSuccess && ms < best) best = ms;
  }
  if (e == hipSuccess) {
    const double threads = (double)blocks * 512.0;
    const double valu_share = mode == 0 ? 1.0 : (mode == 1 ? 0.0 : 0.5);
    const double valu_flop = threads * valu_share * (double)iters * kPeakInner * kPeakChains * 2.0;
    const double waves = threads / 64.0;
    const double mfma_flop = waves * (1.0 - valu_share) * (double)iters * kPeakInner * 2048.0;
    out->valu_tflops = valu_flop / (best * 1e-3) / 1e12;
    out->mfma_tflops = mfma_flop / (best * 1e-3) / 1e12;
    out->ms = best;
    out->cus = cus;
    out->clock_mhz = prop.clockRate / 1000.0;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return e;
}

}  // namespace shp
