// tools/fp64_latency.hip — how many resident waves and how much instruction-level parallelism the FP64 pipe of a
// gfx950 SIMD needs: v_fma_f64 throughput for K independent dependent-chains per lane at W waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_latency.hip -o tools/fp64_latency && tools/fp64_latency
// One workgroup of 64 W threads... no: W single-wave workgroups per SIMD are placed by launching 4 W waves per CU
// (grid = CUs x 4 W workgroups of 64 lanes; LDS padding keeps more from becoming resident).  Prints a table of
// cycles per v_fma_f64 per SIMD (4 = the full rate of 16 lanes per clock).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int K>
__global__ void __launch_bounds__(64) chains(const int iters, const double a, const double b, double* out, const int lds_pad)
{
  extern __shared__ double pad[];
  double v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = a + threadIdx.x + k;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int k = 0; k < K; ++k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += v[k];
  if (lds_pad < 0) pad[threadIdx.x] = s;
  if (s == 12345.678) out[0] = s + pad[0];
}

template <int K>
static double run(const int cus, const int waves_per_simd, double clock_hz)
{
  // LDS per workgroup so that exactly 4 * waves_per_simd single-wave workgroups fit a CU (160 KB)
  const int per_cu = 4 * waves_per_simd;
  size_t lds = (160 * 1024) / per_cu;
  lds = lds / 512 * 512;
  if (lds > 64 * 1024) lds = 64 * 1024;
  const int iters = 4000 / K + 1;
  double* out;
  hipMalloc(&out, 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = cus * per_cu;
  hipFuncSetAttribute((const void*)chains<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(chains<K>, dim3(grid), dim3(64), lds, 0, iters, 1.0000001, 1e-9, out, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  hipFree(out);
  const double fmas_per_simd = (double)iters * 16 * K * waves_per_simd;   // wave-instructions per SIMD
  return best * 1e-3 * clock_hz / fmas_per_simd;                           // cycles per wave-instruction per SIMD
}

int main()
{
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 2;
  const int cus = p.multiProcessorCount;
  const double clk = p.clockRate * 1e3;
  printf("# %s, %d CUs, %.0f MHz; cycles per v_fma_f64 (wave64) per SIMD; 4.0 = full rate\n", p.name, cus, clk / 1e6);
  printf("# waves/SIMD :   K=1    K=2    K=3    K=4    K=6    K=8\n");
  for (int w : {1, 2, 3, 4, 5, 6, 8}) {
    printf("  %d          : %6.2f %6.2f %6.2f %6.2f %6.2f %6.2f\n", w, run<1>(cus, w, clk), run<2>(cus, w, clk), run<3>(cus, w, clk),
           run<4>(cus, w, clk), run<6>(cus, w, clk), run<8>(cus, w, clk));
  }
  return 0;
}
