// tools/lds_rate.hip — what a wave-wide LDS read costs a gfx950 CU, by address pattern: does the LDS serve lanes that
// read the SAME address faster than lanes that read different ones?  (It decides whether per-pair tables that many
// lanes share can be read from LDS at every use, or have to be held in registers.)
//   hipcc -O3 --offload-arch=gfx950 tools/lds_rate.hip -o tools/lds_rate && tools/lds_rate
// Prints LDS-pipe cycles per instruction per CU for ds_read_b64 / ds_read_b128 with 16 waves per CU.
#include <hip/hip_runtime.h>

#include <cstdio>

template <int WIDTH>
__global__ void __launch_bounds__(256) reads(const int iters, const int pattern, double* out)
{
  __shared__ __attribute__((aligned(16))) double tab[4096];
  for (int t = threadIdx.x; t < 4096; t += 256) tab[t] = t;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  int idx;   // in doubles
  switch (pattern) {
    case 0: idx = 0; break;                        // every lane the same address
    case 1: idx = (lane & 15) * 14; break;         // 16 distinct rows, 112 bytes apart (4 lanes share a row)
    case 2: idx = (lane & 15) * 18; break;         // 16 rows, 144 bytes apart
    case 3: idx = lane * 2; break;                 // 64 distinct, consecutive 16-byte chunks
    default: idx = lane * 14; break;               // 64 distinct rows, 112 bytes apart
  }
  const double* p = tab + idx;
  double acc = 0.0;
  for (int it = 0; it < iters; ++it) {
    // eight reads in flight per wave, then one wait: the pipe, not the latency, sets the rate
    if (WIDTH == 16) {
      double2 v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_read_b128 %0, %1" : "=v"(v[r]) : "v"((unsigned)(size_t)p));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(v[r].x), "+v"(v[r].y));
      acc += v[0].x + v[7].y;
    } else {
      double v[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("ds_read_b64 %0, %1" : "=v"(v[r]) : "v"((unsigned)(size_t)p));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int r = 0; r < 8; ++r) asm volatile("" : "+v"(v[r]));
      acc += v[0] + v[7];
    }
  }
  if (acc == 12345.678) out[0] = acc;
}

template <int WIDTH>
static double run(const int cus, const int pattern, const double clk)
{
  const int iters = 2000;
  double* out;
  (void)hipMalloc(&out, 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(reads<WIDTH>, dim3(cus * 4), dim3(256), 0, 0, iters, pattern, out);   // 16 waves per CU
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipFree(out);
  const double insts_per_cu = (double)iters * 8 * 16;
  return best * 1e-3 * clk / insts_per_cu;
}

int main()
{
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) return 2;
  const int cus = p.multiProcessorCount;
  const double clk = p.clockRate * 1e3;
  const char* names[5] = {"same address", "16 rows x 112 B", "16 rows x 144 B", "64 consecutive", "64 rows x 112 B"};
  printf("# cycles per wave-wide LDS read per CU (16 waves per CU, eight reads in flight per wave)\n");
  for (int pat = 0; pat < 5; ++pat)
    printf("%-18s ds_read_b64 %6.2f   ds_read_b128 %6.2f\n", names[pat], run<8>(cus, pat, clk), run<16>(cus, pat, clk));
  return 0;
}
